"""BASELINE.json's full size (configs[1]: 1 M points per scan) through properties that do not need the CPU restatement to
finish: voxel-filter determinism, keypoint ordering / separation / maximality, self-overlap and self-registration, closeness to
the ground-truth pose, and bit-equality of the two product paths (pair API vs cached clouds).  (bench.py additionally compares
one full-size pair against the CPU path on every run.)"""
import numpy as np
import pytest

from conftest import rot_err, trans_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scan(synth):
    return synth.tls_pair(1_000_000)


def test_front_end_properties_at_full_size(ctx, scan):
    import torch

    cloud = torch.from_numpy(scan.target).to(ctx.dev)
    keep = ctx.voxel_filter(cloud, 0.1)
    ds = cloud[keep.long()].contiguous()
    assert int(keep[0]) == 0 and 200_000 < ds.shape[0] < 600_000
    # deterministic; a second pass can only merge (the voxel lattice is anchored at the min corner of ITS input, filter.hpp:33,
    # so the filter is not idempotent) and keeps the lowest index of every voxel: ascending voxel keys, no repeated index
    np.testing.assert_array_equal(ctx.voxel_filter(cloud, 0.1).cpu().numpy(), keep.cpu().numpy())
    keep2 = ctx.voxel_filter(ds, 0.1).cpu().numpy()
    assert 0.9 * keep.shape[0] < keep2.shape[0] <= keep.shape[0]
    assert keep2[0] == 0 and np.unique(keep2[1:]).size == keep2.size - 1
    # keypoints: descending curvature, pairwise separation >= R (suppression is d^2 < R^2), maximal among the candidates
    lam, curv, cnt = ctx.pca_curvature(ds, 0.5)
    cand = ctx.prune(lam, cnt)
    kp = ctx.nms(ds, curv, cand, 1.5)
    np.testing.assert_array_equal(kp.cpu().numpy(), ctx.keypoints(ds, 0.5, 1.5).cpu().numpy())
    ck = curv[kp.long()].cpu().numpy()
    assert 300 < kp.shape[0] < 5000 and (np.diff(ck) <= 0).all()
    P = ds[kp.long()][:, :3].double()
    d2 = torch.cdist(P, P).pow(2)
    d2.fill_diagonal_(1e9)
    assert float(d2.min()) >= 1.5 ** 2 * (1 - 1e-6)
    C = ds[cand.long()][:, :3].double()
    nearest = torch.cdist(C, P).min(dim=1).values
    assert float(nearest.max()) < 1.5  # every candidate is a keypoint or suppressed by one


def test_fine_registration_identities_at_full_size(ctx, api, scan):
    import torch

    cloud = torch.from_numpy(scan.target).to(ctx.dev)
    ds = cloud[ctx.voxel_filter(cloud, 0.1).long()][:, :3].contiguous()
    n = ds.shape[0]
    assert ctx.cal_overlap(ds, ds, 0.05) == np.float32((0.01 + n) / n)
    idx, d2 = ctx.nn_search(ds, ds)
    assert float(d2.max()) == 0.0  # every point finds itself (or an exact duplicate with a lower index)
    assert bool((ds[idx.long()] == ds).all())
    r = ctx.icp(ds, ds, api.icp_params(10, trimmed=True, thre_dis=0.05))
    assert r["done"] == 1 and r["converged"] == 1 and r["iterations"] <= 2
    np.testing.assert_allclose(r["T"], np.eye(4), atol=1e-6)
    assert r["fitness"] < 1e-10


def test_pair_at_full_size_close_to_ground_truth_and_paths_agree(ctx, api, synth, scan):
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, 6, 0.6, 0.1, 0.5, 1.5, synth.bsc_pattern_glibc(), max_iter=200)
    st, _ = ctx.register_pair(cfg, scan.source, scan.target, want_trace=False)
    Rt = np.array(st.Rt[:]).reshape(4, 4)
    assert st.converged == 1 and 5 <= st.iterations < 200
    assert rot_err(Rt, scan.gt) < 0.03 and trans_err(Rt, scan.gt) < 0.3  # coarse registration: within the BSC/KM basin of the truth
    S, T = ctx.cloud_create(cfg, scan.source), ctx.cloud_create(cfg, scan.target)
    st2 = ctx.register_clouds(cfg, [(S, T)])[0]
    assert (st2.k_s, st2.k_t, st2.iterations) == (st.k_s, st.k_t, st.iterations)
    np.testing.assert_array_equal(np.array(st2.Rt[:]), np.array(st.Rt[:]))


@pytest.mark.parametrize("config_id", [3, 5])
def test_cfg3_cfg5_at_full_size(ctx, api, synth, config_id):
    """BASELINE.json configs[2] (5 M points per scan, FPFH + reciprocal NN) and configs[4] (10 M points, low overlap, levelled,
    BSC + KM with the 4-DoF variant set) at FULL size, through properties that do not need the CPU restatement on the GPU box:
    the registration is deterministic (two runs bit-identical), the pair API and the cached-cloud API agree bit for bit, the
    keypoint / iteration counts are the ones the oracle produced OFF the box for the same seeds (profiles/r02_cfg{3,5}_fullsize_
    parity.json: identical keypoints and iterations, 4x4 within 1e-6), and the KM path respects the max_iter guard."""
    import json
    import os

    import bench  # the config table of the benchmark (repo root is on sys.path through conftest)

    CF = bench.CONFIGS[config_id]
    p = bench.make_pair(config_id, 0, CF["hits"])
    feature = {"BSC": api.FEATURE_BSC, "FPFH": api.FEATURE_FPFH}[CF["feature"]]
    corr = {"KM": api.CORR_KM, "NN": api.CORR_NN, "NNR": api.CORR_NNR}[CF["corr"]]
    max_iter = 40 if config_id == 5 else 200  # cfg5 pair 0 does not converge within 200 iterations (19 s): the guard is what is tested
    cfg = api.pair_config(feature, corr, CF["dof"], CF["iou"], CF["voxel"], CF["r"], CF["R"], synth.bsc_pattern_glibc(), max_iter=max_iter)
    st, _ = ctx.register_pair(cfg, p.source, p.target, want_trace=False)
    st_again, _ = ctx.register_pair(cfg, p.source, p.target, want_trace=False)
    assert st.n_s == CF["hits"] and 200_000 < st.m_s < 1_500_000 and st.k_s > 100 and st.k_t > 100
    assert (st.k_s, st.k_t, st.iterations) == (st_again.k_s, st_again.k_t, st_again.iterations)
    np.testing.assert_array_equal(np.array(st.Rt[:]), np.array(st_again.Rt[:]))
    assert np.isfinite(np.array(st.Rt[:])).all()
    S, T = ctx.cloud_create(cfg, p.source), ctx.cloud_create(cfg, p.target)
    st2 = ctx.register_clouds(cfg, [(S, T)])[0]
    assert (st2.k_s, st2.k_t, st2.iterations) == (st.k_s, st.k_t, st.iterations)
    np.testing.assert_array_equal(np.array(st2.Rt[:]), np.array(st.Rt[:]))
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_cfg%d_fullsize_parity.json" % config_id)
    o = json.load(open(ref))["pairs"][0]["oracle"]
    assert (st.m_s, st.m_t, st.k_s, st.k_t) == (o["m_s"], o["m_t"], o["k_s"], o["k_t"])
    if config_id == 3:
        assert st.iterations == o["iterations"] and st.converged == 1
    else:
        assert st.iterations == max_iter  # the oracle needs all 200 iterations for this pair as well (ghicp_reg.cpp:49 has no guard)
