"""GPU parity of the fine-registration path (CRegistration::icp_reg / ptplicp_reg / calOverlap / transformcloud /
invTransform, reference src/common_reg.cpp:45-199, 294-370) against the CPU restatement, through the C ABI.
Index work (1-NN correspondences, overlap counts, iteration counts) is bit-exact; the 4x4 is held to the
north-star tolerance (1e-4 rotation, 1e-3 m translation) and in practice agrees to float rounding."""
import numpy as np
import pytest

from conftest import rot_err, trans_err
from test_icp_cpu import small_pair

pytestmark = pytest.mark.gpu


def test_nn_search_bit_exact(ctx, oracle, synth):
    src, tgt, _ = small_pair(synth, n=20000)
    far = np.array([[500.0, -300.0, 80.0], [-1000.0, 0.0, 0.0], [0.0, 0.0, 400.0]], np.float32)  # resolved on the coarse grid
    q = np.vstack([src, far, tgt[:100]])
    io, do = oracle.nn1(q, tgt)
    ig, dg = ctx.nn_search(q, tgt)
    np.testing.assert_array_equal(ig.cpu().numpy(), io)
    np.testing.assert_array_equal(dg.cpu().numpy(), do)
    # duplicates in the target: ties go to the lower index
    dup = np.vstack([tgt[:500], tgt[:500]])
    ig, dg = ctx.nn_search(tgt[:500], dup)
    np.testing.assert_array_equal(ig.cpu().numpy(), np.arange(500))
    assert float(dg.abs().max()) == 0.0
    # a single target point
    ig, _ = ctx.nn_search(src[:10], tgt[:1])
    np.testing.assert_array_equal(ig.cpu().numpy(), np.zeros(10, np.int32))


def test_cal_overlap(ctx, oracle, synth):
    src, tgt, _ = small_pair(synth, n=20000)
    for r in (0.05, 0.2, 1.0):
        assert ctx.cal_overlap(src, tgt, r) == oracle.cal_overlap(src, tgt, r)
    assert ctx.cal_overlap(src + np.float32(900.0), tgt, 0.2) == oracle.cal_overlap(src + np.float32(900.0), tgt, 0.2)


def test_knn_normals(ctx, oracle, synth):
    _, tgt, _ = small_pair(synth, n=8000)
    for k in (5, 12, 20):
        no = oracle.knn_normals(tgt, k)
        ng = ctx.knn_normals(tgt, k).cpu().numpy()
        assert np.abs(ng - no).max() <= 1e-6
    np.testing.assert_array_equal(ctx.knn_normals(tgt[:2], 5).cpu().numpy(), np.full((2, 3), 0.577, np.float32))


def test_transform_and_inverse(ctx, api, oracle, synth):
    src, _, gt = small_pair(synth, n=5000)
    T = gt.astype(np.float32)
    np.testing.assert_array_equal(ctx.transform_cloud_f32(src, T).cpu().numpy(), oracle.transform_cloud(src, T.astype(np.float64)))
    np.testing.assert_array_equal(api.inv_transform(T), oracle.inv_transform(T))


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("reciprocal,trimmed", [(False, False), (False, True), (True, False), (True, True)])
def test_icp_matches_cpu(ctx, api, oracle, synth, metric, reciprocal, trimmed):
    src, tgt, gt = small_pair(synth, n=12000)
    ro = oracle.icp(src, tgt, oracle.icp_params(40, reciprocal, trimmed, metric, 0.2, 0.1, 12))
    rg = ctx.icp(src, tgt, api.icp_params(40, reciprocal, trimmed, metric, 0.2, 0.1, 12))
    assert rg["done"] == ro["done"] == 1
    assert rg["overlap"] == ro["overlap"]
    assert (rg["iterations"], rg["converged"], rg["reason"]) == (ro["iterations"], ro["converged"], ro["reason"])
    assert rg["correspondences"] == ro["correspondences"]
    Tg, To = rg["T"].astype(np.float64), ro["T"].astype(np.float64)
    assert rot_err(Tg, To) <= 1e-4 and trans_err(Tg, To) <= 1e-3
    assert rot_err(Tg, To) <= 5e-6 and trans_err(Tg, To) <= 5e-5  # what the shared numerics contract actually delivers
    assert rot_err(Tg, gt) < 2e-3 and trans_err(Tg, gt) < 0.02
    np.testing.assert_allclose(rg["transformed"].cpu().numpy(), ro["transformed"], atol=2e-4)
    np.testing.assert_allclose(rg["mse"], ro["mse"], rtol=1e-3)
    np.testing.assert_allclose(rg["fitness"], ro["fitness"], rtol=1e-3)


def test_icp_after_coarse_registration_of_a_scan_pair(ctx, api, oracle, synth):
    """The use the reference documents: fine registration of the down-sampled clouds after GH-ICP."""
    pair = synth.tls_pair(120_000)
    S = pair.source[oracle.voxel_filter(pair.source, 0.2)][:, :3]
    T = pair.target[oracle.voxel_filter(pair.target, 0.2)][:, :3]
    a = np.deg2rad(1.0)
    d = np.eye(4)
    d[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    d[:3, 3] = [0.15, -0.1, 0.03]
    coarse = d @ pair.gt  # a coarse estimate 1 degree / 0.18 m off the truth
    S0 = oracle.transform_cloud(S, coarse)
    for metric in (0, 1):
        ro = oracle.icp(S0, T, oracle.icp_params(30, False, True, metric, 0.3, 0.1, 15))
        rg = ctx.icp(S0, T, api.icp_params(30, False, True, metric, 0.3, 0.1, 15))
        assert rg["overlap"] == ro["overlap"] and rg["iterations"] == ro["iterations"] and rg["reason"] == ro["reason"]
        Tg, To = rg["T"].astype(np.float64), ro["T"].astype(np.float64)
        assert rot_err(Tg, To) <= 1e-4 and trans_err(Tg, To) <= 1e-3
        total = Tg @ coarse
        assert rot_err(total, pair.gt) < 5e-3 and trans_err(total, pair.gt) < 0.05  # pulls the estimate back to the truth


def test_icp_edge_cases(ctx, api, synth):
    src, tgt, _ = small_pair(synth, n=3000)
    r = ctx.icp(src + np.float32(500.0), tgt, api.icp_params(10, False, True, 0, 0.2, 0.5))
    assert r["done"] == 0 and r["overlap"] < 0.01 and not r["T"].any()  # refused: T untouched (reference returns false)
    r = ctx.icp(src, tgt, api.icp_params(2, False, False, 0))
    assert r["iterations"] == 2 and r["reason"] == 1
    r = ctx.icp(src[:2], tgt, api.icp_params(5, False, False, 0))
    assert r["done"] == 1 and r["converged"] == 0 and r["reason"] == 5 and r["iterations"] == 0  # < 3 correspondences
    np.testing.assert_array_equal(r["T"], np.eye(4, dtype=np.float32))
    r = ctx.icp(np.zeros((0, 3), np.float32), tgt, api.icp_params(5))
    assert r["reason"] == 5 and r["iterations"] == 0
    with pytest.raises(api.GhicpError):
        ctx.icp(src, tgt, api.icp_params(5, metric=1, covariance_k=64))


def test_trimmed_rejector_splits_ties_by_index(ctx, api, oracle, synth):
    """Repeated source points give equal distances at the trimming threshold: the kept set is the lowest indices."""
    src, tgt, _ = small_pair(synth, n=6000)
    near = np.flatnonzero(oracle.nn1(src, tgt)[1] < 0.1 ** 2)[:2]  # two unrepeated points inside the overlap radius
    src4 = np.vstack([np.repeat(np.delete(src, near, axis=0), 4, axis=0), src[near]])
    for max_iter in (1, 6):
        ro = oracle.icp(src4, tgt, oracle.icp_params(max_iter, False, True, 0, 0.2, 0.05))
        rg = ctx.icp(src4, tgt, api.icp_params(max_iter, False, True, 0, 0.2, 0.05))
        assert rg["overlap"] == ro["overlap"] and rg["correspondences"] == ro["correspondences"]
        assert rg["correspondences"] % 4 == 2  # the threshold falls inside a group of four equal distances
        assert (rg["iterations"], rg["reason"]) == (ro["iterations"], ro["reason"])
        np.testing.assert_allclose(rg["mse"], ro["mse"], rtol=1e-9)
        assert rot_err(rg["T"].astype(np.float64), ro["T"].astype(np.float64)) <= 5e-6
        assert trans_err(rg["T"].astype(np.float64), ro["T"].astype(np.float64)) <= 5e-5
