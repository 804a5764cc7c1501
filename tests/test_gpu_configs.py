"""BASELINE.json configs[3] and [4] at test scale, and edge cases of the ABI, on the GPU vs the oracle."""
import ctypes

import numpy as np
import pytest

from conftest import rot_err, trans_err

pytestmark = pytest.mark.gpu


def _oracle_pair(oracle, synth, S, T, voxel, r_pca, R, dof, corr, est_iou, pat, max_iter=80):
    ds, kp, feat = {}, {}, {}
    for name, cloud, d in (("T", T, 0), ("S", S, dof)):
        ds[name] = cloud[oracle.voxel_filter(cloud, voxel)]
        kp[name], _ = oracle.keypoints(ds[name], r_pca, R)
        feat[name], _, _ = oracle.bsc(ds[name], kp[name], R, d, pat)
    V = 4 if dof == 6 else 2
    FD = oracle.fd_bsc(feat["S"][:V], feat["T"][0])
    P = oracle.default_params(oracle.BSC, corr, dof, est_iou, R, oracle.bbx_magnitude(ds["S"]), max_iter=max_iter)
    return oracle.register(P, ds["S"][kp["S"]].astype(np.float64), ds["T"][kp["T"]].astype(np.float64), FD), kp


def test_cfg4_indoor_fragment_batch(ctx, api, oracle, synth):
    """configs[3]: a batch of 3DMatch-like fragment pairs (centimetres: voxel 1.25, r 5, R 15, BSC + NN) through the batched API."""
    import torch

    import bench

    CF = bench.CONFIGS[4]
    vx, rp, Rn = CF["voxel"], CF["r"], CF["R"]
    pat = synth.bsc_pattern_glibc()
    pairs = [synth.indoor_pair(i, 60_000) for i in range(3)]
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_NN, 6, 0.6, vx, rp, Rn, pat, max_iter=80)
    stats = ctx.register_pairs(cfg, [(torch.from_numpy(p.source).to(ctx.dev), torch.from_numpy(p.target).to(ctx.dev)) for p in pairs])
    iterating = 0
    for p, st in zip(pairs, stats):
        ro, kp = _oracle_pair(oracle, synth, p.source, p.target, vx, rp, Rn, 6, oracle.NN, 0.6, pat)
        iterating += int(ro["iters"] >= 5 and kp["S"].size >= 100)
        assert (st.k_s, st.k_t, st.iterations) == (kp["S"].size, kp["T"].size, ro["iters"])
        Rg = np.array(st.Rt[:]).reshape(4, 4)
        if np.isfinite(ro["Rt"]).all():
            assert rot_err(Rg, ro["Rt"]) < 1e-4 and trans_err(Rg, ro["Rt"]) < 1e-3
        else:
            assert not np.isfinite(Rg).all()
    assert iterating >= 2  # the configuration iterates on 150-250 keypoints (rounds 1-4: 1-3 iterations on 19-53 keypoints)


def test_cfg5_four_dof_km(ctx, api, oracle, synth):
    """configs[4] shape: levelled low-overlap pair, BSC + KM, dof 4 (only 2 source variants, ghicp_reg.cpp:178-182)."""
    p = synth.tls_pair(120_000, config_id=5)
    pat = synth.bsc_pattern_glibc()
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, 4, 0.3, 0.1, 0.5, 1.5, pat, max_iter=60)
    st, tr = ctx.register_pair(cfg, p.source, p.target)
    ro, kp = _oracle_pair(oracle, synth, p.source, p.target, 0.1, 0.5, 1.5, 4, oracle.KM, 0.3, pat, max_iter=60)
    assert (st.k_s, st.k_t, st.iterations) == (kp["S"].size, kp["T"].size, ro["iters"])
    assert [t["cor"] for t in tr] == [t["cor"] for t in ro["trace"]]
    Rg = np.array(st.Rt[:]).reshape(4, 4)
    assert rot_err(Rg, ro["Rt"]) < 1e-4 and trans_err(Rg, ro["Rt"]) < 1e-3


def test_edge_cases(ctx, api, oracle):
    import torch

    # no keypoints at all: the loop does not run, Rt stays identity
    pg = api.default_params(api.FEATURE_NONE, api.CORR_NN, 6, 0.6, 1.5, 10.0, max_iter=5)
    r = ctx.register(pg, np.zeros((0, 3)), np.zeros((0, 3)))
    assert r["iters"] == 0 and np.array_equal(r["Rt"], np.eye(4))
    # fewer than min_cor correspondences: one iteration, converged flag (ghicp_reg.cpp:796)
    rng = np.random.default_rng(1)
    a = rng.normal(size=(6, 3))
    ro = oracle.register(oracle.default_params(oracle.NONE, oracle.NN, 6, 0.6, 1.5, 10.0, max_iter=5), a, a + 0.01)
    rg = ctx.register(pg, a, a + 0.01)
    assert rg["iters"] == ro["iters"] == 1 and rg["trace"][0]["converged"] == 1
    np.testing.assert_allclose(rg["Rt"], ro["Rt"], atol=1e-6)
    # max_iter guard
    pk = api.default_params(api.FEATURE_NONE, api.CORR_NNR, 6, 0.6, 1.5, 100.0, max_iter=2)
    b = rng.normal(size=(200, 3)) * 5
    assert ctx.register(pk, b, b[::-1] + 1.0)["iters"] <= 2
    # argument validation returns an error code and a message instead of crashing
    bad = api.default_params(api.FEATURE_BSC, api.CORR_KM, 6, 0.6, 1.5, 10.0, max_iter=5)
    with pytest.raises(api.GhicpError):
        ctx.register(bad, a, a)  # BSC without an FD matrix
    with pytest.raises(api.GhicpError):
        ctx.voxel_filter(a.astype(np.float32), -1.0)
    with pytest.raises(api.GhicpError):
        ctx.bsc_encode(a.astype(np.float32), np.zeros(1, np.int32), 1.5, 6, np.full((49, 2), 77))
    # empty cloud front end
    assert ctx.keypoints(np.zeros((0, 3), np.float32), 0.5, 1.5).numel() == 0
    assert ctx.voxel_filter(np.zeros((0, 3), np.float32), 0.1).numel() == 0
    # a tiny cloud: nothing survives the prune (ptNum > 20), the pair API still returns cleanly
    tiny = torch.from_numpy(rng.normal(size=(50, 3)).astype(np.float32)).to(ctx.dev)
    st, _ = ctx.register_pair(api.pair_config(api.FEATURE_BSC, api.CORR_NN, 6, 0.6, 0.1, 0.5, 1.5, None, max_iter=5), tiny, tiny)
    assert (st.k_s, st.k_t, st.iterations) == (0, 0, 0)
    # host-pointer mode of the C ABI (what the C++ drop-in classes use)
    lib = ctx.lib
    h = ctypes.c_void_p()
    assert lib.ghicp_ctx_create(0, ctypes.byref(h)) == 0
    lib.ghicp_ctx_set_host_pointers(h, 1)
    W = np.ascontiguousarray([[-5, -2, -100], [-4, -2, -6], [-100, -1, -7]], dtype=np.float64)
    m = np.zeros(3, np.int32)
    assert lib.ghicp_km_solve(h, W.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(3), ctypes.c_double(0.01), m.ctypes.data_as(ctypes.c_void_p)) == 0
    assert m.tolist() == [0, 2, 1]
    lib.ghicp_ctx_destroy(h)


def test_host_pointer_mode_uploads_a_cloud_once_and_sees_in_place_changes(ctx, api, oracle, synth):
    """The C++ drop-in classes run the context in host-pointer mode; the reference's call sequence on one cloud (main:86-116: bounds,
    keypoints, BSC) must upload it once -- the staged copy is found again by address + size + content fingerprint -- and a cloud that was
    CHANGED in place between two calls must be uploaded again (ghicp_ctx_stage_stats)."""
    lib = ctx.lib
    h = ctypes.c_void_p()
    assert lib.ghicp_ctx_create(0, ctypes.byref(h)) == 0
    lib.ghicp_ctx_set_host_pointers(h, 1)

    def stats():
        a, b, c = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        assert lib.ghicp_ctx_stage_stats(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
        return a.value, b.value, c.value

    p = synth.tls_pair(120_000, pair_id=9)
    raw = np.ascontiguousarray(p.target[:, :3], np.float32)  # 1.4 MB: above the 256 KB threshold
    vp = ctypes.c_void_p

    def voxel(x):
        keep = np.zeros(x.shape[0] + 1, np.int32)
        m = ctypes.c_int64(0)
        assert lib.ghicp_voxel_filter(h, x.ctypes.data_as(vp), ctypes.c_int64(x.shape[0]), 3, ctypes.c_float(0.1), keep.ctypes.data_as(vp), ctypes.byref(m)) == 0
        return keep[:m.value].copy()

    def bounds(x):
        out = (ctypes.c_double * 6)()
        assert lib.ghicp_cloud_bounds(h, x.ctypes.data_as(vp), ctypes.c_int64(x.shape[0]), 3, out) == 0
        return np.array(out[:])

    k1 = voxel(raw)
    assert stats()[:2] == (0, 1)
    b1 = bounds(raw)
    k2 = voxel(raw)
    assert stats()[:2] == (2, 1) and stats()[2] >= raw.nbytes  # two more calls on the same cloud: no upload
    np.testing.assert_array_equal(k1, k2)
    np.testing.assert_array_equal(k1, oracle.voxel_filter(raw, 0.1))
    raw[7] += 1000.0  # same address, same size, other content: must not be served from the kept copy
    b2 = bounds(raw)
    assert stats()[:2] == (2, 2)
    assert b2[3] > b1[3] + 500.0 or b2[4] > b1[4] + 500.0
    np.testing.assert_array_equal(voxel(raw), oracle.voxel_filter(raw, 0.1))
    assert stats()[:2] == (3, 2)
    # many distinct clouds: the cache stays bounded (16 arrays)
    others = [np.ascontiguousarray(raw + float(i), np.float32) for i in range(20)]
    for o in others:
        bounds(o)
    hits, misses, kept = stats()
    assert (hits, misses) == (3, 22) and kept <= 16 * raw.nbytes
    # only point clouds are kept: a 3 MB Kuhn-Munkres weight matrix is staged for its call and gone with it (no fingerprint, no entry)
    W = np.ascontiguousarray(-np.abs(np.random.default_rng(3).normal(size=(620, 620))) - 1.0)
    m = np.zeros(620, np.int32)
    assert W.nbytes > 256 * 1024
    assert lib.ghicp_km_solve(h, W.ctypes.data_as(vp), ctypes.c_int64(620), ctypes.c_double(0.01), m.ctypes.data_as(vp)) == 0
    assert sorted(m.tolist()) == list(range(620)) and stats() == (hits, misses, kept)
    # ghicp_ctx_stage_clear frees the kept copies; so does leaving host-pointer mode
    assert kept > 0 and lib.ghicp_ctx_stage_clear(h) == 0 and stats() == (hits, misses, 0)
    bounds(raw)
    assert stats()[1:] == (misses + 1, raw.nbytes)
    lib.ghicp_ctx_set_host_pointers(h, 0)
    assert stats()[2] == 0
    lib.ghicp_ctx_destroy(h)
