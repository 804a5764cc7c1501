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


def test_overlap_of_icp_equals_cal_overlap_back_to_back_in_host_pointer_mode(ctx, api, oracle, synth):
    """Round-5 verdict, weak #1 (ii): CRegistration::calOverlap and then icp_reg on the SAME two host arrays (tests/cpp/test_dropin.cpp), in
    host-pointer mode, right after a Kuhn-Munkres registration on the same context -- the second call finds both clouds in the staged-input
    cache (ctx.h).  The ratio ghicp_icp computes for itself (common_reg.cpp:64-74) must be the float ghicp_cal_overlap returned, bit for
    bit, and `done` must flip exactly at min_overlap_for_reg == that float (`ratio < min_overlap` refuses, common_reg.cpp:66)."""
    import ctypes as C

    lib = ctx.lib
    h = C.c_void_p()
    assert lib.ghicp_ctx_create(0, C.byref(h)) == 0
    lib.ghicp_ctx_set_host_pointers(h, 1)
    vp = C.c_void_p

    def stats():
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        assert lib.ghicp_ctx_stage_stats(h, C.byref(a), C.byref(b), C.byref(c)) == 0
        return a.value, b.value

    try:
        # a Kuhn-Munkres registration first (host arrays in, like GHRegistration::ghicp_reg through the drop-in header)
        g = synth.gauss_pair(n=20_000, n_kp=300)
        kpS = np.ascontiguousarray(g.source[g.kp_source], np.float64)
        kpT = np.ascontiguousarray(g.target[g.kp_target], np.float64)
        P = api.default_params(api.FEATURE_NONE, api.CORR_KM, 6, 0.9, 1.5, oracle.bbx_magnitude(g.source), max_iter=40)
        Rt = (C.c_double * 16)()
        n_iter = C.c_int32(0)
        assert lib.ghicp_register(h, C.byref(P), kpS.ctypes.data_as(vp), C.c_int64(300), kpT.ctypes.data_as(vp), C.c_int64(300), None, Rt, None,
                                  C.byref(n_iter), None) == 0
        ro = oracle.register(oracle.default_params(oracle.NONE, oracle.KM, 6, 0.9, 1.5, oracle.bbx_magnitude(g.source), max_iter=40), kpS, kpT)
        assert n_iter.value == ro["iters"] and np.abs(np.array(Rt[:]).reshape(4, 4) - ro["Rt"]).max() < 1e-6

        pair = synth.tls_pair(120_000, pair_id=5)
        S = np.ascontiguousarray(pair.source[oracle.voxel_filter(pair.source, 0.1)][:, :3], np.float32)
        T = np.ascontiguousarray(pair.target[oracle.voxel_filter(pair.target, 0.1)][:, :3], np.float32)
        assert S.nbytes >= 256 * 1024 and T.nbytes >= 256 * 1024  # both above the staging threshold: the cached path is the one under test
        # coarse poses from "registered" to "barely overlapping": the ratio must agree at every one of them
        for k, (deg, shift) in enumerate([(0.5, 0.05), (4.0, 1.5), (9.0, 4.0), (25.0, 12.0)]):
            a = np.deg2rad(deg)
            d = np.eye(4)
            d[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
            d[:3, 3] = [shift, -0.5 * shift, 0.02]
            S1 = np.ascontiguousarray(oracle.transform_cloud(S, d @ pair.gt), np.float32)
            ratio = C.c_float(-1.0)
            h0, m0 = stats()
            assert lib.ghicp_cal_overlap(h, S1.ctypes.data_as(vp), C.c_int64(len(S1)), 3, T.ctypes.data_as(vp), C.c_int64(len(T)), 3, C.c_float(0.3),
                                         C.byref(ratio)) == 0
            r = np.float32(ratio.value)
            assert r == np.float32(oracle.cal_overlap(S1, T, 0.3)), (k, r)
            for min_ov, want_done in ((float(r), 1), (float(np.nextafter(r, np.float32(2.0))), 0), (0.1, int(r >= np.float32(0.1)))):
                prm = api.icp_params(3, False, True, 0, 0.3, min_ov)
                st = api.IcpStats()
                T16 = np.zeros(16, np.float32)
                out = np.zeros((len(S1), 3), np.float32)
                assert lib.ghicp_icp(h, S1.ctypes.data_as(vp), C.c_int64(len(S1)), 3, T.ctypes.data_as(vp), C.c_int64(len(T)), 3, C.byref(prm),
                                     T16.ctypes.data_as(vp), out.ctypes.data_as(vp), C.byref(st)) == 0
                assert np.float32(st.overlap) == r, "pose %d: ghicp_icp overlap %r, ghicp_cal_overlap %r" % (k, st.overlap, float(r))
                assert st.done == want_done, "pose %d: overlap %r against min_overlap %r -> done %d" % (k, float(r), min_ov, st.done)
                io = oracle.icp(S1, T, oracle.icp_params(3, False, True, 0, 0.3, min_ov))
                assert io["done"] == want_done and np.float32(io["overlap"]) == r
            h1, m1 = stats()
            # S1 is new at every pose (one upload), T is uploaded at the first pose only; the three ghicp_icp calls find both again
            assert m1 - m0 == (2 if k == 0 else 1) and h1 - h0 == 6 + (0 if k == 0 else 1), (k, h0, m0, h1, m1)
    finally:
        lib.ghicp_ctx_destroy(h)
