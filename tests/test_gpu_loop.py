"""GPU parity of the iteration loop, FD kernels and KM against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest

from conftest import rot_err, trans_err

pytestmark = pytest.mark.gpu


def _cfg1(synth, oracle, n_kp=2000):
    p = synth.gauss_pair(n_kp=n_kp)
    kpS = p.source[p.kp_source].astype(np.float64)
    kpT = p.target[p.kp_target].astype(np.float64)
    return p, kpS, kpT, oracle.bbx_magnitude(p.source)


def _compare_traces(tg, to, rel=1e-9):
    assert len(tg) == len(to)
    for a, b in zip(tg, to):
        assert a["cor"] == b["cor"] and a["converged"] == b["converged"]
        for k in ("penalty", "cdmean", "cdstd", "rmse", "rmse_after", "iou", "para1", "para2", "fdm", "fdstd"):
            if np.isnan(b[k]):
                assert np.isnan(a[k]), k
            else:
                assert a[k] == pytest.approx(b[k], rel=rel, abs=1e-12), k
        np.testing.assert_allclose(a["Rt"], b["Rt"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("corr", [0, 1])
def test_loop_none_nn_nnr_cfg1(ctx, api, synth, oracle, corr):
    """cfg1 (BASELINE configs[0]): N/N 6-DoF; GPU loop vs oracle, iteration by iteration."""
    p, kpS, kpT, bbx = _cfg1(synth, oracle)
    po = oracle.default_params(oracle.NONE, corr, 6, 0.9, 1.5, bbx)
    ro = oracle.register(po, kpS, kpT, want_matchlist=True)
    pg = api.default_params(api.FEATURE_NONE, corr, 6, 0.9, 1.5, bbx)
    rg = ctx.register(pg, kpS, kpT, want_matchlist=True)
    assert rg["iters"] == ro["iters"]
    np.testing.assert_array_equal(rg["matchlist"], ro["matchlist"])  # index work: bit-exact
    _compare_traces(rg["trace"], ro["trace"])
    # north_star tolerance: 1e-4 rotation, 1e-3 m translation
    assert rot_err(rg["Rt"], ro["Rt"]) < 1e-4 and trans_err(rg["Rt"], ro["Rt"]) < 1e-3
    assert rot_err(rg["Rt"], p.gt) < 1e-3 and trans_err(rg["Rt"], p.gt) < 5e-3


def _fake_bsc_fd(rng, ks, kt):
    FD = rng.integers(60, 200, size=(ks, kt)).astype(np.float64)
    idx = np.arange(min(ks, kt))
    FD[idx, idx] = rng.integers(5, 40, size=idx.size)
    return FD


@pytest.mark.parametrize("corr", [0, 1, 2])
def test_loop_bsc_energy(ctx, api, synth, oracle, corr):
    """BSC-style hybrid energy (u16 FD) with NN / NNR / KM matching on unequal keypoint counts."""
    import torch

    p, kpS, kpT, bbx = _cfg1(synth, oracle, n_kp=700)
    kpS, kpT = kpS[:600], kpT[:700]
    rng = np.random.default_rng(3 + corr)
    FD = _fake_bsc_fd(rng, 600, 700)
    po = oracle.default_params(oracle.BSC, corr, 6, 0.6, 1.5, bbx, max_iter=40)
    ro = oracle.register(po, kpS, kpT, FD, want_matchlist=True)
    pg = api.default_params(api.FEATURE_BSC, corr, 6, 0.6, 1.5, bbx, max_iter=40)
    FDg = torch.from_numpy(FD.astype(np.int16)).to(ctx.dev)
    rg = ctx.register(pg, kpS, kpT, FDg, want_matchlist=True)
    assert rg["iters"] == ro["iters"]
    np.testing.assert_array_equal(rg["matchlist"], ro["matchlist"])
    _compare_traces(rg["trace"], ro["trace"])
    np.testing.assert_allclose(rg["Rt"], ro["Rt"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("corr", [0, 1])
def test_loop_fpfh_energy(ctx, api, synth, oracle, corr):
    import torch

    p, kpS, kpT, bbx = _cfg1(synth, oracle, n_kp=500)
    rng = np.random.default_rng(11)
    FD = (0.2 + 0.6 * rng.random((500, 500))).astype(np.float32)
    FD[np.arange(500), np.arange(500)] = 0.97
    po = oracle.default_params(oracle.FPFH, corr, 6, 0.6, 1.5, bbx, max_iter=40)
    ro = oracle.register(po, kpS, kpT, FD.astype(np.float64), want_matchlist=True)
    pg = api.default_params(api.FEATURE_FPFH, corr, 6, 0.6, 1.5, bbx, max_iter=40)
    rg = ctx.register(pg, kpS, kpT, torch.from_numpy(FD).to(ctx.dev), want_matchlist=True)
    assert rg["iters"] == ro["iters"]
    np.testing.assert_array_equal(rg["matchlist"], ro["matchlist"])
    _compare_traces(rg["trace"], ro["trace"], rel=1e-7)  # device pow() vs glibc pow(): <= 1 ulp apart
    np.testing.assert_allclose(rg["Rt"], ro["Rt"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("feature,corr", [("none", 0), ("bsc", 1), ("bsc", 2)])
def test_iterate_equals_register(ctx, api, synth, oracle, feature, corr):
    """ghicp_loop_create + ghicp_iterate (one pass of ghicp_reg.cpp:49-103 per call): the same records, match lists and final 4x4 as
    ghicp_register, bit for bit, and the oracle's iteration count; iterating a converged loop is an error."""
    import torch

    p, kpS, kpT, bbx = _cfg1(synth, oracle, n_kp=500)
    kpS, kpT = kpS[:430], kpT[:500]
    FDg, FDo = None, None
    if feature == "bsc":
        FDo = _fake_bsc_fd(np.random.default_rng(21 + corr), 430, 500)
        FDg = torch.from_numpy(FDo.astype(np.int16)).to(ctx.dev)
    ft_g, ft_o = (api.FEATURE_BSC, oracle.BSC) if feature == "bsc" else (api.FEATURE_NONE, oracle.NONE)
    pg = api.default_params(ft_g, corr, 6, 0.6 if feature == "bsc" else 0.9, 1.5, bbx, max_iter=40)
    whole = ctx.register(pg, kpS, kpT, FDg, want_matchlist=True)
    steps = ctx.register_stepwise(pg, kpS, kpT, FDg)
    assert steps["iters"] == whole["iters"] == len(steps["trace"])
    np.testing.assert_array_equal(steps["matchlist"], whole["matchlist"])
    for a, b in zip(steps["trace"], whole["trace"]):
        for k in a:
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=k)  # identical bits, NaN == NaN
    np.testing.assert_array_equal(steps["Rt"], whole["Rt"])
    ro = oracle.register(oracle.default_params(ft_o, corr, 6, 0.6 if feature == "bsc" else 0.9, 1.5, bbx, max_iter=40), kpS, kpT, FDo)
    assert steps["iters"] == ro["iters"] and steps["converged"] == int(ro["trace"][-1]["converged"])
    if steps["converged"]:
        with pytest.raises(Exception):
            ctx.register_stepwise(pg, kpS, kpT, FDg, extra_calls=1)


def test_final_transform_of_a_batch(ctx, oracle):
    """S7 (main:153) for a batch in one launch (ghicp_transform_clouds): bit-identical to ghicp_transform_cloud cloud by cloud and to the
    restatement, for packed clouds (the float4 path incl. its 1-3 point tail), an empty cloud and a NaN transform."""
    import torch

    rng = np.random.default_rng(5)
    sizes = [4096, 1, 0, 4099, 30001, 2]
    clouds = [torch.from_numpy((rng.standard_normal((n, 3)) * 20).astype(np.float32)).to(ctx.dev) for n in sizes]
    Rts = []
    for i in range(len(sizes)):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        Rt = np.eye(4)
        Rt[:3, :3] = q
        Rt[:3, 3] = rng.standard_normal(3) * 5
        if i == 4:
            Rt[0, 3] = np.nan
        Rts.append(Rt)
    outs = [torch.full((n, 3), -7.0, dtype=torch.float32, device=ctx.dev) for n in sizes]
    ctx.transform_clouds(clouds, Rts, outs)
    for c, Rt, o in zip(clouds, Rts, outs):
        single = ctx.transform_cloud(c, Rt).cpu().numpy() if c.shape[0] else np.zeros((0, 3), np.float32)
        np.testing.assert_array_equal(o.cpu().numpy(), single)
        np.testing.assert_array_equal(o.cpu().numpy(), oracle.transform_cloud(c.cpu().numpy(), Rt))
    # a strided view base that is not 16-byte aligned takes the scalar path: same bits
    big = torch.from_numpy((rng.standard_normal((1001, 3)) * 3).astype(np.float32)).to(ctx.dev)
    off = big[1:]  # 12 bytes past an aligned base
    o2 = torch.empty((1000, 3), dtype=torch.float32, device=ctx.dev)
    ctx.transform_clouds([off], [Rts[0]], [o2])
    np.testing.assert_array_equal(o2.cpu().numpy(), oracle.transform_cloud(off.cpu().numpy(), Rts[0]))


def test_km_kat_and_random(ctx, oracle):
    """km.cpp:237-259 known-answer vector + random matrices vs the oracle (and the reference's own km.cpp)."""
    W = np.array([[-5, -2, -100], [-4, -2, -6], [-100, -1, -7]], float)
    assert ctx.km_solve(W).cpu().numpy().tolist() == [0, 2, 1]
    rng = np.random.default_rng(5)
    for n, frac in ((1, 0), (2, 0), (7, 0.3), (64, 0.1), (257, 0.02), (1000, 0.004), (1500, 0.01)):
        cd = 5 + 60 * rng.random((n, n))
        cd[np.arange(n), (np.arange(n) * 7) % n] = 3 * rng.random(n)
        cd = np.where(rng.random((n, n)) < frac, 8 * rng.random((n, n)), cd)
        w = np.where(cd < 8.0, -cd, -8.0)
        m_o, _ = oracle.km(w)
        m_g = ctx.km_solve(w).cpu().numpy()
        np.testing.assert_array_equal(m_g, m_o)
        if n <= 300:
            m_r = oracle.km_reference(w)
            if m_r is not None:
                np.testing.assert_array_equal(m_g, m_r)


def test_km_all_equal_and_loop_km_none(ctx, api, synth, oracle):
    w = np.full((33, 33), -2.5)
    np.testing.assert_array_equal(ctx.km_solve(w).cpu().numpy(), oracle.km(w)[0])
    p, kpS, kpT, bbx = _cfg1(synth, oracle, n_kp=400)
    po = oracle.default_params(oracle.NONE, oracle.KM, 6, 0.9, 1.5, bbx, max_iter=30)
    ro = oracle.register(po, kpS, kpT, want_matchlist=True)
    pg = api.default_params(api.FEATURE_NONE, api.CORR_KM, 6, 0.9, 1.5, bbx, max_iter=30)
    rg = ctx.register(pg, kpS, kpT, want_matchlist=True)
    assert rg["iters"] == ro["iters"]
    np.testing.assert_array_equal(rg["matchlist"], ro["matchlist"])
    _compare_traces(rg["trace"], ro["trace"])


def test_fd_bsc_hamming(ctx, oracle):
    rng = np.random.default_rng(2)
    for V, ks, kt in ((1, 1, 1), (4, 130, 77), (2, 64, 200), (4, 257, 65)):
        fS = rng.integers(0, 256, size=(V, ks, 56), dtype=np.uint8)
        fT = rng.integers(0, 256, size=(kt, 56), dtype=np.uint8)
        FDg = ctx.fd_bsc(fS, fT).cpu().numpy().astype(np.int64) & 0xFFFF
        FDo = oracle.fd_bsc(fS, fT).astype(np.int64)
        np.testing.assert_array_equal(FDg, FDo)
        # independent check: popcount via numpy
        ref = np.unpackbits(fS[:, :, None, :] ^ fT[None, None, :, :], axis=-1).sum(-1).min(0)
        np.testing.assert_array_equal(FDg, ref)


def test_fd_fpfh_correlation(ctx, oracle):
    rng = np.random.default_rng(4)
    hS = (rng.random((150, 33)) * 100).astype(np.float32)
    hT = (rng.random((90, 33)) * 100).astype(np.float32)
    hT[3] = 7.0  # constant histogram -> NaN similarity, as in the reference
    FDg = ctx.fd_fpfh(hS, hT).cpu().numpy()
    FDo = oracle.fd_fpfh(hS, hT).astype(np.float32)
    np.testing.assert_array_equal(FDg, FDo)  # same sequential f32 order: bit-exact (NaN == NaN)


def test_rigid_svd(ctx, oracle):
    rng = np.random.default_rng(8)
    src = rng.normal(size=(500, 3)) * [10, 5, 2]
    R, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(R) < 0:
        R[:, 0] *= -1
    tgt = src @ R.T + [1.0, -2.0, 0.5] + 0.01 * rng.normal(size=(500, 3))
    Rg, Ro = ctx.rigid_svd(src, tgt), oracle.rigid_svd(src, tgt)
    np.testing.assert_array_equal(Rg, Ro)
    assert np.abs(Rg[:3, :3] - R).max() < 1e-3


def _fresh_context(api):
    """a context of its own: the diagnostic switches are read from the environment once, when a context is created"""
    import os

    if os.environ.get("GHICP_SIM") == "1":
        from hipsim import simctx

        return simctx.make_context(api)
    return api.Context(0)


@pytest.mark.parametrize("mode", ["default", "GHICP_KM_FORCE_HAZARD"])
def test_km_solver_paths_fuzz(api, oracle, mode):
    """The Kuhn-Munkres kernel on the fuzz generators of scripts/km4_model_fuzz.py (ties, dense rows, empty rows, ulp-perturbed
    lattices): the flood-first solver and the literal fallback behind its slack-hazard check (GHICP_KM_FORCE_HAZARD=1, the library's one
    test hook, sends a phase of every solve through it) -- bit-exact against the restatement of km.cpp:13-126."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import km4_model_fuzz as F
    rng = np.random.default_rng(77)
    sizes = [1, 2, 3, 5, 17, 64, 65, 130, 200] if mode == "GHICP_KM_FORCE_HAZARD" else [1, 2, 3, 5, 17, 64, 65, 130, 333, 700]
    if mode != "default":
        os.environ[mode] = "1"
    try:
        c = _fresh_context(api)
        for t in range(60):
            n = int(rng.choice(sizes))
            w = F.gen(rng, n, t % 5)
            np.testing.assert_array_equal(c.km_solve(w).cpu().numpy(), oracle.km(w)[0], err_msg="%s t=%d n=%d" % (mode, t, n))
        c.close()
    finally:
        os.environ.pop(mode, None)


def test_km_beyond_the_lds_resident_solver(ctx, oracle):
    """n = 3800: the solver state (44 B per row) no longer fits the 160 KB of a CU, ghicp_km_solve goes to the dense solver of km.hip
    (its only live range).  A graph the O(n^3) restatement finishes in seconds: one cheap column per row through a permutation, a few
    dozen rows competing for the same columns (failed phases + relabelling), everything else background."""
    n = 3800
    rng = np.random.default_rng(9)
    w = np.full((n, n), -8.0)
    perm = rng.permutation(n)
    w[np.arange(n), perm] = -rng.uniform(0.5, 3.0, n)
    rows = rng.choice(n, 40, replace=False)
    w[rows, perm[rng.choice(n, 40)]] = -rng.uniform(0.1, 0.4, 40)  # 40 rows prefer somebody else's column
    m_o, _ = oracle.km(w)
    m_g = ctx.km_solve(w).cpu().numpy()
    np.testing.assert_array_equal(m_g, m_o)


def test_pair_loop_persistent_batch(ctx, api, synth, oracle):
    """The persistent pair loop (loop.hip:k_pair_loop; Kuhn-Munkres batches): pairs of very different sizes in ONE batch -- several
    LDS-occupancy classes, each its own launch and queue, more pairs than one class has slots for is not needed for the logic -- against
    the oracle pair by pair (iterations, every iteration's transform, final 4x4), and the batch against the same pairs registered alone."""
    rng = np.random.default_rng(21)
    pat = synth.bsc_pattern_glibc()
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, dof=6, est_iou=0.6, voxel=0.2, pattern=pat, max_iter=25)
    p = synth.gauss_pair(n_kp=1000)
    bbx = float(oracle.bbx_magnitude(p.source))
    # (940, 930): a graph of the three-per-CU class beside the four-per-CU ones -- the launch that confines that class to its own CUs (round 5)
    shapes = [(600, 700), (150, 100), (40, 300), (940, 930), (64, 64), (0, 50), (300, 310), (2, 3)]
    clouds, refs = [], []
    for ks, kt in shapes:
        kpS = p.source[p.kp_source[:ks]].astype(np.float64)
        kpT = p.target[p.kp_target[:kt]].astype(np.float64)
        fS = rng.integers(0, 256, size=(4, ks, 56), dtype=np.uint8)
        fT = rng.integers(0, 256, size=(4, kt, 56), dtype=np.uint8)
        m = min(ks, kt)
        fT[0, :m] = fS[0, :m] ^ (rng.random((m, 56)) < 0.03).astype(np.uint8)  # matching keypoints: close strings
        S = ctx.cloud_from_features(cfg, kpS, fS, bbx)
        T = ctx.cloud_from_features(cfg, kpT, fT, bbx)
        clouds.append((S, T))
        if ks and kt:
            po = oracle.default_params(oracle.BSC, oracle.KM, 6, 0.6, 1.5, bbx, max_iter=25)
            refs.append(oracle.register(po, kpS, kpT, oracle.fd_bsc(fS, fT[0]).astype(np.float64)))
        else:
            refs.append(None)
    ctx.kernel_timing(True)
    got = ctx.register_clouds(cfg, clouds)
    stats = ctx.pair_loop_stats()
    ms, launches = ctx.kernel_time("pair_loop")
    ctx.kernel_timing(False)
    assert launches == 1 and stats["launches"] == 1 and stats["slots"] >= 7 and stats["solves"] >= sum(r["iters"] for r in refs if r), stats  # the persistent path ran
    for (ks, kt), st, r in zip(shapes, got, refs):
        if r is None:
            assert st.iterations == 0
            continue
        assert st.iterations == r["iters"], (ks, kt)
        assert st.converged == r["trace"][-1]["converged"]
        np.testing.assert_allclose(np.array(st.Rt[:]).reshape(4, 4), r["Rt"], rtol=0, atol=1e-6)
        assert st.rmse_after == pytest.approx(r["trace"][-1]["rmse_after"], rel=1e-9)
    alone = [ctx.register_clouds(cfg, [c])[0] for c in clouds]
    for a, b in zip(alone, got):
        assert (a.iterations, a.converged) == (b.iterations, b.converged)
        np.testing.assert_array_equal(np.array(a.Rt[:]), np.array(b.Rt[:]))


def test_cost_hints_change_the_queue_order_not_the_results(ctx, api, synth, oracle):
    """ghicp_ctx_set_loop_cost_hints: the persistent pair loop takes the costliest pairs of a class first; the pairs share nothing, so
    every pair's iterations and 4x4 are the same bits with, without and with adversarial hints, and a hint for another batch size is ignored."""
    if not hasattr(ctx, "set_loop_cost_hints"):
        pytest.skip("context without the hint entry")
    rng = np.random.default_rng(17)
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, dof=6, est_iou=0.6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=12)
    p = synth.gauss_pair(n_kp=400)
    bbx = float(oracle.bbx_magnitude(p.source))
    hs = []
    for ks, kt in [(120, 100), (300, 310), (64, 200), (350, 340), (90, 90), (260, 250)]:
        fS = rng.integers(0, 256, size=(4, ks, 56), dtype=np.uint8)
        fT = rng.integers(0, 256, size=(4, kt, 56), dtype=np.uint8)
        m = min(ks, kt)
        fT[0, :m] = fS[0, :m] ^ (rng.random((m, 56)) < 0.03).astype(np.uint8)
        hs.append((ctx.cloud_from_features(cfg, p.source[p.kp_source[:ks]].astype(np.float64), fS, bbx),
                   ctx.cloud_from_features(cfg, p.target[p.kp_target[:kt]].astype(np.float64), fT, bbx)))
    base = ctx.register_clouds(cfg, hs)
    assert max(st.iterations for st in base) > 1
    for hints in ([float(st.iterations) for st in base], [-float(st.iterations) for st in base], [float("nan")] * len(hs), [1.0] * (len(hs) + 1)):
        ctx.set_loop_cost_hints(hints)
        again = ctx.register_clouds(cfg, hs)
        for a, b in zip(base, again):
            assert a.iterations == b.iterations and a.converged == b.converged and list(a.Rt) == list(b.Rt)
    for a, b in hs:
        a.close()
        b.close()


@pytest.mark.parametrize("order", ["ascending", "descending", "shuffled"])
def test_one_slot_takes_many_pairs_in_any_order(api, synth, oracle, order):
    """A solve slot of the persistent pair loop keeps its LDS between pairs, and a class launch gets the LDS of the LARGEST graph of its
    class.  With ONE slot per class (GHICP_LOOP_SLOTS=1, the library's test hook) and the queue order forced by cost hints -- small graphs
    before large ones, large before small, mixed; all seven graphs in one LDS class, all beyond the stage-scratch floor of the launch --
    every pair must come out exactly as when it runs alone.  (Round 4: the first version of the hinted order sized a class's LDS by its
    FIRST problem; the solver now also refuses a problem that does not fit the LDS it was given, status 6.)"""
    import os

    os.environ["GHICP_LOOP_SLOTS"] = "1"
    try:
        c = _fresh_context(api)
    finally:
        del os.environ["GHICP_LOOP_SLOTS"]
    ref = _fresh_context(api)
    rng = np.random.default_rng(33)
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, dof=6, est_iou=0.6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=10)
    p = synth.gauss_pair(n_kp=600)
    bbx = float(oracle.bbx_magnitude(p.source))
    shapes = [(300, 280), (330, 300), (360, 360), (400, 390), (440, 430), (310, 320), (420, 400)]
    feats = []
    for ks, kt in shapes:
        fS = rng.integers(0, 256, size=(4, ks, 56), dtype=np.uint8)
        fT = rng.integers(0, 256, size=(4, kt, 56), dtype=np.uint8)
        m = min(ks, kt)
        fT[0, :m] = fS[0, :m] ^ (rng.random((m, 56)) < 0.03).astype(np.uint8)
        feats.append((p.source[p.kp_source[:ks]].astype(np.float64), p.target[p.kp_target[:kt]].astype(np.float64), fS, fT))
    mk = lambda cx: [(cx.cloud_from_features(cfg, kS, fS, bbx), cx.cloud_from_features(cfg, kT, fT, bbx)) for kS, kT, fS, fT in feats]  # noqa: E731
    hs, hr = mk(c), mk(ref)
    alone = [ref.register_clouds(cfg, [h])[0] for h in hr]  # every pair alone, on a context without the cap
    n_of = [max(ks, kt) for ks, kt in shapes]
    cost = {"ascending": [-float(n) for n in n_of], "descending": [float(n) for n in n_of], "shuffled": list(rng.permutation(len(hs)).astype(float))}[order]
    for rep in range(2):  # the second batch starts on the LDS the first one left behind
        c.set_loop_cost_hints(cost)
        got = c.register_clouds(cfg, hs)
        for a, b in zip(alone, got):
            assert a.iterations == b.iterations and a.converged == b.converged and list(a.Rt) == list(b.Rt), (order, rep)
    for a, b in hs + hr:
        a.close()
        b.close()
    c.close()
    ref.close()


def test_confined_three_per_cu_class_changes_nothing_but_the_schedule(api, synth, oracle):
    """Round 5: graphs that fit only three slots per CU run on their own CUs (masked streams, loop.hip:run_pair_loop) and the four-per-CU
    class on the others -- where a slot runs cannot reach a result: the same batch through a context with GHICP_LOOP_CONFINE=0 (one launch per
    class on every CU, as in round 4) and through a default context gives the same iterations and the same 4x4 bits, pair by pair."""
    import os

    import torch

    if os.environ.get("GHICP_SIM") != "1" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    rng = np.random.default_rng(33)
    pat = synth.bsc_pattern_glibc()
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, dof=6, est_iou=0.6, voxel=0.2, pattern=pat, max_iter=4)
    p = synth.gauss_pair(n_kp=1000)
    bbx = float(oracle.bbx_magnitude(p.source))
    shapes = [(940, 930), (600, 700), (300, 310), (950, 900), (120, 100)]
    feats = []
    for ks, kt in shapes:
        fS = rng.integers(0, 256, size=(4, ks, 56), dtype=np.uint8)
        fT = rng.integers(0, 256, size=(4, kt, 56), dtype=np.uint8)
        m = min(ks, kt)
        fT[0, :m] = fS[0, :m] ^ (rng.random((m, 56)) < 0.03).astype(np.uint8)
        feats.append((p.source[p.kp_source[:ks]].astype(np.float64), fS, p.target[p.kp_target[:kt]].astype(np.float64), fT))
    results = []
    for confine in ("0", "1"):
        os.environ["GHICP_LOOP_CONFINE"] = confine  # read once, when the context is created
        try:
            if os.environ.get("GHICP_SIM") == "1":
                from hipsim import simctx

                c = simctx.make_context(api)
            else:
                c = api.Context(0)
        finally:
            del os.environ["GHICP_LOOP_CONFINE"]
        clouds = [(c.cloud_from_features(cfg, kS, fS, bbx), c.cloud_from_features(cfg, kT, fT, bbx)) for kS, fS, kT, fT in feats]
        c.set_loop_cost_hints([float(max(ks, kt)) ** 2 for ks, kt in shapes])
        results.append([(st.iterations, st.converged, list(st.Rt)) for st in c.register_clouds(cfg, clouds)])
        for a, b in clouds:
            a.close()
            b.close()
        c.close()
    assert max(r[0] for r in results[0]) >= 2
    assert results[0] == results[1]
