"""Batched front end (gh-icp_amd/csrc/batch.hip): ghicp_clouds_recompute must reproduce ghicp_cloud_recompute cloud by cloud, bit for
bit -- down-sampled points, keypoint ids and coordinates, BSC strings -- for batches of different clouds, including empty ones, and the
registrations that follow must be identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    ia, ib = a.info(), b.info()
    assert (ia.n, ia.m, ia.k, ia.variants, ia.feature) == (ib.n, ib.m, ib.k, ib.variants, ib.feature)
    assert ia.bbx_magnitude == ib.bbx_magnitude
    da, db = a.download(), b.download()
    for key in ("ds", "kp", "kp_xyz", "feat"):
        if da[key] is None or db[key] is None:
            assert da[key] is None and db[key] is None, key
        else:
            np.testing.assert_array_equal(da[key].cpu().numpy(), db[key].cpu().numpy(), err_msg=key)


@pytest.mark.parametrize("feature,dof", [("bsc", 6), ("bsc", 4), ("none", 6), ("fpfh", 6)])
def test_batch_equals_cloud_by_cloud(ctx, api, synth, feature, dof):
    check_batch_equals_cloud_by_cloud(ctx, api, synth, feature, dof)


def check_batch_equals_cloud_by_cloud(ctx, api, synth, feature, dof):
    feat = dict(bsc=api.FEATURE_BSC, none=api.FEATURE_NONE, fpfh=api.FEATURE_FPFH)[feature]
    cfg = api.pair_config(feat, api.CORR_NNR if feature == "fpfh" else api.CORR_NN, dof=dof, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=40)
    a = synth.tls_pair(40_000, pair_id=21)
    b = synth.tls_pair(25_000, pair_id=22)
    g = synth.gauss_pair(3000)
    raws = [a.source, a.target, b.source, g.source, b.target[:7000], a.source[:1]]
    seed = a.source[:500]
    single = [ctx.cloud_create(cfg, seed) for _ in raws]
    batch = [ctx.cloud_create(cfg, seed) for _ in raws]
    for c, r in zip(single, raws):
        c.recompute(r)
    ctx.clouds_recompute(batch, raws)
    for c, d in zip(single, batch):
        _same(c, d)
    assert batch[0].info().k > 10 and batch[2].info().k > 10  # the comparison is not vacuous
    # a second batch into the same handles, other sizes, other order: buffers are reused
    order = [4, 2, 0, 1]
    ctx.clouds_recompute([batch[i] for i in order], [raws[(i + 1) % len(raws)] for i in order])
    for i in order:
        single[i].recompute(raws[(i + 1) % len(raws)])
        _same(single[i], batch[i])
    # batch of one
    ctx.clouds_recompute([batch[3]], [raws[0]])
    single[3].recompute(raws[0])
    _same(single[3], batch[3])
    if feature in ("bsc", "fpfh"):  # and the registrations from the batched handles are the registrations from the others
        pairs = [(0, 1), (2, 1)]
        got = ctx.register_clouds(cfg, [(batch[i], batch[j]) for i, j in pairs])
        ref = ctx.register_clouds(cfg, [(single[i], single[j]) for i, j in pairs])
        for x, y in zip(got, ref):
            assert (x.iterations, x.converged, x.k_s, x.k_t) == (y.iterations, y.converged, y.k_s, y.k_t)
            np.testing.assert_array_equal(np.array(x.Rt[:]), np.array(y.Rt[:]))


def test_batch_matches_oracle_keypoints(ctx, api, synth, oracle):
    """and not only itself: keypoints of every cloud of a batch against the CPU restatement of the reference"""
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_NN, dof=6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=40)
    p = synth.tls_pair(30_000, pair_id=23)
    raws = [p.source, p.target, p.source[:12_000]]
    clouds = [ctx.cloud_create(cfg, p.source[:100]) for _ in raws]
    ctx.clouds_recompute(clouds, raws)
    for c, raw in zip(clouds, raws):
        keep = oracle.voxel_filter(raw, 0.2)
        ds = raw[keep]
        kp, _ = oracle.keypoints(ds, cfg.neighborhood_radius, cfg.reg.radius_nonmax, cfg.ratio_max, cfg.min_neighbors)
        d = c.download()
        np.testing.assert_array_equal(d["ds"].cpu().numpy(), ds)
        np.testing.assert_array_equal(d["kp"].cpu().numpy(), kp)
        f, _, _ = oracle.bsc(ds, kp, cfg.reg.radius_nonmax, 6, synth.bsc_pattern_glibc())
        got = d["feat"].cpu().numpy()
        assert got.shape == f.shape
        bad = np.flatnonzero((got != f).any(axis=(0, 2)))
        assert bad.size <= max(1, kp.size // 100), bad.size  # N2/N4 of the numerics contract: a bit on < 1 % of the keypoints


def test_batch_edge_cases(ctx, api, synth):
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_NN, dof=6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=40)
    p = synth.tls_pair(20_000, pair_id=24)
    empty = np.zeros((0, 3), np.float32)
    one = p.source[:1]
    flat = np.tile(p.source[:1], (50, 1))  # one voxel
    raws = [empty, one, flat, p.source, empty]
    single = [ctx.cloud_create(cfg, p.source[:300]) for _ in raws]
    batch = [ctx.cloud_create(cfg, p.source[:300]) for _ in raws]
    for c, r in zip(single, raws):
        c.recompute(r)
    ctx.clouds_recompute(batch, raws)
    for c, d in zip(single, batch):
        _same(c, d)
    ctx.clouds_recompute([], [])
    ctx.clouds_recompute([batch[0], batch[4]], [empty, empty])
    assert batch[0].info().m == 0 and batch[4].info().k == 0
    with pytest.raises(api.GhicpError):
        ctx.clouds_recompute([batch[0], batch[0]], [one, one])  # the same handle twice
    other = api.pair_config(api.FEATURE_BSC, api.CORR_NN, dof=6, voxel=0.3, pattern=synth.bsc_pattern_glibc(), max_iter=40)
    c2 = ctx.cloud_create(other, p.source[:300])
    with pytest.raises(api.GhicpError):
        ctx.clouds_recompute([batch[0], c2], [one, one])  # two front-end configurations


def test_batch_large_extent_splits_instead_of_failing(ctx, api, synth):
    """Round-2 advisor finding: the cell tables of a batch are summed into one, and clouds of large extent (tens of millions of PCA
    cells each) used to return GHICP_ERR_CAPACITY where the cloud-by-cloud path works.  The batch now halves itself."""
    rng = np.random.default_rng(5)
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_NN, dof=6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=40)
    dense = synth.tls_pair(20_000, pair_id=25).source
    raws = []
    for i in range(6):  # 20 k scan points + a sprinkle over a 300 m cube: ~2^26 cells of 0.5 m per cloud, 6 clouds > the 2^28 budget
        far = (rng.random((400, 3), dtype=np.float32) - 0.5) * np.float32(300.0)
        raws.append(np.ascontiguousarray(np.concatenate([dense[i::6], far]).astype(np.float32)))
    single = [ctx.cloud_create(cfg, dense[:300]) for _ in raws]
    batch = [ctx.cloud_create(cfg, dense[:300]) for _ in raws]
    for c, r in zip(single, raws):
        c.recompute(r)
    ctx.clouds_recompute(batch, raws)
    for c, d in zip(single, batch):
        _same(c, d)
    for c in single + batch:
        c.close()
