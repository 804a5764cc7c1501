"""Per-cloud front-end cache: cached handles must reproduce the pair API bit for bit, survive a round
trip through the reference's SBF dump format, and serve both roles (source / target) in an all-pairs sweep."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scans(synth):
    # three stations of the same scene: pair ids share the scene (config 2), stations differ
    a = synth.tls_pair(60_000, pair_id=11)
    b = synth.tls_pair(60_000, pair_id=12)
    return [a.source, a.target, b.source]


@pytest.mark.parametrize("feature,corr", [("bsc", "km"), ("bsc", "nn"), ("fpfh", "nnr"), ("none", "nn")])
def test_cached_clouds_match_pair_api(ctx, api, synth, scans, feature, corr):
    feat = dict(bsc=api.FEATURE_BSC, fpfh=api.FEATURE_FPFH, none=api.FEATURE_NONE)[feature]
    cr = dict(km=api.CORR_KM, nn=api.CORR_NN, nnr=api.CORR_NNR)[corr]
    cfg = api.pair_config(feat, cr, dof=6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=60)
    clouds = [ctx.cloud_create(cfg, s) for s in scans]
    order = [(0, 1), (2, 1), (1, 0), (0, 2)]  # every cloud is used as a source and as a target
    got = ctx.register_clouds(cfg, [(clouds[i], clouds[j]) for i, j in order])
    ref = ctx.register_pairs(cfg, [(scans[i], scans[j]) for i, j in order])
    for g, r in zip(got, ref):
        assert (g.k_s, g.k_t, g.m_s, g.m_t, g.iterations, g.converged) == (r.k_s, r.k_t, r.m_s, r.m_t, r.iterations, r.converged)
        assert g.bbx_magnitude == r.bbx_magnitude
        np.testing.assert_array_equal(np.array(g.Rt[:]), np.array(r.Rt[:]))  # same kernels, same inputs: bit-identical
    info = clouds[0].info()
    assert info.variants == 4 and info.feature == feat and info.k == got[0].k_s


def test_sbf_dump_round_trip_and_rebuilt_handles(ctx, api, oracle, synth, scans, tmp_path):
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, dof=6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=60)
    clouds = [ctx.cloud_create(cfg, s) for s in scans[:2]]
    ref = ctx.register_clouds(cfg, [(clouds[0], clouds[1])])[0]
    rebuilt = []
    for ci, c in enumerate(clouds):
        d = c.download()
        info = c.info()
        feat = d["feat"].cpu().numpy()
        # the descriptors are what the CPU path computes on the cached down-sampled cloud
        ds, kp = d["ds"].cpu().numpy(), d["kp"].cpu().numpy()
        fo, _, _ = oracle.bsc(ds, kp, 1.5, 6, synth.bsc_pattern_glibc())
        np.testing.assert_array_equal(feat, fo[: info.variants])
        np.testing.assert_array_equal(d["kp_xyz"].cpu().numpy(), ds[kp].astype(np.float64))
        back = []
        for v in range(info.variants):  # one dump per variant vector, as the reference writes one vector<SBF> per file
            path = tmp_path / ("cloud%d_v%d.bsc" % (ci, v))
            api.sbf_write(path, feat[v])
            raw = np.fromfile(path, np.uint8)
            assert raw.size == 12 + 56 * info.k
            assert np.frombuffer(raw[:12].tobytes(), np.uint32).tolist() == [441, 56, info.k]  # size_, byte_, count
            back.append(api.sbf_read(path))
        back = np.stack(back)
        np.testing.assert_array_equal(back, feat)
        rebuilt.append(ctx.cloud_from_features(cfg, d["kp_xyz"], back, info.bbx_magnitude))
    got = ctx.register_clouds(cfg, [(rebuilt[0], rebuilt[1])])[0]
    assert got.iterations == ref.iterations
    np.testing.assert_array_equal(np.array(got.Rt[:]), np.array(ref.Rt[:]))
    with pytest.raises(api.GhicpError):  # a handle cached with another front end is refused
        other = api.pair_config(api.FEATURE_BSC, api.CORR_KM, dof=6, voxel=0.3, pattern=synth.bsc_pattern_glibc(), max_iter=60)
        ctx.register_clouds(other, [(clouds[0], clouds[1])])
    with pytest.raises(api.GhicpError):
        api.sbf_read(tmp_path / "missing.bsc")


def test_multiview_queue_on_one_gpu(ctx, api, synth, scans):
    """pairqueue.run_multiview wired to the real product calls: all ordered pairs of three scans, every front end once."""
    import importlib

    pq = importlib.import_module("gh-icp_amd.pairqueue")
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_NN, dof=6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=60)
    pairs = [(i, j) for i in range(3) for j in range(3) if i != j]
    built = []

    def make_cloud(c):
        built.append(c)
        return ctx.cloud_create(cfg, scans[c])

    def register_batch(ids, S, T):
        return [{"Rt": np.array(st.Rt[:]), "it": st.iterations} for st in ctx.register_clouds(cfg, list(zip(S, T)))]

    out = pq.run_multiview(3, pairs, make_cloud, register_batch)
    assert built == [0, 1, 2] and len(out) == 6
    ref = ctx.register_pairs(cfg, [(scans[i], scans[j]) for i, j in pairs])
    for o, r in zip(out, ref):
        assert o["it"] == r.iterations
        np.testing.assert_array_equal(o["Rt"], np.array(r.Rt[:]))
