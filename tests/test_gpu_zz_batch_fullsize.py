"""Batched front end at the size bench.py runs it (BASELINE.json configs[1]: 1 M-point scans): ghicp_clouds_recompute of several full-size
clouds == ghicp_cloud_recompute cloud by cloud, bit for bit.  (The same comparison for 32 clouds = 32 M points runs on the host SIMT
interpreter: tests/hipsim/fullsize_batch_sim.py, profiles/r02_sim_fullsize_batch.txt.)  Sorted last on purpose."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_batch_of_full_size_clouds(ctx, api, synth):
    import bench  # the config table of the benchmark

    CF = bench.CONFIGS[2]
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, CF["dof"], CF["iou"], CF["voxel"], CF["r"], CF["R"], synth.bsc_pattern_glibc(), max_iter=200)
    a = bench.make_pair(2, 0, CF["hits"])
    b = bench.make_pair(2, 1, CF["hits"])
    raws = [ctx._xyz(x) for x in (a.source, a.target, b.source, b.target, a.source[:400_000])]
    single = [ctx.cloud_create(cfg, raws[0][:0]) for _ in raws]
    batch = [ctx.cloud_create(cfg, raws[0][:0]) for _ in raws]
    for c, r in zip(single, raws):
        c.recompute(r)
    ctx.clouds_recompute(batch, raws)
    for c, d in zip(single, batch):
        ia, ib = c.info(), d.info()
        assert (ia.n, ia.m, ia.k, ia.bbx_magnitude) == (ib.n, ib.m, ib.k, ib.bbx_magnitude)
        da, db = c.download(), d.download()
        for key in ("ds", "kp", "kp_xyz", "feat"):
            np.testing.assert_array_equal(da[key].cpu().numpy(), db[key].cpu().numpy(), err_msg=key)
    i0 = batch[0].info()
    assert i0.n == CF["hits"] and 150_000 < i0.m < 500_000 and i0.k > 300


def test_batch_equals_cloud_by_cloud_fpfh(ctx, api, synth):
    """The FPFH branch of the batched front end (written on the host SIMT interpreter after the round's GPU minutes were spent): same
    comparison as tests/test_gpu_batch.py, kept in this last file until it has passed on the MI355X once."""
    import test_gpu_batch as B

    B.check_batch_equals_cloud_by_cloud(ctx, api, synth, "fpfh", 6)
