"""TEST INFRASTRUCTURE: a full-size batch (32 cfg2 clouds of 1 M points = 32 M raw points through ONE ghicp_clouds_recompute) against the cloud-by-cloud
path on the host SIMT interpreter (tests/hipsim) -- checks the batch's index arithmetic at the size bench.py runs it; no GPU needed (~3 min)."""
import sys, importlib, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
api = importlib.import_module("gh-icp_amd.api")
synth = importlib.import_module("gh-icp_amd.synth")
from hipsim import simctx
ctx = simctx.make_context(api)
t=time.time()
pairs = [synth.tls_pair(1_000_000, pair_id=i) for i in range(4)]
print("gen", time.time()-t, flush=True)
cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, dof=6, voxel=0.1, pattern=synth.bsc_pattern_glibc(), max_iter=200)
base = [x for p in pairs for x in (p.source, p.target)]
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
raws = [base[i % len(base)] for i in range(NB)]
batch = [ctx.cloud_create(cfg, base[0][:0]) for _ in raws]
t=time.time()
ctx.clouds_recompute(batch, raws)
print("batch of %d x 1M" % NB, time.time()-t, flush=True)
single = [ctx.cloud_create(cfg, base[0][:0]) for _ in base]
for c, r in zip(single, base): c.recompute(r)
for i, d in enumerate(batch):
    c = single[i % len(base)]
    ia, ib = c.info(), d.info()
    assert (ia.n, ia.m, ia.k, ia.bbx_magnitude) == (ib.n, ib.m, ib.k, ib.bbx_magnitude), i
    da, db = c.download(), d.download()
    for key in ("ds", "kp", "kp_xyz", "feat"):
        assert np.array_equal(da[key].cpu().numpy(), db[key].cpu().numpy()), (i, key)
print("%d x 1M BATCH == SINGLE" % NB, [b.info().k for b in batch[:8]], simctx.counters(ctx.lib))
