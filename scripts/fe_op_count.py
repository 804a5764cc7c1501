"""Host operations of the front end for 16 clouds, cloud by cloud against one batched launch sequence, counted by the host SIMT interpreter
(tests/hipsim; DESIGN.md §4c quotes these).  Development aid:   GHICP_SIM=1 python scripts/fe_op_count.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GHICP_SIM"] = "1"
from hipsim import simctx  # noqa: E402

api = importlib.import_module("gh-icp_amd.api")
synth = importlib.import_module("gh-icp_amd.synth")
ctx = simctx.make_context(api)
cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, dof=6, voxel=0.2, pattern=synth.bsc_pattern_glibc(), max_iter=40)
raws = []
for i in range(8):
    q = synth.tls_pair(30_000, pair_id=i)
    raws += [q.source, q.target]
clouds = [ctx.cloud_create(cfg, raws[0][:0]) for _ in raws]


def delta(f):
    a = simctx.counters(ctx.lib)
    f()
    b = simctx.counters(ctx.lib)
    return {k: b[k] - a[k] for k in ("launches", "library_calls", "memcpys", "memsets", "host_syncs")}


def one_by_one():
    for c, r in zip(clouds, raws):
        c.recompute(r)


one_by_one()  # buffers allocated
ctx.clouds_recompute(clouds, raws)
print("cloud by cloud:", delta(one_by_one))
print("batched       :", delta(lambda: ctx.clouds_recompute(clouds, raws)))
