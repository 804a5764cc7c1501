"""One rank of a ghicp_pairqueue job on the GPU (test helper, started by tests/test_gpu_multirank.py): registers N small scan pairs through
ghicp_pairqueue_register_pairs -- its share of them -- and writes the gathered records of ALL pairs as JSON.
usage: pq_native_worker.py <rendezvous> <rank> <world> <transport 0|1> <chunk> <n_pairs> <out.json>"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    path, rank, world, transport, chunk, n_pairs, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    import torch

    api = importlib.import_module("gh-icp_amd.api")
    pq = importlib.import_module("gh-icp_amd.pairqueue")
    synth = importlib.import_module("gh-icp_amd.synth")
    if os.environ.get("GHICP_SIM") == "1":  # development aid: the same worker on the host SIMT interpreter (tests/hipsim), host transport only
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from hipsim import simctx

        ctx = simctx.make_context(api)
    else:
        ctx = api.Context(0)
    q = pq.NativeQueue(path, rank, world, transport, ctx, timeout_s=240.0)
    # the manifest travels from rank 0 (ncclBroadcast / the segment): pair p registers scene manifest[p]
    manifest = q.broadcast_manifest([(3 * p + 1) % 7 for p in range(n_pairs)] if rank == 0 else [])
    pairs = {sid: synth.tls_pair(int(os.environ.get("GHICP_PQ_HITS", "60000")), pair_id=sid) for sid in sorted(set(manifest))}
    S = [torch.from_numpy(pairs[sid].source).to(ctx.dev) for sid in manifest]
    T = [torch.from_numpy(pairs[sid].target).to(ctx.dev) for sid in manifest]
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, 6, 0.6, 0.2, 0.5, 1.5, synth.bsc_pattern_glibc(), max_iter=60)
    rec = q.register_pairs(cfg, S, T, chunk=chunk)
    extra = {}
    if chunk == 0:  # the primitives on their own: one more gather of a hand-made block
        blk = pq.pack_records(q.static_share(5), [(p, 1, [float(p)] * 16) for p in q.static_share(5)], pq.records_per_rank(5, world))
        extra["gathered_ids"] = sorted(q.gather_records(blk))
    q.close()
    ctx.close()
    json.dump({"rank": rank, "manifest": manifest, "records": rec.tolist(), **extra}, open(out, "w"))


if __name__ == "__main__":
    main()
