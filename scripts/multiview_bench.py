"""All-pairs registration of N clouds: cached per-cloud front ends (ghicp_cloud_create + ghicp_register_clouds) against the
pair API (ghicp_register_pairs, which runs both front ends for every pair).  Prints one JSON line."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=32)
    ap.add_argument("--hits", type=int, default=1_000_000)
    a = ap.parse_args()
    import torch

    api = importlib.import_module("gh-icp_amd.api")
    synth = importlib.import_module("gh-icp_amd.synth")
    ctx = api.Context(0)
    base = []
    for pid in (0, 1):
        p = synth.tls_pair(a.hits, config_id=2, pair_id=pid)
        base += [torch.from_numpy(p.source).cuda(), torch.from_numpy(p.target).cuda()]
    clouds = [base[i % len(base)] for i in range(a.clouds)]
    pairs = [(i, j) for i in range(a.clouds) for j in range(a.clouds) if i != j]
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, 6, 0.6, 0.1, 0.5, 1.5, synth.bsc_pattern_glibc(), max_iter=200)
    ctx.register_clouds(cfg, [(ctx.cloud_create(cfg, clouds[0]), ctx.cloud_create(cfg, clouds[1]))])  # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    handles = [ctx.cloud_create(cfg, c) for c in clouds]
    t1 = time.perf_counter()
    cached = ctx.register_clouds(cfg, [(handles[i], handles[j]) for i, j in pairs])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ref = ctx.register_pairs(cfg, [(clouds[i], clouds[j]) for i, j in pairs])
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    same = all(np.array_equal(np.array(x.Rt[:]), np.array(y.Rt[:])) and x.iterations == y.iterations for x, y in zip(cached, ref))
    print(json.dumps(dict(clouds=a.clouds, pairs=len(pairs), front_ends_cached_s=round(t1 - t0, 3), loops_cached_s=round(t2 - t1, 3),
                          cached_total_s=round(t2 - t0, 3), pair_api_total_s=round(t3 - t2, 3), speedup=round((t3 - t2) / (t2 - t0), 3),
                          identical_results=bool(same))))


if __name__ == "__main__":
    main()
