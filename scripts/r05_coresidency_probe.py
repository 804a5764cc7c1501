#!/usr/bin/env python
"""Does a small kernel make progress on CUs that hold four resident solve slots?  (NEXT_ROUND.md, lever 1: front end BESIDE the slots.)

Thread A registers a batch of cfg2 pairs whose graphs put exactly four LDS-limited slots on a CU (n in (740, 930]: 33-40 KB each, 12-28 KB of LDS
left) -- enough pairs that every slot of the chip stays busy for seconds.  Thread B, on its own context and stream, times two small kernels
before the batch (alone) and while the batch is dense (beside): ghicp_transform_cloud (no LDS, ~20 VGPRs) and ghicp_fd_fpfh (8.5 KB of LDS per
workgroup, 256 threads).  With the shipped k_pair_loop (128 VGPRs x 4 waves x 4 slots = every VGPR of the CU) the expectation is that B's
kernels wait for a slot to leave; with the 96-VGPR build of branch next/pair-loop-96vgpr (GHICP_LIB=...) 128 VGPRs per SIMD lane are free and
the question is whether the dispatcher back-fills them.  Prints one JSON line: median / max call times alone and beside, the slots busy meanwhile.

    python scripts/r05_coresidency_probe.py                       # shipped library
    GHICP_LIB=gh-icp_amd/libghicp_var_occ5.so python scripts/r05_coresidency_probe.py   # 96-VGPR build (scripts/km_variant_lib.sh next/pair-loop-96vgpr occ5)
"""
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse

    import torch

    ap = argparse.ArgumentParser()
    ap.add_argument("--hits", type=int, default=0, help="points per cloud (default: the bench configuration's); small values are for the dry run on the interpreter")
    ap.add_argument("--n-range", type=int, nargs=2, default=(740, 930), help="graph sizes of the four-per-CU LDS class")
    ap.add_argument("--pairs", type=int, default=2600, help="pairs in the batch (~2.5 per slot of the chip)")
    ap.add_argument("--probe-points", type=int, default=1_000_000)
    ap.add_argument("--full-above", type=int, default=1100, help="the slots count as full while more pairs than this are unfinished")
    args = ap.parse_args()

    import bench

    api = importlib.import_module("gh-icp_amd.api")
    synth = importlib.import_module("gh-icp_amd.synth")
    CF = bench.CONFIGS[2]
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, CF["dof"], CF["iou"], CF["voxel"], CF["r"], CF["R"], synth.bsc_pattern_glibc(), max_iter=200)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    A, B = api.Context(0, stream=sa), api.Context(0, stream=sb)
    # scenes whose Kuhn-Munkres graphs are in the four-per-CU LDS class
    picked = []
    # (scene ids with 740 < max(K_S, K_T) <= 930 and ordinary iteration counts, from profiles/r04_bench_default_detail.json: no search on the GPU box)
    for sid in (0, 5, 10, 14, 16, 18, 19, 21, 24, 26, 27, 28):
        p = bench.make_pair(2, sid, args.hits or CF["hits"])
        S, T = A.cloud_create(cfg, torch.from_numpy(p.source).cuda()), A.cloud_create(cfg, torch.from_numpy(p.target).cuda())
        n = max(S.info().k, T.info().k)
        if args.n_range[0] < n <= args.n_range[1]:
            picked.append((S, T, n))
        else:
            S.close()
            T.close()
        if len(picked) >= 12:
            break
    if not picked:
        raise SystemExit("no scene in the four-per-CU class among the first 64")
    pairs = [(S, T) for S, T, _ in picked] * max(1, args.pairs // len(picked))  # ~2.5 pairs per slot: several seconds of full slots
    cloud = torch.from_numpy(np.random.default_rng(1).standard_normal((args.probe_points, 3)).astype(np.float32)).cuda()
    hS = torch.from_numpy(np.random.default_rng(2).random((4096, 33)).astype(np.float32)).cuda()
    Rt = np.eye(4)

    def probe(seconds, stop=None):
        out = {"transform_ms": [], "fd_fpfh_ms": []}
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end and not (stop and stop.is_set()):
            for name, fn in (("transform_ms", lambda: B.transform_cloud(cloud, Rt)), ("fd_fpfh_ms", lambda: B.fd_fpfh(hS, hS))):
                t = time.perf_counter()
                fn()
                B.sync()
                out[name].append(1e3 * (time.perf_counter() - t))
        return out

    probe(0.5)
    alone = probe(1.5)
    done = threading.Event()
    res = {}

    def run_batch():
        A.kernel_timing(True)
        t = time.perf_counter()
        res["stats"] = A.register_clouds(cfg, pairs)
        res["batch_s"] = time.perf_counter() - t
        res["loop"] = A.pair_loop_stats()
        done.set()

    th = threading.Thread(target=run_batch)
    th.start()
    time.sleep(0.8)  # feature distances + hand-over: the slots are full from here on
    beside = {"transform_ms": [], "fd_fpfh_ms": [], "active_pairs": []}
    while not done.is_set():
        a, tot = A.loop_progress()
        if tot == 0:  # feature distances / hand-over still running: the loop has not started
            time.sleep(0.05)
            continue
        if a <= args.full_above:  # the queue is about to drain: no longer "every slot busy"
            break
        part = probe(0.25, done)
        for k in ("transform_ms", "fd_fpfh_ms"):
            beside[k] += part[k]
        beside["active_pairs"].append(int(a))
    th.join()
    summ = lambda v: {"n": len(v), "median": round(float(np.median(v)), 3), "max": round(float(np.max(v)), 3)} if v else None  # noqa: E731
    print(json.dumps({"library": api.LIB_PATH, "pairs": len(pairs), "n_of_scenes": [n for _, _, n in picked], "batch_s": round(res["batch_s"], 2),
                      "loop": res["loop"], "alone": {k: summ(v) for k, v in alone.items()},
                      "beside_full_slots": {k: summ(beside[k]) for k in ("transform_ms", "fd_fpfh_ms")}, "active_pairs_seen": beside["active_pairs"][:40]}))


if __name__ == "__main__":
    main()
