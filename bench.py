#!/usr/bin/env python
"""bench.py -- GH-ICP registration hot path on MI355X: registered pairs/sec (+ ms/iteration).

Workload = BASELINE.json configs[1]: synthetic ETH-like TLS pairs, 1 M points per scan, 0.1 m voxel,
BSC feature + KM (Kuhn-Munkres) matching, 6-DoF.  A "step" = one complete pass of the hot path
(voxel filter -> curvature keypoints -> BSC -> feature distance -> GH-ICP loop -> 4x4) over one BATCH of
`--pairs-per-step` independent pairs whose raw clouds are already resident in HBM.  Independent scan pairs
are the unit of parallelism of this problem: the per-pair KM solve is a dependency chain that
occupies one wave, so a GPU is filled by keeping many pairs in flight.  Default schedule (`--pipeline 1`): `--fe-streams`
worker contexts run the per-cloud front ends (ghicp_cloud_recompute: voxel filter, keypoints, BSC) concurrently on their
own streams, then `--loop-groups` batched loops (ghicp_register_clouds: feature distance + GH-ICP iterations, 1792 pairs
each = 7 Kuhn-Munkres solves per CU x 256 CUs, one wave per solve) run concurrently on their own streams, so that the
solve slots one group frees early are taken by the next group's launch instead of idling until the slowest solve of the
iteration ends.  Every pair's two clouds go through the full front end in the timed region (nothing is reused between
pairs or steps).  Measured alternatives: overlapping the front ends of step k+1 with the loop of step k (`--overlap 1`) is
slower -- a solve launch fills every CU's LDS and the small front-end kernels starve -- as are 2 or 4 loop groups.  `--pipeline 0` is the earlier schedule: the batch split over `--streams` contexts that
each run front ends + loop for their shard.
With N GPUs every rank registers its own batch (no data-path collective): weak scaling,
value = pairs of all ranks / max-over-ranks time.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the kernel's own stream)
and `cpu_baseline` (the oracle = PCL-free restatement of the reference path, 1 thread, rank 0, N=1 only).
"""
import argparse  # noqa: E402
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ROCm maps HIP streams onto 4 hardware queues by default; the front-end workers want one each (read at HIP initialisation)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
KERNELS = ("pca_cells", "bsc", "km_solve", "cd_rowmin", "km_weights", "fd_bsc", "nms_round", "voxel_sort")


def algorithmic_bytes(st, kernel, batch):
    """SURVEY.md §8(d) compulsory traffic of ONE launch of `kernel` (bytes); `batch` pairs per batched launch."""
    ms, mt, ks, kt = st.m_s, st.m_t, st.k_s, st.k_t
    n = max(ks, kt)
    if kernel == "pca_cells":  # S1: 16 B in + 24 B out (lambda 3xf32, curvature f64, count i32) per point; one launch per cloud
        return 0.5 * (16 + 24) * (ms + mt)
    if kernel == "bsc":  # S3: 16*M in + 56*V*K + 48*K out, per cloud (average of S: V=4 and T: V=1)
        return 0.5 * (16 * (ms + mt) + 56 * (4 * ks + kt) + 48 * (ks + kt))
    if kernel == "km_solve":  # S5 KM: the n x n f64 weight matrix must be seen at least once per pair (dense-equivalent)
        return 8.0 * n * n * batch
    if kernel == "cd_rowmin":  # S5 sweep: keypoints + u16 FD in, row minima out
        return (24.0 * (ks + kt) + 2.0 * ks * kt + 12.0 * ks) * batch
    if kernel == "km_weights":  # CSR build: FD read twice (count + fill), <= 12 B per explicit entry written
        return (24.0 * (ks + kt) + 4.0 * ks * kt) * batch
    if kernel == "fd_bsc":
        return 56.0 * (4 * ks + kt) + 2.0 * ks * kt
    if kernel == "nms_round":  # S2: 20 B per down-sampled point in (xyz + curvature), 4 B per keypoint out; one launch per cloud
        return 0.5 * (20.0 * (ms + mt) + 4.0 * (ks + kt))
    if kernel == "voxel_sort":  # S0: 16 B per raw point in + 16 B per kept point out; one launch per cloud
        return 0.5 * (16.0 * (st.n_s + st.n_t) + 16.0 * (ms + mt))
    return float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hits", type=int, default=1_000_000, help="points per scan (cfg2 = 1M)")
    ap.add_argument("--corr", default="KM", choices=["KM", "NN", "NNR"])
    ap.add_argument("--pairs-per-step", type=int, default=5376, help="independent pairs per GPU per step (3 x 1792; 1792 = 7 KM solves per CU x 256 CUs)")
    ap.add_argument("--pipeline", type=int, default=1, help="1: front-end worker streams + one batched loop (default); 0: per-stream register_pairs")
    ap.add_argument("--fe-streams", type=int, default=16, help="front-end worker contexts/streams (pipeline mode)")
    ap.add_argument("--loop-groups", type=int, default=3, help="pipeline mode: the step's pairs are registered by this many concurrent batched loops")
    ap.add_argument("--reserve-cus", type=int, default=0, help="pipeline mode: CUs (multiple of 8) kept free of loop kernels via a CU-masked loop stream")
    ap.add_argument("--overlap", type=int, default=0, help="pipeline mode: run the front ends of step k+1 during the loop of step k")
    ap.add_argument("--streams", type=int, default=4, help="--pipeline 0: contexts/streams the batch is split over")
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic pairs generated per rank (cycled inside the batch)")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 disables the CPU (oracle) baseline leg")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the GH-ICP hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    api = importlib.import_module("gh-icp_amd.api")
    synth = importlib.import_module("gh-icp_amd.synth")

    # ---- synthetic cfg2 pairs for this rank (pair ids are unique across ranks: independent scenes)
    t0 = time.time()
    pairs = [synth.tls_pair(args.hits, config_id=2, pair_id=rank * args.distinct + i) for i in range(args.distinct)]
    gen_s = time.time() - t0
    B = max(1, args.pairs_per_step)
    dev = [(torch.from_numpy(p.source).cuda(), torch.from_numpy(p.target).cuda()) for p in pairs]
    torch.cuda.synchronize()
    corr = {"KM": api.CORR_KM, "NN": api.CORR_NN, "NNR": api.CORR_NNR}[args.corr]
    cfg = api.pair_config(api.FEATURE_BSC, corr, 6, 0.6, 0.1, 0.5, 1.5, synth.bsc_pattern_glibc(), max_iter=200)
    phase_s = {"front_end": 0.0, "loop": 0.0}
    if args.pipeline:
        nstream = max(1, min(args.fe_streams, B))
        G = max(1, min(args.loop_groups, B))
        streams = [torch.cuda.Stream() for _ in range(nstream + G)]
        ctxs = [api.Context(local_rank, stream=s) for s in streams]
        fe_ctxs, loop_ctxs = ctxs[:nstream], ctxs[nstream:]
        if args.reserve_cus > 0:
            # 4 bits at the start of every 32-CU word: no XCD loses all its CUs whichever way the mask bits map onto XCDs
            per = max(1, args.reserve_cus // 8)
            word = (0xFFFFFFFF << per) & 0xFFFFFFFF
            for c in loop_ctxs:
                c.set_cu_mask([word] * 8)
        NB = 2 if args.overlap else 1
        pools = [[None] * B for _ in range(NB)]  # (source, target) cloud handles; two sets when steps overlap (double buffering)
        results = [None] * G

        def fe_worker(w, buf):
            pool, c = pools[buf], fe_ctxs[w]
            for i in range(w, B, nstream):
                S, T = dev[i % len(dev)]
                if pool[i] is None:
                    pool[i] = (c.cloud_create(cfg, S), c.cloud_create(cfg, T))
                else:
                    pool[i][0].recompute(S)
                    pool[i][1].recompute(T)

        def front_ends(buf):
            t = time.perf_counter()
            th = [threading.Thread(target=fe_worker, args=(w, buf)) for w in range(nstream)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            phase_s["front_end"] += time.perf_counter() - t

        def loop_group(g, buf):
            results[g] = loop_ctxs[g].register_clouds(cfg, pools[buf][g * B // G:(g + 1) * B // G])

        def loop(buf):
            t = time.perf_counter()
            th = [threading.Thread(target=loop_group, args=(g, buf)) for g in range(G)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            phase_s["loop"] += time.perf_counter() - t

        def run_steps(nsteps):
            if nsteps <= 0:
                return
            front_ends(0)
            for k in range(nsteps):
                if args.overlap and k + 1 < nsteps:
                    tl = threading.Thread(target=loop, args=(k % NB,))
                    tl.start()
                    front_ends((k + 1) % NB)
                    tl.join()
                else:
                    loop(k % NB)
                    if k + 1 < nsteps:
                        front_ends((k + 1) % NB)
        shard_b = B // G
    else:
        nstream = max(1, min(args.streams, B))
        streams = [torch.cuda.Stream() for _ in range(nstream)]
        ctxs = [api.Context(local_rank, stream=s) for s in streams]
        shards = [[dev[(i * nstream + s) % len(dev)] for i in range((B - s + nstream - 1) // nstream)] for s in range(nstream)]
        results = [None] * nstream

        def run_shard(s, nsteps):
            # every stream works through its shard of each step back to back; the streams are NOT re-synchronised between
            # steps, so one stream's front end overlaps another stream's KM-bound loop
            for _ in range(nsteps):
                results[s] = ctxs[s].register_pairs(cfg, shards[s])

        def run_steps(nsteps):
            if nsteps <= 0:
                return
            th = [threading.Thread(target=run_shard, args=(s, nsteps)) for s in range(nstream)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        shard_b = len(shards[0])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    if args.pipeline and len(pools) > 1 and pools[1][0] is None:
        front_ends(1)  # allocate the second handle set outside the timed region
    for c in ctxs:
        c.kernel_timing(True)
    phase_s["front_end"] = phase_s["loop"] = 0.0
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ktimes = {}
    for k in KERNELS:
        ms = sum(c.kernel_time(k)[0] for c in ctxs)
        nl = sum(c.kernel_time(k)[1] for c in ctxs)
        ktimes[k] = (ms, nl)
    for c in ctxs:
        c.kernel_timing(False)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    stats = results[0][0]
    pairs_total = args.steps * B * world
    value = pairs_total / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    Rg = np.array(stats.Rt[:]).reshape(4, 4)
    iters = max(1, stats.iterations)

    # ---- roofline of the dominant kernel (largest share of HIP-event kernel time over the timed region)
    dom = max(ktimes, key=lambda k: ktimes[k][0])
    dom_ms, dom_n = ktimes[dom]
    avg_ms = dom_ms / max(1, dom_n)
    b_alg = algorithmic_bytes(stats, dom, shard_b)
    achieved = b_alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    per_kernel = {}
    for k, v in ktimes.items():
        ba = algorithmic_bytes(stats, k, shard_b)
        per_kernel[k] = {"ms_total": round(v[0], 3), "launches": v[1],
                         "GBps": round(ba / (v[0] / max(1, v[1]) * 1e-3) / 1e9, 2) if v[0] > 0 and ba == ba else None}
    # HBM-side traffic of the dominant kernel from the committed PMC passes (counters cannot be collected inside this run)
    traffic, traffic_note = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if dom in pmc:
            units = shard_b if pmc[dom]["per"] in ("solve", "pair") else 1
            traffic = int(pmc[dom]["bytes"] * units)
            traffic_note = "%d B per %s x %d (%s)" % (pmc[dom]["bytes"], pmc[dom]["per"], units, pmc["source"])
    except (OSError, ValueError, KeyError):
        pass
    roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_note": traffic_note, "avg_launch_ms": round(avg_ms, 4), "launches": dom_n,
                "alg_bytes_per_launch": int(b_alg),
                "note": ("km_solve is a dependency chain (exact emulation of the reference's DFS order), latency- not bandwidth-bound; "
                         "%d solves run concurrently per launch" % shard_b) if dom == "km_solve" else
                        ("%s has the largest summed HIP-event time over all streams (launches of different streams overlap in wall time)" % dom),
                "per_kernel": per_kernel}

    # ---- CPU baseline: the oracle (faithful PCL-free restatement of the reference path), 1 thread
    cpu = None
    check = None
    workload_stats = None
    if world == 1 and args.cpu_baseline:
        from oracle import oracle as O  # checker / baseline only

        pair = pairs[0]
        tc = time.perf_counter()
        ds, kp, feat, kbar, mbar = {}, {}, {}, {}, {}
        for name, cloud, dof in (("T", pair.target, 0), ("S", pair.source, 6)):
            keep = O.voxel_filter(cloud, 0.1)
            ds[name] = cloud[keep]
            kp[name], kbar[name] = O.keypoints(ds[name], 0.5, 1.5)  # + mean neighbours in the PCA radius (k-bar of SURVEY.md §8)
            feat[name], _, mbar[name] = O.bsc(ds[name], kp[name], 1.5, dof, synth.bsc_pattern_glibc())  # + mean points per BSC sphere (m-bar)
        FD = O.fd_bsc(feat["S"], feat["T"][0])
        t_front = time.perf_counter() - tc
        P = O.default_params(O.BSC, {"KM": O.KM, "NN": O.NN, "NNR": O.NNR}[args.corr], 6, 0.6, 1.5, O.bbx_magnitude(ds["S"]))
        tl = time.perf_counter()
        ro = O.register(P, ds["S"][kp["S"]].astype(np.float64), ds["T"][kp["T"]].astype(np.float64), FD)
        t_loop = time.perf_counter() - tl
        t_pair = t_front + t_loop
        cpu = {"value": round(1.0 / t_pair, 5), "unit": "pairs/s", "cores": 1, "kind": "port",
               "sample": "ONE complete cfg2 pair (pair 0 of the batch): front end + FD %.1f s, %d loop iterations %.1f s; oracle = PCL-free "
                         "restatement of the reference path (the reference needs PCL/Eigen/FLANN, not installable here), g++ -O2, 1 thread; "
                         "host has %d logical CPUs" % (t_front, ro["iters"], t_loop, os.cpu_count())}
        workload_stats = {"k_bar": round(0.5 * (float(kbar["S"]) + float(kbar["T"])), 1), "m_bar": round(0.5 * (float(mbar["S"]) + float(mbar["T"])), 1)}
        check = {"iterations_match": int(stats.iterations) == int(ro["iters"]),
                 "keypoints_match": (int(stats.k_s), int(stats.k_t)) == (int(kp["S"].size), int(kp["T"].size)),
                 "rot_err_vs_oracle": round(synth.rot_err(Rg, ro["Rt"]), 9), "trans_err_vs_oracle_m": round(synth.trans_err(Rg, ro["Rt"]), 9)}

    out = {
        "metric": "registered_pairs_per_sec", "value": round(value, 4), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "cfg2: synthetic ETH-like TLS pairs, %d pts/scan, voxel 0.1 m, r_pca 0.5, R_nms 1.5, BSC + %s, 6-DoF; "
                               "%d independent pairs in flight per GPU per step (%d distinct scenes cycled), %s"
                               % (args.hits, args.corr, B, args.distinct,
                                  ("%d front-end streams, then %d concurrent batched loop(s)%s" % (nstream, args.loop_groups, ", front ends of step k+1 overlap the loop of step k" if args.overlap else ""))
                                  if args.pipeline else "%d streams" % nstream),
                   "pairs_per_step": B, "n_s": int(stats.n_s), "m_s": int(stats.m_s), "m_t": int(stats.m_t), "k_s": int(stats.k_s),
                   "k_t": int(stats.k_t), "iterations": int(stats.iterations), "parallelism": "pairs sharded over ranks, no data-path collective"},
        "ms_per_iteration": round(ms_per_step / iters, 4),
        "ms_per_iteration_note": "wall time of one batched step / iterations of pair 0: every in-flight pair advances one iteration in that time",
        "batch_ms": ({"front_end_wall_per_step": round(1e3 * phase_s["front_end"] / max(1, args.steps), 1),
                      "loop_wall_per_step": round(1e3 * phase_s["loop"] / max(1, args.steps), 1),
                      "fd_per_pair": round(stats.ms_fd, 3), "loop_per_pair": round(stats.ms_loop, 3)} if args.pipeline else
                     {"front_end_per_pair": round(stats.ms_keypoints, 3), "loop_per_pair": round(stats.ms_loop, 3)}),
        "gt_error": {"rot": round(synth.rot_err(Rg, pairs[0].gt), 6), "trans_m": round(synth.trans_err(Rg, pairs[0].gt), 5)},
        "roofline": roofline, "cpu_baseline": cpu, "parity_check": check, "gen_seconds": round(gen_s, 1),
    }
    if workload_stats:
        out["config"].update(workload_stats)  # measured on the CPU leg: mean neighbours per PCA query / points per BSC sphere
    if cpu:
        out["speedup_vs_cpu_1thread"] = round(value / cpu["value"], 2)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
