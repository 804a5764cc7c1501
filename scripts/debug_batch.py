"""Batched loop (ghicp_register_clouds over cached front ends, mixed scenes) against the single-pair API on the same scenes:
iterations and 4x4 must be identical -- through cloud_create, through cloud_recompute, and under bench.py's schedule
(16 front-end threads, 3 concurrent loop groups).  python scripts/debug_batch.py FIRST COUNT [BATCH]"""
import importlib, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    import multiprocessing as mp
    import bench
    first, count = int(sys.argv[1]), int(sys.argv[2])
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1792
    with mp.get_context("fork").Pool(min(count, 32)) as pool:
        gen = pool.map(bench._gen_worker, [(2, sid, 1_000_000) for sid in range(first, first + count)])
    import torch
    api = importlib.import_module("gh-icp_amd.api")
    synth = importlib.import_module("gh-icp_amd.synth")
    ctx = api.Context(0)
    cfg = api.pair_config(api.FEATURE_BSC, api.CORR_KM, 6, 0.6, 0.1, 0.5, 1.5, synth.bsc_pattern_glibc(), max_iter=200)
    dev = [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()) for s, t, _ in gen]
    single = []
    for i, (S, T) in enumerate(dev):
        st, _ = ctx.register_pair(cfg, S, T, want_trace=False)
        single.append((st.iterations, np.array(st.Rt[:]), st.k_s, st.k_t))
    print("single-pair results:", [(s[0], max(s[2], s[3])) for s in single], flush=True)

    def compare(tag, res, idx):
        bad = 0
        for st, i in zip(res, idx):
            it0, R0, ks, kt = single[i]
            if not (st.iterations == it0 and np.array_equal(np.array(st.Rt[:]), R0) and (st.k_s, st.k_t) == (ks, kt)):
                bad += 1
                if bad <= 6:
                    print("  MISMATCH", tag, "scene", first + i, "iterations", st.iterations, "vs", it0, "k", (st.k_s, st.k_t), "vs", (ks, kt),
                          "max |dRt| %.3e" % float(np.abs(np.array(st.Rt[:]) - R0).max()), flush=True)
        print(tag, ": items", len(res), "mismatches", bad, flush=True)

    handles = [(ctx.cloud_create(cfg, S), ctx.cloud_create(cfg, T)) for S, T in dev]
    compare("A create, 1 context", ctx.register_clouds(cfg, handles), range(count))
    for (hs, ht), (S, T) in zip(handles, dev):
        hs.recompute(S); ht.recompute(T)
    compare("B recompute, 1 context", ctx.register_clouds(cfg, handles), range(count))
    compare("B2 batch x8, 1 context", ctx.register_clouds(cfg, [handles[i % count] for i in range(8 * count)]), [i % count for i in range(8 * count)])
    # bench.py's schedule
    nstream, G = 16, 3
    ctxs = [api.Context(0, stream=torch.cuda.Stream()) for _ in range(nstream + G)]
    pool_h = [None] * B
    results = [None] * G

    def fe_worker(w):
        c = ctxs[w]
        for i in range(w, B, nstream):
            S, T = dev[i % count]
            if pool_h[i] is None:
                pool_h[i] = (c.cloud_create(cfg, S), c.cloud_create(cfg, T))
            else:
                pool_h[i][0].recompute(S); pool_h[i][1].recompute(T)

    def loop_group(g):
        results[g] = ctxs[nstream + g].register_clouds(cfg, pool_h[g * B // G:(g + 1) * B // G])

    def run(fn, n):
        th = [threading.Thread(target=fn, args=(w,)) for w in range(n)]
        [x.start() for x in th]
        [x.join() for x in th]

    for rnd in range(2):
        run(fe_worker, nstream)
        run(loop_group, G)
        flat = [st for r in results for st in r]
        compare("C bench schedule round %d (B=%d)" % (rnd, B), flat, [i % count for i in range(B)])


if __name__ == "__main__":
    main()
