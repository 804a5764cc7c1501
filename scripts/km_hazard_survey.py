"""TEST / ANALYSIS TOOL (CPU only): runs whole cfg2 registrations through the oracle and hands every Kuhn-Munkres weight matrix they
solve to the rule-level model of the GPU solver (oracle/km4_model.inc, the kernel's configuration: cap 3, hints 6, the rest exact).
Answers what a default bench run cannot: how often rule R4's hazard fires on real registrations (the kernel then re-solves the whole
problem on one lane -- the suspected source of the 1-4 s solves), and how the phase / flood / DFS counts that drive a
solve's time are distributed over the iterations of a pair.
    python scripts/km_hazard_survey.py [first_pair] [pairs] [procs]      -> one line per pair + a summary; --json FILE keeps the records;
    --seed runs the model with the seeded flood (R3', design groundwork) and reports how many floods could not be certified"""
import ctypes as C
import importlib
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ("phases", "failed", "flood_rows", "push_rows", "rebuild_rows", "pull_rounds", "dfs_steps", "dfs_pops", "overflow_rows", "seeded", "unseeded")
SEED = "--seed" in sys.argv  # rule R3' of the model (seeded flood with a tree-edge certificate; self-checked: status 7 on a wrong closure)


def work(pair_id):
    import numpy as np

    from oracle import oracle as O  # the checker, used here as an analysis tool

    bench = importlib.import_module("bench")
    synth = importlib.import_module("gh-icp_amd.synth")
    cfg = bench.CONFIGS[2]
    p = synth.tls_pair(cfg["hits"], config_id=2, pair_id=pair_id)
    recs = []
    lib = O.lib()
    lib.orc_set_km_observer.argtypes = [C.c_void_p]

    @C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_double)
    def observer(it, n, w, penalty):
        match = np.empty(n, np.int32)
        st = np.zeros(11, np.int64)
        t = time.perf_counter()
        rc = lib.orc_km4_model(w, n, C.c_double(0.01), match.ctypes.data_as(C.POINTER(C.c_int)), st.ctypes.data_as(C.POINTER(C.c_longlong)),
                               3 | (6 << 10) | 0x10000 | (0x20000 if SEED else 0))
        a = np.ctypeslib.as_array(w, shape=(n * n,))
        recs.append(dict(it=it, n=n, rc=rc, nnz=int((a != -penalty).sum()), model_s=time.perf_counter() - t, **dict(zip(NAMES, (int(v) for v in st)))))

    lib.orc_set_km_observer(C.cast(observer, C.c_void_p))
    t = time.perf_counter()
    r = O.register_pair(p.source, p.target, cfg["voxel"], cfg["r"], cfg["R"], cfg["dof"], O.BSC, O.KM, cfg["iou"], pattern=synth.bsc_pattern_glibc())
    lib.orc_set_km_observer(None)
    return dict(pair=pair_id, iters=r["iters"], k_s=r["k_s"], k_t=r["k_t"], seconds=time.perf_counter() - t, km=recs)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if len(args) > 0 else 0
    count = int(args[1]) if len(args) > 1 else 8
    procs = int(args[2]) if len(args) > 2 else min(8, os.cpu_count() or 1)
    out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    from oracle import oracle as O

    O.build()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(work, range(first, first + count))
    solves = hazards = 0
    worst = []
    for r in res:
        km = r["km"]
        hz = [k["it"] for k in km if k["rc"] == 4]
        other = [k["rc"] for k in km if k["rc"] not in (0, 4)]
        solves += len(km)
        hazards += len(hz)
        top = max(km, key=lambda k: k["flood_rows"] + k["dfs_steps"])
        worst.append((top["flood_rows"] + top["dfs_steps"], r["pair"], top))
        print("pair %3d  K %4d / %4d  iterations %3d  solves %3d  hazard at iterations %s  other status %s  failed phases min/mean/max %d / %.0f / %d  "
              "flood rows max %d  DFS activations max %d" % (r["pair"], r["k_s"], r["k_t"], r["iters"], len(km), hz or "-", other or "-",
                                                          min(k["failed"] for k in km), sum(k["failed"] for k in km) / len(km), max(k["failed"] for k in km),
                                                          max(k["flood_rows"] for k in km), max(k["dfs_steps"] for k in km)), flush=True)
    print("solves %d, hazard reports %d (%.2f %%)" % (solves, hazards, 100.0 * hazards / max(1, solves)))
    if SEED:
        print("seeded floods %d, not certified (flood from the root) %d, flood rows %d" % (sum(k["seeded"] for r in res for k in r["km"]),
              sum(k["unseeded"] for r in res for k in r["km"]), sum(k["flood_rows"] for r in res for k in r["km"])))
    worst.sort(reverse=True, key=lambda t: t[0])
    for _, pair, k in worst[:3]:
        print("heaviest solve: pair %d iteration %d n %d: %s" % (pair, k["it"], k["n"], {n: k[n] for n in NAMES}))
    if out:
        with open(out, "w") as f:
            json.dump(res, f)


if __name__ == "__main__":
    main()
