"""TEST INFRASTRUCTURE: fuzzes the Kuhn-Munkres KERNELS themselves (k_km4 and its variants, compiled for the host SIMT interpreter of
tests/hipsim) against the restatement of the reference's km.cpp on the generators of scripts/km4_model_fuzz.py -- thousands of matrices,
no GPU.  python tests/hipsim/km_fuzz_sim.py [count] [procs] [mode]   (mode: default | GHICP_KM_FORCE_HAZARD)"""
import importlib
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def work(args):
    seed, count, mode = args
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    os.environ["HIPSIM_THREADS"] = "1"
    if mode != "default":
        os.environ[mode] = "1"
    import numpy as np

    import km4_model_fuzz as F
    from hipsim import simctx
    from oracle import oracle as O

    api = importlib.import_module("gh-icp_amd.api")
    ctx = simctx.make_context(api)
    O.build()
    rng = np.random.default_rng(seed)
    sizes = [1, 2, 3, 4, 5, 7, 16, 17, 31, 32, 33, 63, 64, 65, 100, 130, 200]
    bad = []
    for t in range(count):
        n = int(rng.choice(sizes))
        g = t % 5
        w = F.gen(rng, n, g)
        got = ctx.km_solve(w).cpu().numpy()
        ref = O.km(w)[0]
        if not (got == ref).all():
            bad.append((seed, t, n, g))
    return bad


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    mode = sys.argv[3] if len(sys.argv) > 3 else "default"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hipsim import build

    build.build()
    per = max(1, count // procs)
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(work, [(1000 + p, per, mode) for p in range(procs)])
    bad = [b for r in res for b in r]
    print("mode %s: %d matrices through the kernel on the interpreter, %d mismatches %s" % (mode, per * procs, len(bad), bad[:10]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
