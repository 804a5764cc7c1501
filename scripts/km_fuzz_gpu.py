"""GPU Kuhn-Munkres vs the CPU restatement on fuzzed matrices (generators of scripts/km4_model_fuzz.py) and, with
GHICP_KM_FORCE_HAZARD=1 in the environment, through the hazard fallback.  python scripts/km_fuzz_gpu.py SEED COUNT [NMAX]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import km4_model_fuzz as F
from oracle import oracle as O  # checker only


def main():
    import torch
    api = importlib.import_module("gh-icp_amd.api")
    ctx = api.Context(0)
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    sizes = [s for s in (1, 2, 3, 5, 8, 17, 40, 64, 65, 100, 130, 200, 333, 400, 700, 1000) if s <= nmax]
    bad = 0
    t0 = time.time()
    for t in range(N):
        n = int(rng.choice(sizes))
        w = F.gen(rng, n, t % 5)
        ref, _ = O.km(w)
        m = ctx.km_solve(torch.from_numpy(w).cuda()).cpu().numpy()
        if not (m == ref).all():
            bad += 1
            print("MISMATCH t", t, "n", n, "kind", t % 5, flush=True)
    print("gpu km fuzz: matrices", N, "mismatches", bad, "seconds %.1f" % (time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
