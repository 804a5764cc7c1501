"""Times ghicp_km_solve on real Kuhn-Munkres weight matrices of cfg2 registrations (stored sparsely under tests/golden/km_cfg2_*.npz):
iterations 0, 10 and 30 of scene 0 (n = 840: DFS-heavy, flood-heavy, average) and, with --more, iteration 46 of scene 22 (n = 758, the
heaviest of the 2220 matrices of the 64 bench scenes: 6.9 k failed phases, 0.6 M flood rows) and iteration 0 of scene 53 (n = 1131, the
largest).  --check compares the matching with the CPU restatement; --lib PATH times another build of the library (variants of a kernel
in ONE gpurun call: profiles/r03_km4_second_half.txt); GHICP_KM_STATS=1 prints the solver's step mix and cycle split."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(it):
    z = np.load(os.path.join(ROOT, "tests", "golden", "km_cfg2_%s.npz" % (it if isinstance(it, str) else "it%d" % it)))
    n = int(z["n"])
    w = np.full((n, n), float(z["bg"]))
    w[z["rows"].astype(np.int64), z["cols"].astype(np.int64)] = z["vals"]
    return w


def main():
    import torch

    api = importlib.import_module("gh-icp_amd.api")
    if "--lib" in sys.argv:
        api.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    ctx = api.Context(0)
    check = "--check" in sys.argv
    for it in (0, 10, 30) + (("s22_it46", "s53_it0") if "--more" in sys.argv else ()):
        w = load(it)
        wd = torch.from_numpy(w).cuda()
        ctx.km_solve(wd)
        torch.cuda.synchronize()
        t = time.perf_counter()
        m = ctx.km_solve(wd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        line = "it %s n %d solve ms %.1f checksum %d" % (it, w.shape[0], dt * 1e3, int(m.sum()))
        if check:
            from oracle import oracle as O  # checker only

            line += " exact %s" % bool((m.cpu().numpy() == O.km(w)[0]).all())
        print(line, flush=True)


if __name__ == "__main__":
    main()
