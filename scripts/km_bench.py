"""Times ghicp_km_solve on the three real Kuhn-Munkres weight matrices of a cfg2 registration (iterations 0, 10 and 30 of
pair 0; stored sparsely under tests/golden/km_cfg2_it*.npz) and checks the matching against the CPU restatement.
GHICP_KM_STATS=1 prints the solver's step mix and cycle split (profiles/r01_km_step_counters_v2.txt)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(it):
    z = np.load(os.path.join(ROOT, "tests", "golden", "km_cfg2_it%d.npz" % it))
    n = int(z["n"])
    w = np.full((n, n), float(z["bg"]))
    w[z["rows"].astype(np.int64), z["cols"].astype(np.int64)] = z["vals"]
    return w


def main():
    import torch

    api = importlib.import_module("gh-icp_amd.api")
    ctx = api.Context(0)
    check = "--check" in sys.argv
    for it in (0, 10, 30):
        w = load(it)
        wd = torch.from_numpy(w).cuda()
        ctx.km_solve(wd)
        torch.cuda.synchronize()
        t = time.perf_counter()
        m = ctx.km_solve(wd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        line = "it %d n %d solve ms %.1f checksum %d" % (it, w.shape[0], dt * 1e3, int(m.sum()))
        if check:
            from oracle import oracle as O  # checker only

            line += " exact %s" % bool((m.cpu().numpy() == O.km(w)[0]).all())
        print(line, flush=True)


if __name__ == "__main__":
    main()
