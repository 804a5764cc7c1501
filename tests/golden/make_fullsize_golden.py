"""Generates tests/golden/fullsize.json: the CPU restatement (oracle, contract build) on BASELINE.json's FULL-SIZE configurations
that are too slow to run live inside `pytest -m gpu` (cfg3 needs ~75 s per pair on a CPU core, cfg5 pair 1 ~200 s):
  cfg2 pair 0 (1 M pts, BSC + KM), cfg3 pairs 0 and 1 (5 M pts, FPFH + NNR), cfg5 pair 1 (10 M pts, BSC + KM, 4-DoF; converges in 77
  iterations -- pair 0 runs into the 200-iteration guard on both sides and is a weak test of the configuration).
The inputs are the seeded synthetic generators of gh-icp_amd/synth.py (seed = 0x5EED0000 + 256 * config + pair), so the GPU test
regenerates identical clouds on the GPU box and compares its 4x4 with the one stored here (tests/test_gpu_fullsize.py).
    python tests/golden/make_fullsize_golden.py            # ~7 minutes on one core
    python tests/golden/make_fullsize_golden.py 3          # only the cases of config 3, merged into the existing file
Reference for the pipeline being restated: /root/reference/test/ghicp_main.cpp:86-153."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (config table + generators)
from oracle import oracle as O  # noqa: E402  (the checker)

CASES = [(2, 0), (3, 0), (3, 1), (5, 1), (5, 8)]  # cfg5: pair 1 converges to a rejected pose (77 iterations), pair 8 registers (25)


def main():
    synth = importlib.import_module("gh-icp_amd.synth")
    O.build()
    out_path = os.path.join(ROOT, "tests", "golden", "fullsize.json")
    # arguments: config ids (every case of those configs is recomputed) and / or "cfg:pair" (only that case); the other cases are kept
    only = {int(a) for a in sys.argv[1:] if ":" not in a}
    only_cases = {tuple(int(v) for v in a.split(":")) for a in sys.argv[1:] if ":" in a}
    sel = lambda c, pid: (not only and not only_cases) or c in only or (c, pid) in only_cases  # noqa: E731
    rows = [c for c in json.load(open(out_path))["cases"] if not sel(c["config"], c["pair_id"])] if (only or only_cases) and os.path.exists(out_path) else []
    for cfg_id, pair_id in CASES:
        if not sel(cfg_id, pair_id):
            continue
        CF = bench.CONFIGS[cfg_id]
        p = bench.make_pair(cfg_id, pair_id, CF["hits"])
        t = time.time()
        r = O.register_pair(p.source, p.target, CF["voxel"], CF["r"], CF["R"], CF["dof"], {"BSC": O.BSC, "FPFH": O.FPFH}[CF["feature"]],
                            {"KM": O.KM, "NN": O.NN, "NNR": O.NNR}[CF["corr"]], CF["iou"], synth.bsc_pattern_glibc(), max_iter=200)
        rows.append({"config": cfg_id, "pair_id": pair_id, "hits": CF["hits"], "m_s": r["m_s"], "m_t": r["m_t"], "k_s": r["k_s"], "k_t": r["k_t"],
                     "iterations": r["iters"], "converged": r["converged"], "registered_ok": r["registered_ok"], "rmse_after": r["rmse_after"],
                     "Rt": [float(v) for v in np.asarray(r["Rt"]).reshape(-1)], "gt": [float(v) for v in np.asarray(p.gt).reshape(-1)],
                     "source_sha": int(np.frombuffer(p.source.tobytes()[:4096], np.uint32).sum()),
                     "oracle_seconds": {k: round(v, 2) for k, v in r["seconds"].items()}, "wall_s": round(time.time() - t, 1)})
        print(rows[-1], flush=True)
    json.dump({"made_by": "tests/golden/make_fullsize_golden.py", "oracle": "oracle/libghicp_oracle.so (g++ -O2 -ffp-contract=off contract build)",
               "tolerance": "1e-4 rotation (||R_gpu R_cpu^T - I||_F), 1e-3 m translation",
               "cases": sorted(rows, key=lambda c: (c["config"], c["pair_id"]))}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
