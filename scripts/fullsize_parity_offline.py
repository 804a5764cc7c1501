"""Full-size cfg3 / cfg5 parity, off the GPU box: the per-scene results that `bench.py --config N` measured on the MI355X (its
JSON line, key "scenes") against the CPU restatement run here on the same seeded inputs.
    python scripts/fullsize_parity_offline.py CONFIG gpurun_out/r02_bench_cfgN.json OUT.json"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import oracle as O  # checker only


def main():
    cfg_id, src, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    synth = importlib.import_module("gh-icp_amd.synth")
    CF = bench.CONFIGS[cfg_id]
    d = json.loads(open(src).read().strip().splitlines()[-1])
    rows = []
    for sc in d["scenes"]:
        p = bench.make_pair(cfg_id, sc["pair_id"], CF["hits"])
        t = time.time()
        r = O.register_pair(p.source, p.target, CF["voxel"], CF["r"], CF["R"], CF["dof"], {"BSC": O.BSC, "FPFH": O.FPFH}[CF["feature"]],
                            {"KM": O.KM, "NN": O.NN, "NNR": O.NNR}[CF["corr"]], CF["iou"], synth.bsc_pattern_glibc(), max_iter=200)
        Rg = np.array(sc["Rt"]).reshape(4, 4)
        rows.append({"pair_id": sc["pair_id"], "gpu": {k: sc[k] for k in ("m_s", "m_t", "k_s", "k_t", "iterations")},
                     "oracle": {"m_s": r["m_s"], "m_t": r["m_t"], "k_s": r["k_s"], "k_t": r["k_t"], "iterations": r["iters"]},
                     "rot_err": synth.rot_err(Rg, r["Rt"]), "trans_err_m": synth.trans_err(Rg, r["Rt"]),
                     "oracle_seconds": {k: round(v, 2) for k, v in r["seconds"].items()}, "oracle_wall_s": round(time.time() - t, 1)})
        print(rows[-1], flush=True)
    json.dump({"config": cfg_id, "workload": d["config"]["workload"], "gpu_pairs_per_s": d["value"], "gpu_single_pair_latency_s": d["single_pair_latency_s"],
               "tolerance": "1e-4 rotation, 1e-3 m translation", "pairs": rows,
               "cpu": "oracle (PCL-free restatement), g++ -O2 contract build, 1 thread, this container (8 x Xeon 2.1 GHz)"}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
