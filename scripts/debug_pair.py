"""Stage-by-stage parity of ONE synthetic pair, GPU (through the C ABI) against the CPU restatement: voxel filter, keypoints,
BSC strings, feature distance, per-iteration trace.  python scripts/debug_pair.py CONFIG PAIR_ID [HITS]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # checker only


def main():
    import torch
    cfg_id, pid = int(sys.argv[1]), int(sys.argv[2])
    api = importlib.import_module("gh-icp_amd.api")
    synth = importlib.import_module("gh-icp_amd.synth")
    sys.path.insert(0, ROOT)
    import bench
    CF = bench.CONFIGS[cfg_id]
    hits = int(sys.argv[3]) if len(sys.argv) > 3 else CF["hits"]
    p = bench.make_pair(cfg_id, pid, hits)
    ctx = api.Context(0)
    pat = synth.bsc_pattern_glibc()
    feats, kps, dss = {}, {}, {}
    for name, cloud, dof in (("T", p.target, 0), ("S", p.source, CF["dof"])):
        ko = O.voxel_filter(cloud, CF["voxel"])
        kg = ctx.voxel_filter(cloud, CF["voxel"]).cpu().numpy()
        print(name, "voxel keep equal", np.array_equal(ko, kg), len(ko))
        ds = cloud[ko]
        kpo, _ = O.keypoints(ds, CF["r"], CF["R"])
        kpg = ctx.keypoints(ds, CF["r"], CF["R"]).cpu().numpy()
        print(name, "keypoints equal", np.array_equal(kpo, kpg), kpo.size, kpg.size)
        fo, lo, _ = O.bsc(ds, kpo, CF["R"], dof, pat)
        fg, lg = ctx.bsc_encode(ds, kpo, CF["R"], dof, pat)
        fg, lg = fg.cpu().numpy(), lg.cpu().numpy()
        ham = np.unpackbits(fg ^ fo, axis=-1).sum(-1)
        print(name, "LCS equal", np.array_equal(lg, lo), "BSC strings differing", int((ham > 0).sum()), "max bits", int(ham.max()),
              "keypoints:", np.argwhere(ham > 0)[:10].tolist())
        feats[name], kps[name], dss[name] = (fo, fg), kpo, ds
    V = 4 if CF["dof"] > 4 else 2
    FDo = O.fd_bsc(feats["S"][0][:V], feats["T"][0][0])
    FDg = ctx.fd_bsc(feats["S"][1][:V], feats["T"][1][0]).cpu().numpy().astype(np.float64)
    print("FD entries differing", int((FDo != FDg).sum()), "max |diff|", float(np.abs(FDo - FDg).max()))
    bbx = O.bbx_magnitude(dss["S"])
    corr = {"KM": (O.KM, api.CORR_KM), "NN": (O.NN, api.CORR_NN), "NNR": (O.NNR, api.CORR_NNR)}[CF["corr"]]
    kS, kT = dss["S"][kps["S"]].astype(np.float64), dss["T"][kps["T"]].astype(np.float64)
    ro = O.register(O.default_params(O.BSC, corr[0], CF["dof"], CF["iou"], CF["R"], bbx, max_iter=200), kS, kT, FDo)
    for label, FD in (("oracle FD", FDo), ("gpu FD", FDg)):
        rg = ctx.register(api.default_params(api.FEATURE_BSC, corr[1], CF["dof"], CF["iou"], CF["R"], bbx, max_iter=200), kS, kT,
                          torch.from_numpy(FD.astype(np.int16)).cuda())
        n = min(rg["iters"], ro["iters"])
        first = next((i for i in range(n) if rg["trace"][i]["cor"] != ro["trace"][i]["cor"] or abs(rg["trace"][i]["penalty"] - ro["trace"][i]["penalty"]) > 1e-9 * abs(ro["trace"][i]["penalty"])), None)
        print("loop with", label, ": iters gpu/oracle", rg["iters"], ro["iters"], "first differing iteration", first,
              "rot err %.3e trans err %.3e" % (synth.rot_err(rg["Rt"], ro["Rt"]), synth.trans_err(rg["Rt"], ro["Rt"])))
        if first is not None:
            for i in range(max(0, first - 1), min(n, first + 2)):
                tg, to = rg["trace"][i], ro["trace"][i]
                print("   it", i, {k: (tg[k], to[k]) for k in ("cor", "penalty", "CDmean", "CDstd", "rmse") if k in tg})


if __name__ == "__main__":
    main()
