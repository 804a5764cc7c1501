"""TEST INFRASTRUCTURE: whole cfg2 registrations (BSC + Kuhn-Munkres, the bench scenes at full size) through the library's GH-ICP loop
compiled for the host SIMT interpreter, iteration by iteration against the oracle.  The front end (voxel filter, keypoints, BSC, feature
distance) comes from the oracle so that the minutes go to what is being checked: every Kuhn-Munkres matrix a real registration produces
(n = 350 ... 1131, 26 ... 55 per pair) through the persistent pair loop and the shipped solver's source.  No GPU.
    python tests/hipsim/loop_real_sim.py [scene ids ...]      (default: 0 22 53; ~5-15 min per scene, one process each)"""
import importlib
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def work(scene):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["HIPSIM_THREADS"] = "1"
    import numpy as np
    import torch

    from hipsim import simctx
    from oracle import oracle as O

    api = importlib.import_module("gh-icp_amd.api")
    synth = importlib.import_module("gh-icp_amd.synth")
    bench = importlib.import_module("bench")
    cfg = bench.CONFIGS[2]
    p = synth.tls_pair(cfg["hits"], config_id=2, pair_id=scene)
    t0 = time.time()
    ds = {k: c[O.voxel_filter(c, cfg["voxel"])] for k, c in (("S", p.source), ("T", p.target))}
    kp = {k: O.keypoints(ds[k], cfg["r"], cfg["R"])[0] for k in ("T", "S")}
    pattern = synth.bsc_pattern_glibc()
    fT = O.bsc(ds["T"], kp["T"], cfg["R"], 0, pattern)[0]
    fS = O.bsc(ds["S"], kp["S"], cfg["R"], cfg["dof"], pattern)[0]
    FD = O.fd_bsc(fS[:4], fT[0])
    kpS, kpT = ds["S"][kp["S"]][:, :3].astype(np.float64), ds["T"][kp["T"]][:, :3].astype(np.float64)
    bbx = O.bbx_magnitude(ds["S"])
    ro = O.register(O.default_params(O.BSC, O.KM, cfg["dof"], cfg["iou"], cfg["R"], bbx, max_iter=200), kpS, kpT, FD, want_matchlist=True)
    t1 = time.time()
    ctx = simctx.make_context(api)
    pg = api.default_params(api.FEATURE_BSC, api.CORR_KM, cfg["dof"], cfg["iou"], cfg["R"], bbx, max_iter=200)
    rg = ctx.register(pg, kpS, kpT, torch.from_numpy(FD.astype(np.int16)).to(ctx.dev), want_matchlist=True)
    t2 = time.time()
    same_iters = rg["iters"] == ro["iters"]
    n_it = min(rg["iters"], ro["iters"])
    bad_it = [i for i in range(n_it) if not (rg["matchlist"][i] == ro["matchlist"][i]).all()]
    dRt = float(np.abs(rg["Rt"] - ro["Rt"]).max())
    ok = same_iters and not bad_it and dRt < 1e-6
    return dict(scene=scene, ok=ok, k=(int(kpS.shape[0]), int(kpT.shape[0])), iters=(rg["iters"], ro["iters"]), first_bad_iteration=bad_it[:1], dRt=dRt,
                oracle_s=round(t1 - t0, 1), sim_s=round(t2 - t1, 1))


def main():
    scenes = [int(a) for a in sys.argv[1:]] or [0, 22, 53]
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hipsim import build
    from oracle import oracle as O

    build.build()
    O.build()
    with mp.get_context("spawn").Pool(min(len(scenes), os.cpu_count() or 1)) as pool:
        bad = 0
        for r in pool.imap_unordered(work, scenes):
            print(r, flush=True)
            bad += not r["ok"]
    print("%d scene(s), %d mismatch(es)" % (len(scenes), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
