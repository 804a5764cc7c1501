"""Times the fine-registration path (ghicp_icp) on the down-sampled clouds of a cfg2-sized scan pair that was
coarsely aligned first -- the situation CRegistration::icp_reg is meant for.  Prints one JSON line per variant."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hits", type=int, default=1_000_000)
    ap.add_argument("--voxel", type=float, default=0.1)
    ap.add_argument("--max-iter", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch

    api = importlib.import_module("gh-icp_amd.api")
    synth = importlib.import_module("gh-icp_amd.synth")
    ctx = api.Context(0)
    pair = synth.tls_pair(a.hits)
    S = torch.from_numpy(pair.source).cuda()
    T = torch.from_numpy(pair.target).cuda()
    dsS = S[ctx.voxel_filter(S, a.voxel).long()][:, :3].contiguous()
    dsT = T[ctx.voxel_filter(T, a.voxel).long()][:, :3].contiguous()
    ang = np.deg2rad(1.0)
    d = np.eye(4)
    d[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    d[:3, 3] = [0.15, -0.1, 0.03]
    coarse = d @ pair.gt
    S0 = ctx.transform_cloud(dsS, coarse)
    for name, kw in (("p2p", dict()), ("p2p_trimmed", dict(trimmed=True)), ("p2p_trimmed_reciprocal", dict(trimmed=True, reciprocal=True)),
                     ("p2plane_trimmed", dict(trimmed=True, metric=api.ICP_POINT_TO_PLANE))):
        p = api.icp_params(a.max_iter, thre_dis=0.3, min_overlap=0.1, covariance_k=15, **kw)
        best = None
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = ctx.icp(S0, dsT, p)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        total = r["T"].astype(np.float64) @ coarse
        R = total[:3, :3] @ pair.gt[:3, :3].T
        print(json.dumps(dict(variant=name, ns=int(S0.shape[0]), nt=int(dsT.shape[0]), iterations=r["iterations"], reason=r["reason"],
                              correspondences=r["correspondences"], overlap=round(r["overlap"], 4), seconds=round(best, 4),
                              ms_per_iteration=round(1e3 * best / max(1, r["iterations"]), 3),
                              rot_err_vs_gt=float(np.linalg.norm(R - np.eye(3))), trans_err_vs_gt=float(np.linalg.norm(total[:3, 3] - pair.gt[:3, 3])),
                              fitness=r["fitness"])), flush=True)


if __name__ == "__main__":
    main()
