#!/usr/bin/env python
"""Generates tests/golden/*.npz from the CPU oracle on seeded synthetic inputs.

The reference ships no golden vectors and cannot be built here (PCL/Eigen/FLANN absent), so these
fixtures freeze the ORACLE's outputs (whose KM part is pinned against the reference's own km.cpp).  They guard the
oracle against drift (tests/test_golden.py, CPU) and give the GPU tests a second, file-based target.
Inputs are regenerated from seeds by gh-icp_amd/synth.py, only outputs (+ tiny inputs) are stored.

    python tests/golden/make_golden.py        # rewrites the fixtures
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def inputs():
    synth = importlib.import_module("gh-icp_amd.synth")
    from oracle import oracle as O

    tls = synth.tls_pair(40_000, pair_id=7)
    ds = tls.target[O.voxel_filter(tls.target, 0.1)]
    g = synth.gauss_pair(n=5000, n_kp=300)
    return synth, O, tls, ds, g


ICP_VARIANTS = (("p2p", 0, False, False), ("p2p_trim", 0, False, True), ("p2p_recip_trim", 0, True, True), ("p2plane_trim", 1, False, True))


def icp_inputs():
    """Source / target of the fine-registration fixture (same generator as tests/test_icp_cpu.py::small_pair)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from test_icp_cpu import small_pair

    synth = importlib.import_module("gh-icp_amd.synth")
    return small_pair(synth, n=5000, seed=9)


def icp_fixture(O):
    src, tgt, gt = icp_inputs()
    out = {"overlap_0p2": np.float32(O.cal_overlap(src, tgt, 0.2)), "normals_k12": O.knn_normals(tgt, 12)}
    idx, d2 = O.nn1(src, tgt)
    out["nn_idx"], out["nn_d2"] = idx, d2
    for name, metric, recip, trim in ICP_VARIANTS:
        r = O.icp(src, tgt, O.icp_params(40, recip, trim, metric, 0.2, 0.1, 12))
        out[name + "_T"] = r["T"]
        out[name + "_meta"] = np.array([r["iterations"], r["converged"], r["reason"], r["correspondences"]], np.int64)
        out[name + "_corr0"] = r["corr0"]
    np.savez_compressed(os.path.join(HERE, "icp.npz"), **out)


def main():
    synth, O, tls, ds, g = inputs()
    pat = synth.bsc_pattern_glibc()
    # --- KM: the reference's commented 3x3 example (km.cpp:237-259) + a seeded 40x40
    W3 = np.array([[-5, -2, -100], [-4, -2, -6], [-100, -1, -7]], float)
    rng = np.random.default_rng(42)
    cd = 5 + 60 * rng.random((40, 40))
    cd = np.where(rng.random((40, 40)) < 0.15, 8 * rng.random((40, 40)), cd)
    W40 = np.where(cd < 8.0, -cd, -8.0)
    np.savez_compressed(os.path.join(HERE, "km.npz"), W3=W3, match3=O.km(W3)[0], W40=W40, match40=O.km(W40)[0])
    # --- front end on the down-sampled target of tls_pair(40000, pair_id=7)
    keep = O.voxel_filter(tls.target, 0.1)
    lam, curv, cnt = O.pca(ds, 0.5)
    cand = O.prune(lam, cnt)
    kp = O.nms(ds, curv, cand, 1.5)
    feat, lcs, _ = O.bsc(ds, kp, 1.5, 6, pat)
    nrm, hist = O.fpfh(ds[:3000])
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), keep=keep, count=cnt, lam=lam, cand=cand, kp=kp, feat=feat, lcs=lcs,
                        normals=nrm, fpfh=hist)
    # --- the loop on cfg1-like keypoints (Gaussian blobs, 300 explicit keypoints)
    kpS = g.source[g.kp_source].astype(np.float64)
    kpT = g.target[g.kp_target].astype(np.float64)
    bbx = O.bbx_magnitude(g.source)
    out = {}
    for name, corr in (("nn", O.NN), ("nnr", O.NNR), ("km", O.KM)):
        r = O.register(O.default_params(O.NONE, corr, 6, 0.9, 1.5, bbx, max_iter=60), kpS, kpT, want_matchlist=True)
        out[name + "_Rt"] = r["Rt"]
        out[name + "_cor"] = np.array([t["cor"] for t in r["trace"]], np.int32)
        out[name + "_matchlist"] = r["matchlist"]
    np.savez_compressed(os.path.join(HERE, "loop.npz"), **out)
    icp_fixture(O)
    print("golden fixtures written:", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
