"""An INDEPENDENT check of the oracle's FPFH restatement (SURVEY.md §8c calls it "the least certain restatement"; round-4 verdict, missing #4).

`FPFHfeature::compute_fpfh_feature` (reference include/fpfh.hpp:36-58) is two PCL calls: `pcl::NormalEstimation` (k = 20, viewpoint at the
origin) followed by `pcl::FPFHEstimationOMP` (k = 20) over the whole cloud.  PCL is not installed and its sources are not on disk, so
`oracle/ghicp_oracle.cpp` restates them, and `pytest -m gpu` only proves GPU == oracle.  This file is a slow numpy transcription written from
the UPSTREAM ALGORITHM DESCRIPTION (Rusu et al. 2009 "Fast Point Feature Histograms" + the documented behaviour of PCL's classes), not from
anything under `oracle/` -- it shares no code and no arithmetic contract with the restatement (scipy's KD-tree instead of the grid search,
numpy's `eigh` instead of the contract's Jacobi, `np.arctan2` instead of the contract's atan2f, f64 throughout):

* NormalEstimation: k nearest neighbours INCLUDING the query, covariance of the neighbourhood, normal = eigenvector of the smallest
  eigenvalue, flipped so that (viewpoint - p) . n >= 0.
* SPFH of p over its k neighbours: the neighbour at distance 0 (p itself) is skipped; per pair the Darboux frame is anchored at the point
  whose normal makes the SMALLER angle with the connecting line; u = n_s, v = (p_t - p_s) x u normalised, w = u x v;
  f1 = atan2(w . n_t, u . n_t) in [-pi, pi], f2 = v . n_t in [-1, 1], f3 = u . (p_t - p_s) / d in [-1, 1]; three 11-bin histograms,
  bin = floor(11 * normalised feature) clamped, increment 100 / (k - 1).
* FPFH of p = sum over its neighbours at non-zero distance of SPFH(q) / d^2 (PCL leaves SPFH(p) itself out), each 11-bin block rescaled to 100.

Where the two can legitimately differ: a feature value within rounding of a bin edge lands in the neighbouring bin (one increment of
100 / 19 moves, and spreads over the ~20 FPFH rows that weight this SPFH) and the oracle's covariance is the N2 contract (f64 sums rounded
once to the f32 grid).  The bounds below are what that allows, not more.
"""
import numpy as np
import pytest
from scipy.spatial import cKDTree

K = 20
NB = 11


def pcl_like_normals(p, idx):
    m = p.shape[0]
    nrm = np.zeros((m, 3))
    for i in range(m):
        q = p[idx[i]]
        c = q.mean(0)
        d = q - c
        w, v = np.linalg.eigh(d.T @ d / len(q))
        n = v[:, 0]
        if np.dot(-p[i], n) < 0:  # flipNormalTowardsViewpoint, viewpoint (0, 0, 0)
            n = -n
        nrm[i] = n
    return nrm


def pair_features(p1, n1, p2, n2):
    dp = p2 - p1
    f4 = np.linalg.norm(dp)
    if f4 == 0.0:
        return None
    a1 = np.dot(n1, dp) / f4
    a2 = np.dot(n2, dp) / f4
    if np.arccos(min(1.0, abs(a1))) > np.arccos(min(1.0, abs(a2))):  # the frame sits where the normal is closer to the line
        u, nt, dp, f3 = n2, n1, -dp, -a2
    else:
        u, nt, f3 = n1, n2, a1
    v = np.cross(dp, u)
    vn = np.linalg.norm(v)
    if vn == 0.0:
        return None
    v = v / vn
    w = np.cross(u, v)
    return np.arctan2(np.dot(w, nt), np.dot(u, nt)), np.dot(v, nt), f3


def spfh(p, nrm, idx):
    m = p.shape[0]
    h = np.zeros((m, 3, NB))
    inc = 100.0 / (K - 1)
    for i in range(m):
        for j in idx[i]:
            if j == i:
                continue
            f = pair_features(p[i], nrm[i], p[j], nrm[j])
            if f is None:
                continue
            b1 = int(np.floor(NB * ((f[0] + np.pi) / (2.0 * np.pi))))
            b2 = int(np.floor(NB * ((f[1] + 1.0) * 0.5)))
            b3 = int(np.floor(NB * ((f[2] + 1.0) * 0.5)))
            h[i, 0, min(NB - 1, max(0, b1))] += inc
            h[i, 1, min(NB - 1, max(0, b2))] += inc
            h[i, 2, min(NB - 1, max(0, b3))] += inc
    return h


def fpfh(p, idx, d2, s):
    m = p.shape[0]
    out = np.zeros((m, 3, NB))
    for i in range(m):
        for j, dd in zip(idx[i], d2[i]):
            if dd == 0.0:
                continue
            out[i] += s[j] / dd
        for b in range(3):
            t = out[i, b].sum()
            if t != 0.0:
                out[i, b] *= 100.0 / t
    return out.reshape(m, 33)


@pytest.mark.parametrize("scene", ["tls", "blob"])
def test_oracle_fpfh_against_an_independent_numpy_transcription_of_pcl(oracle, synth, scene):
    if scene == "tls":
        pr = synth.tls_pair(30_000)
        ds = pr.target[oracle.voxel_filter(pr.target, 0.3)][:, :3]
    else:
        pr = synth.gauss_pair(6_000)
        ds = pr.target[:, :3]
    rng = np.random.default_rng(5)
    # a compact region (the kNN sets must be those of the cloud both sides see): the 2000 points nearest to a random one
    centre = ds[rng.integers(len(ds))]
    sub = np.ascontiguousarray(ds[np.argsort(((ds - centre) ** 2).sum(1))[:2000]], dtype=np.float32)
    p = sub.astype(np.float64)
    d, idx = cKDTree(p).query(p, k=K)
    # a tie at the k-th distance makes the neighbour SET implementation-defined (FLANN's order; the regular angular grid of a synthetic
    # scan of a plane produces a few): such rows, and the rows that weight their SPFH, are left out of the comparison
    dk1 = np.sort(((p[:, None, :] - p[None, :, :]) ** 2).sum(2), axis=1)[:, K - 1:K + 1]
    amb = dk1[:, 1] <= dk1[:, 0]
    keep = ~(amb | amb[idx].any(1))
    assert keep.mean() > 0.8
    nrm = pcl_like_normals(p, idx)
    ref = fpfh(p, idx, d * d, spfh(p, nrm, idx))

    o_nrm, o_hist = oracle.fpfh(sub, K)
    # normals: same direction (the contract's covariance is rounded to the f32 grid once; eigenvectors of nearly planar patches are well
    # conditioned, of nearly isotropic ones they are not: the bound is on the bulk, the tail is counted)
    cosang = np.abs((nrm * o_nrm).sum(1))[~amb]
    same_side = ((nrm * o_nrm).sum(1) > 0)[~amb]
    assert (cosang > 1.0 - 1e-6).mean() > 0.97
    assert (cosang > 1.0 - 1e-3).mean() > 0.995
    assert same_side.mean() > 0.995  # a normal perpendicular to the line of sight may flip either way
    # histograms: block sums and the values themselves
    np.testing.assert_allclose(o_hist.reshape(-1, 3, NB).sum(2)[keep], ref.reshape(-1, 3, NB).sum(2)[keep], atol=2e-3)
    diff = np.abs(o_hist - ref).max(1)[keep]
    # a bin-edge flip moves one increment of 100 / 19 inside ONE SPFH and reaches the ~20 rows that weight it; rows without one agree to
    # float precision.  Measured (round 5): tls 1981 rows, median 0, 90 % below 1.8e-5, 99.5 % below 2.5, max 7.6 (of a block mass of 100);
    # blob (isotropic noise: features spread over every bin edge) median 9e-6, 90 % below 0.09, max 4.7; every normal within 1e-6
    assert np.median(diff) < 5e-4
    assert (diff < 0.05).mean() > (0.95 if scene == "tls" else 0.85)
    assert (diff < 3.0).mean() > 0.99 and diff.max() < 12.0
    # and the quantity the registration consumes (|Pearson| between rows, fpfh.hpp:135-165) is the same matrix
    sel = rng.choice(np.flatnonzero(keep), 60, replace=False)
    FD_o = oracle.fd_fpfh(o_hist[sel], o_hist[sel])
    FD_r = oracle.fd_fpfh(ref[sel].astype(np.float32), ref[sel].astype(np.float32))
    assert np.abs(FD_o - FD_r).max() < 0.02
