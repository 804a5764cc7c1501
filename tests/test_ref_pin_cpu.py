"""Pins the oracle's restatement to the REFERENCE'S OWN compiled code wherever that code does not need PCL (oracle/_ref/
libghicp_ref.so, built by oracle/Makefile from the sources where they lie): Hamming distance and calFD_BSC, the FPFH histogram
distance, calED / calCD_{NF,BSC,FPFH} / findcorrespondence{NN,NNR,KM} / adjustweight per iteration, and the feature dump format.
CPU only.  Skipped when /root/reference is absent and no prebuilt oracle/_ref exists (the GPU box)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref(oracle):
    if oracle.ref2_lib() is None:
        pytest.skip("oracle/_ref/libghicp_ref.so not built (reference sources absent)")
    return oracle


def test_hamming_and_calfd_bsc(ref):
    """stereo_binary_feature.cpp:16-104 and ghicp_reg.cpp:143-200 against orc::fd_bsc (min over the V source variants)."""
    O = ref
    rng = np.random.default_rng(3)
    g = np.load(os.path.join(HERE, "golden", "frontend.npz"))
    a, b = rng.integers(0, 256, 56, dtype=np.uint8), rng.integers(0, 256, 56, dtype=np.uint8)
    a[55] &= 1; b[55] &= 1  # 441 bits: only bit 0 of the last byte belongs to the string
    assert O.ref_hamming(a, b) == int(np.unpackbits(a ^ b).sum()) == int(O.fd_bsc(a[None, None], b[None])[0, 0])
    assert O.ref_hamming(a, a) == 0
    for dof, V in ((6, 4), (4, 2)):
        fS = rng.integers(0, 256, (4, 23, 56), dtype=np.uint8); fS[..., 55] &= 1
        fT = rng.integers(0, 256, (17, 56), dtype=np.uint8); fT[..., 55] &= 1
        np.testing.assert_array_equal(O.ref_fd_bsc(fS, fT, dof), O.fd_bsc(fS[:V], fT))
        fS = g["feat"]  # golden descriptors (variants 1..3 carry the Q3 layout)
        np.testing.assert_array_equal(O.ref_fd_bsc(fS, fS[0][:40], dof), O.fd_bsc(fS[:V], fS[0][:40]))


def test_fpfh_distance(ref):
    """fpfh.hpp:135-165 (|Pearson r| in f32, NaN for a constant histogram) against orc::fpfh_distance."""
    O = ref
    rng = np.random.default_rng(5)
    g = np.load(os.path.join(HERE, "golden", "frontend.npz"))
    H = np.concatenate([g["fpfh"][:40], rng.uniform(0, 100, (10, 33)).astype(np.float32), np.full((1, 33), 3.0, np.float32)])
    FD = O.fd_fpfh(H, H)
    for i in range(0, H.shape[0], 3):
        for j in range(H.shape[0]):
            r = O.ref_fpfh_distance(H[i], H[j])
            assert (np.isnan(r) and np.isnan(FD[i, j])) or np.float32(r) == np.float32(FD[i, j]), (i, j, r, FD[i, j])


def _apply(kp, Rt):
    """the keypoint update of ghicp_reg.cpp:891-898 in the oracle's operation order"""
    x, y, z = kp[:, 0].copy(), kp[:, 1].copy(), kp[:, 2].copy()
    out = np.empty_like(kp)
    for r in range(3):
        out[:, r] = ((Rt[r, 0] * x + Rt[r, 1] * y) + Rt[r, 2] * z) + Rt[r, 3]
    return out


@pytest.mark.parametrize("feature", ["NONE", "BSC", "FPFH"])
@pytest.mark.parametrize("corr", ["NN", "NNR", "KM"])
def test_iteration_members(ref, feature, corr, tmp_path, monkeypatch):
    """Every iteration of an oracle run is replayed through the reference's own calED / calCD_* / findcorrespondence* (and
    adjustweight): penalty, correspondences, RMSE, FDM, FDstd and the adjusted weights must be IDENTICAL (f64 bit patterns)."""
    O = ref
    monkeypatch.chdir(tmp_path)  # findcorrespondenceKM -> Km::output writes Corres.txt into the cwd
    rng = np.random.default_rng(11 + 7 * len(feature) + len(corr))
    ks, kt = 46, 41
    kpT = rng.normal(size=(kt, 3)) * np.array([10.0, 6.0, 2.0])
    R = np.array([[0.98, -0.17, 0.03], [0.17, 0.98, 0.02], [-0.03, -0.01, 1.0]])
    perm = rng.permutation(kt)[:min(ks, kt)]
    kpS = (kpT[perm] - np.array([0.8, -0.5, 0.2])) @ R
    kpS = np.concatenate([kpS, rng.normal(size=(ks - kpS.shape[0], 3)) * 8.0]) + rng.normal(size=(ks, 3)) * 0.01
    bbx = 400.0
    FD = None
    if feature == "BSC":  # Hamming distances: true pairs close, the rest far
        FD = rng.integers(150, 260, (ks, kt)).astype(np.float64)
        FD[np.arange(perm.size), perm] = rng.integers(15, 60, perm.size)
    elif feature == "FPFH":  # |correlation| in (0, 1], f32 values
        FD = rng.uniform(0.05, 0.5, (ks, kt)).astype(np.float32).astype(np.float64)
        FD[np.arange(perm.size), perm] = rng.uniform(0.8, 1.0, perm.size).astype(np.float32)
    f, c = getattr(O, feature), getattr(O, corr)
    P = O.default_params(f, c, 6, 0.6, 1.5, bbx, max_iter=8)
    ro = O.register(P, kpS, kpT, FD, want_matchlist=True)
    assert ro["iters"] >= 2, ro["iters"]
    kp = kpS.copy()
    RMS, FDM, FDstd, p1, p2 = 99999.0, 0.0, 0.0, P.para1, P.para2
    for it in range(ro["iters"]):
        tr = ro["trace"][it]
        rr = O.ref_iter_step(kp, kpT, FD, f, c, bbx, it, RMS, FDM, FDstd, p1, p2)
        assert rr["penalty"] == tr["penalty"], (it, rr["penalty"], tr["penalty"])
        ml = ro["matchlist"][it]
        assert len(rr["SP"]) == tr["cor"] == int((ml >= 0).sum())
        assert {int(s): int(t) for s, t in zip(rr["SP"], rr["TP"])} == {int(i): int(ml[i]) for i in np.flatnonzero(ml >= 0)}
        if tr["cor"] > 0:
            assert rr["rmse"] == tr["rmse"]
            if FD is not None:
                assert rr["fdm"] == tr["fdm"] and rr["fdstd"] == tr["fdstd"]
        if corr == "KM":
            assert rr["energy"] == tr["energy"]
        # the scalar tail of transformestimation (ghicp_reg.cpp:795-799, 870-914) on the iteration's 4x4 and updated correspondences
        if tr["cor"] > 0:
            sp = np.array([int(v) for v in rr["SP"]]); tp = np.array([int(v) for v in rr["TP"]])
            iou, rmse_after, conv = O.ref_te_tail(tr["Rt"], _apply(kp[sp], tr["Rt"]), kpT[tp], ks, kt, P.min_cor, P.converge_t, P.converge_r)
            assert iou == tr["iou"] and rmse_after == tr["rmse_after"] and conv == tr["converged"], (it, iou, rmse_after, conv, tr)
        a1, a2 = O.ref_adjustweight(P.est_iou, tr["iou"], P.adjust_ratio, P.adjust_step, p1, p2)
        assert (a1, a2) == (tr["para1"], tr["para2"])
        RMS, FDM, FDstd, p1, p2 = tr["rmse"], tr["fdm"], tr["fdstd"], tr["para1"], tr["para2"]
        kp = _apply(kp, tr["Rt"])


def test_bbx_magnitude(ref, synth):
    """CloudUtility::getCloudBound (utility.h:153-183) + test/ghicp_main.cpp:93 (f64 bounds, float result) == orc_bbx_magnitude."""
    rng = np.random.default_rng(2)
    for c in (synth.tls_pair(30_000, pair_id=2).source, (rng.normal(size=(1000, 3)) * [50, 3, 0.1]).astype(np.float32), np.ones((1, 3), np.float32)):
        assert np.float32(ref.ref_bbx_magnitude(c)) == np.float32(ref.bbx_magnitude(c))


def test_feature_dump_format(ref, api, tmp_path):
    """stereo_binary_feature.cpp:107-124 written by the reference == ghicp_sbf_write of the product (host code, no GPU), and
    the product reads the reference's file back."""
    O = ref
    g = np.load(os.path.join(HERE, "golden", "frontend.npz"))
    feat = g["feat"][0]
    O.ref_sbf_write(tmp_path / "ref.sbf", feat)
    api.sbf_write(tmp_path / "ours.sbf", feat)
    assert (tmp_path / "ref.sbf").read_bytes() == (tmp_path / "ours.sbf").read_bytes()
    np.testing.assert_array_equal(api.sbf_read(tmp_path / "ref.sbf"), feat)


@pytest.fixture(scope="module")
def ref3(oracle):
    if oracle.ref3_lib() is None:
        pytest.skip("oracle/_ref/libfrontend_ref.so not built (reference sources absent)")
    return oracle


def test_bsc_binarisation_and_flip_variants(ref3, synth):
    """binary_feature_extraction.hpp:463-565 (occupancy + depth / density comparison bits, Q4) and 678-758 (ReArrangeGrid, Q3: the
    294-cell grids of the flip variants) run on the same 147 cells as the oracle's restatement: identical 441-bit strings for all
    variants, with the glibc rand() pattern, the shipped all-zero pattern and random patterns."""
    O = ref3
    rng = np.random.default_rng(9)
    pats = [synth.bsc_pattern_glibc(), synth.bsc_pattern_zero(), rng.integers(0, 49, (49, 2))]
    for t in range(60):
        w = np.where(rng.random(147) < 0.35, 0.0, rng.gamma(1.5, 0.4, 147)).astype(np.float32)  # empty cells and cells around the 0.1 threshold
        w[rng.integers(0, 147, 5)] = np.float32(0.1)
        d = (rng.uniform(0.0, 3.0, 147) * (w > 0)).astype(np.float32)
        for dof in (0, 4, 6):
            pat = pats[t % 3]
            np.testing.assert_array_equal(O.ref_bsc_strings(w, d, dof, pat), O.bsc_strings(w, d, dof, pat), err_msg="t=%d dof=%d" % (t, dof))


def test_bsc_sample_pattern_of_a_fresh_process(ref3, synth, tmp_path, monkeypatch):
    """BSCEncoder(R, 7, build_sample_pattern = true) (bfe:62-117, contain2DPair 855-872) after srand(1) -- rand() of a fresh
    process -- draws exactly the 49 pairs SURVEY.md Q2 lists and the bench / tests use; it writes sample_pattern.txt like the reference."""
    monkeypatch.chdir(tmp_path)
    pat = ref3.ref_bsc_pattern()
    np.testing.assert_array_equal(pat, np.asarray(synth.bsc_pattern_glibc()).reshape(49, 2))
    np.testing.assert_array_equal(np.loadtxt(tmp_path / "sample_pattern.txt", dtype=np.int32), pat)


def test_voxelfilter_and_prune(ref3, synth):
    """CFilter::voxelfilter (filter.hpp:28-88: Q1 phantom entries, voxel-key order) and pruneUnstablePoints
    (keypoint_detect.hpp:132-147) of the reference itself against the restatement."""
    O = ref3
    rng = np.random.default_rng(21)
    # (a) at most one point per voxel (input point 0 anchors the lattice at the origin): the kept points and their order are fully
    # determined -> identical sequences; point 0 shares the min-corner voxel with the N phantom entries (Q1), so it leads the output
    cells = rng.choice(np.arange(1, 40 * 40 * 10), 3000, replace=False)
    pts = (np.stack([cells // 400, (cells // 10) % 40, cells % 10], 1) + rng.uniform(0.25, 0.75, (3000, 3))).astype(np.float32) * np.float32(0.5)
    pts = np.concatenate([np.zeros((1, 3), np.float32), pts])
    keep = O.voxel_filter(pts, 0.5)
    out = O.ref_voxelfilter(pts, 0.5)
    assert keep[0] == 0 and out.shape[0] == keep.size == 3001
    np.testing.assert_array_equal(out, pts[keep])
    # the same cloud without the anchor: the min corner voxel is empty, the phantom entries add ONE extra copy of input point 0 in front
    keep = O.voxel_filter(pts[1:], 0.5)
    out = O.ref_voxelfilter(pts[1:], 0.5)
    assert keep[0] == 0 and out.shape[0] == keep.size
    assert np.array_equal(out[0], pts[1]) and np.array_equal(pts[1:][keep[0]], pts[1])
    # (b) a real scan (many points per voxel): std::sort is not stable, so only the NUMBER of voxels and the voxel sequence are
    # defined by the reference; the restatement keeps the lowest input index of every voxel (documented deviation)
    scan = synth.tls_pair(60_000, pair_id=3).target
    keep = O.voxel_filter(scan, 0.1)
    out = O.ref_voxelfilter(scan, 0.1)
    assert out.shape[0] == keep.size
    mn = scan[:, :3].min(0)
    vox = lambda p: np.floor((p[:, :3] - mn) * np.float32(1.0 / np.float32(0.1))).astype(np.int64)
    np.testing.assert_array_equal(vox(out)[1:], vox(scan[keep])[1:])
    # prune: float ratios, NaN (< 3 neighbours: zero eigenvalues) rejected, ptNum > min
    lam = np.abs(rng.normal(size=(5000, 3))).astype(np.float32)
    lam.sort(axis=1)
    lam = lam[:, ::-1].copy()
    lam[::17] = 0.0
    cnt = rng.integers(0, 60, 5000).astype(np.int32)
    np.testing.assert_array_equal(O.ref_prune(lam, cnt, 0.65, 20), O.prune(lam, cnt, 0.65, 20))


def test_non_maxima_suppression_logic(ref3, synth):
    """keypoint_detect.hpp:149-191 itself (std::sort by curvature, std::set of unvisited ranks, erase within the radius).  (The radius
    search under it is the stand-in KdTreeFLANN of oracle/ref_stubs -- exact, strict d^2 < r^2 -- so what is pinned is the suppression
    logic, not FLANN.)
    (1) distinct curvatures: the same keypoints in the same order as the restatement.
    (2) candidates of real scans: neighbouring points with identical neighbourhoods have EQUAL curvature (a few dozen exact ties per
        cloud); std::sort is not stable, so which of two tied points the reference keeps is decided by libstdc++'s introsort, while the
        restatement (and the GPU) break ties by the lower index.  Pinned here: the curvature sequence of the kept points
        is identical, and every position where the point differs is such a tie."""
    O = ref3
    rng = np.random.default_rng(33)
    pts = (rng.uniform(0, 30, (6000, 3)) * np.array([1, 1, 0.2])).astype(np.float32)
    curv = rng.permutation(6000) / 6000.0
    cand = np.sort(rng.choice(6000, 4000, replace=False)).astype(np.int32)
    np.testing.assert_array_equal(O.ref_nms(pts, curv, cand, 1.5), O.nms(pts, curv, cand, 1.5))
    ties = 0
    for pid, hits in ((1, 120_000), (4, 80_000)):
        scan = synth.tls_pair(hits, pair_id=pid).source
        ds = scan[O.voxel_filter(scan, 0.1)]
        lam, cv, cnt = O.pca(ds, 0.5)
        cand = O.prune(lam, cnt)
        assert cand.size > 500
        a, b = O.ref_nms(ds, cv, cand, 1.5), O.nms(ds, cv, cand, 1.5)
        assert a.size == b.size
        np.testing.assert_array_equal(cv[a], cv[b])
        ties += int((a != b).sum())
    assert ties <= 6  # one tied pair on this data
    assert O.ref_nms(np.zeros((3, 3), np.float32), np.zeros(3), np.zeros(0, np.int32), 1.5).size == 0


def test_cubic_grid_contract_against_the_reference_arithmetic(ref3, synth):
    """constructCubicGrid (binary_feature_extraction.hpp:196-373) ITSELF -- f64 point_num of expf terms, f32 running sum of depth x
    expf in the order of the 2-D radius search (ascending distance; stand-in exact KdTreeFLANN) -- on the LCS neighbourhoods of real
    keypoints, against the restatement's contract N2 / N4 (f64 sums rounded once, expf = correctly rounded exp).  This QUANTIFIES the
    contract's distance from the reference's own arithmetic rather than pinning it: weights differ only where glibc's expf is not the
    correctly rounded exp, depths by about one ulp, and after the reference's own binarisation at most one keypoint in a hundred
    differs, by one bit."""
    O = ref3
    pat = synth.bsc_pattern_glibc()
    scan = synth.tls_pair(150_000, pair_id=6).target
    ds = scan[O.voxel_filter(scan, 0.1)]
    kp, _ = O.keypoints(ds, 0.5, 1.5)
    assert kp.size >= 120
    wdiff = bits = differing = 0
    ulps = []
    for p in kp[:120]:
        loc, w, d = O.bsc_cells(ds, p, 1.5, pat)
        rw, rd = O.ref_cubic_grid(loc, 1.5)
        wdiff += int((rw != w).sum())
        ulps.append(np.abs(rd.view(np.int32).astype(np.int64) - d.view(np.int32).astype(np.int64))[np.abs(d) > 1e-3])
        b = int(np.unpackbits(O.bsc_strings(w, d, 6, pat) ^ O.ref_bsc_strings(rw, rd, 6, pat)).sum())
        bits += b
        differing += b > 0
    ulps = np.concatenate(ulps)
    assert wdiff <= 0.002 * 120 * 147, wdiff
    assert np.median(ulps) <= 2 and np.percentile(ulps, 99) <= 64
    assert differing <= 3 and bits <= 4, (differing, bits)


def test_curvature_formula_keyfpfh_and_overlap(ref3, synth):
    """Three small plain-C++ fragments compiled from where they lie: CalculatePcaFeature's curvature (pca.h:228-239; float eigenvalues
    into double fields, 0 when they sum to 0), FPFHfeature::keyfpfh (fpfh.hpp:93-115) and CRegistration::calOverlap's counting loop and
    (0.01 + count) / n ratio (common_reg.cpp:302-313, over the stand-in exact radius search)."""
    O = ref3
    scan = synth.tls_pair(60_000, pair_id=8).target
    ds = scan[O.voxel_filter(scan, 0.1)]
    lam, curv, cnt = O.pca(ds, 0.5)
    ok = np.isfinite(lam).all(axis=1)
    np.testing.assert_array_equal(O.ref_pca_curvature(lam[ok][:5000]), curv[ok][:5000])
    assert O.ref_pca_curvature(np.zeros((1, 3), np.float32))[0] == 0.0
    rng = np.random.default_rng(12)
    hist = rng.random((400, 33)).astype(np.float32)
    kp = rng.choice(400, 57, replace=False).astype(np.int32)
    np.testing.assert_array_equal(O.ref_keyfpfh(hist, kp), hist[kp])
    a = ds[:1500]
    b = (ds[:2500] + np.float32(0.03)).astype(np.float32)[500:]
    for thr in (0.02, 0.06, 0.3):
        assert O.ref_cal_overlap(a, b, thr) == O.cal_overlap(a, b, thr)


def test_adaptive_keypoint_loop(ref3, synth):
    """keypointDetectionBasedOnCurvature_adaptive (keypoint_detect.hpp:53-111): the threshold loop after the PCA call, compiled from the
    reference with its literal 50000 / 5000 made parameters (the loop is never entered below 50000 keypoints), against the restatement's
    loop on the SAME PCA features.  Exact curvature ties (the reference's unstable sort decides them, see the NMS pin) are removed by an
    index-proportional perturbation far below the curvature spacing, so that the loop logic -- float ratio stepped by the double
    constants 0.05 / 0.025, the float-against-double floor test `ratioMax >= 0.65`, the one step back up -- is what is compared."""
    O = ref3
    scan = synth.tls_pair(80_000, pair_id=9).target
    ds = scan[O.voxel_filter(scan, 0.1)]
    lam, curv, cnt = O.pca(ds, 0.5)
    curv = np.where(np.isfinite(curv), curv, 0.0) + np.arange(curv.size) * 1e-13
    plain = O.adaptive_tail(ds, lam, cnt, curv, 0.6, 0.9, upper=10 ** 9, lower=0)
    entered = 0
    for ratio in (0.9, 0.8, 0.7):
        for upper, lower in ((10 ** 6, 10), (plain.size - 1, plain.size // 2), (plain.size // 3, plain.size // 4), (plain.size // 2, plain.size // 2 - 5), (5, 2)):
            ko = O.adaptive_tail(ds, lam, cnt, curv, 0.6, ratio, upper=upper, lower=lower)
            kr = O.ref_adaptive_tail(ds, lam, cnt, curv, 0.6, ratio, upper=upper, lower=lower)
            np.testing.assert_array_equal(kr, ko, err_msg=str((ratio, upper, lower)))
            entered += int(ko.size != plain.size)
    assert entered >= 6


def test_weighted_covariance_contract_against_the_reference_arithmetic(ref3, synth):
    """computeEigenVectorsByWeightPCA up to its eigen solver (binary_feature_extraction.hpp:947-989) ITSELF -- f64 centroid, weight
    sqrt(2) R - distance, nine FLOAT running sums in the radius search's order, division by the weight sum -- on the sqrt(3) R
    neighbourhoods of real keypoints, against the contract N2 (six f64 sums rounded once onto the f32 grid of the matrix scale).  Like the
    cubic-grid test this QUANTIFIES the contract's distance from the reference's arithmetic: measured 83 ulp of the matrix scale at worst
    (float running sums over the 1000-4000 neighbours of a keypoint), and the principal axis the LCS is built from turns by 1.6e-5 rad."""
    from scipy.spatial import cKDTree

    O = ref3
    scan = synth.tls_pair(150_000, pair_id=6).target
    ds = scan[O.voxel_filter(scan, 0.1)]
    kp, _ = O.keypoints(ds, 0.5, 1.5)
    assert kp.size >= 100
    tree = cKDTree(ds.astype(np.float64))
    R = 1.5
    worst_ulp, worst_angle = 0.0, 0.0
    for p in kp[:100]:
        idx = np.array(tree.query_ball_point(ds[p].astype(np.float64), np.sqrt(3.0) * R), np.int64)
        d2 = ((ds[idx] - ds[p]).astype(np.float32) ** 2)
        d2 = (d2[:, 0] + d2[:, 1]) + d2[:, 2]
        idx = idx[np.lexsort((idx, d2))].astype(np.int32)  # ascending distance, ties by index: the radius search's order
        co, cr = O.weighted_cov(ds, idx, int(p), R), O.ref_weighted_cov(ds, idx, int(p), R)
        assert cr is not None
        scale = float(np.abs(cr).max())
        worst_ulp = max(worst_ulp, float(np.abs(co.astype(np.float64) - cr.astype(np.float64)).max() / (scale * 2.0 ** -23)))
        wo, vo = np.linalg.eigh(co.astype(np.float64))
        wr, vr = np.linalg.eigh(0.5 * (cr + cr.T).astype(np.float64))
        worst_angle = max(worst_angle, float(np.arccos(min(1.0, abs(vo[:, 2] @ vr[:, 2])))))
    assert worst_ulp <= 256 and worst_angle < 1e-4, (worst_ulp, worst_angle)
    assert O.ref_weighted_cov(ds, np.array([0, 1], np.int32), 0, R) is None  # fewer than 3 neighbours: no LCS (bfe:947)


def test_local_coordinate_system_steps_after_the_eigen_solver(ref3, synth):
    """computeEigenVectorsByWeightPCA (bfe:939-1035) and computeLocalCoordinateSystem (bfe:119-155) THEMSELVES, compiled over a stand-in
    Eigen::EigenSolver that returns injected eigenpairs: what they do with the solver's output -- largest / smallest eigenvalue by strict
    compares (first index on ties), principal / normal direction, middle = principal x normal, x = principal, y = middle, z = x x y taken
    BEFORE x and y are normalised, origin = the keypoint -- is bit for bit what the restatement does with the same eigenpairs.  (The solver's
    own output -- values, order, signs -- is library arithmetic and stays unpinned.)"""
    from scipy.spatial import cKDTree

    O = ref3
    scan = synth.tls_pair(120_000, pair_id=6).target
    ds = scan[O.voxel_filter(scan, 0.1)]
    kp, _ = O.keypoints(ds, 0.5, 1.5)
    assert kp.size >= 60
    tree = cKDTree(ds.astype(np.float64))
    R = 1.5
    for p in kp[:60]:
        idx = np.array(tree.query_ball_point(ds[p].astype(np.float64), np.sqrt(3.0) * R), np.int64)
        d2 = ((ds[idx] - ds[p]).astype(np.float32) ** 2)
        idx = idx[np.lexsort((idx, (d2[:, 0] + d2[:, 1]) + d2[:, 2]))].astype(np.int32)
        axes, vals, vecs = O.lcs_from_cov(O.weighted_cov(ds, idx, int(p), R))
        got = O.ref_lcs(ds, idx, int(p), R, vals, vecs)
        np.testing.assert_array_equal(got[:3], axes)
        np.testing.assert_array_equal(got[3], ds[p])
    # ties and permuted spectra: every order of three eigenvalues, and two equal ones, through both sides
    rng = np.random.default_rng(3)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    idx = np.arange(10, dtype=np.int32)
    for vals in ([3, 2, 1], [1, 2, 3], [2, 3, 1], [2, 2, 1], [1, 2, 2], [2, 1, 2], [5, 5, 5]):
        vals = np.array(vals, np.float32)
        vecs = q.astype(np.float32)
        got = O.ref_lcs(ds, idx, 0, R, vals, vecs)
        imax, imin = int(np.argmax(vals)), int(np.argmin(vals))  # numpy's argmax / argmin = first index on ties = strict compares
        P, N = vecs[:, imax], vecs[:, imin]
        mid = np.cross(P, N).astype(np.float32)
        np.testing.assert_allclose(got[0], P / np.linalg.norm(P), rtol=1e-6)
        np.testing.assert_allclose(got[1], mid / np.linalg.norm(mid) if np.linalg.norm(mid) > 0 else got[1], rtol=1e-5, atol=1e-7)
