"""CPU checks of the fine-registration restatement (oracle/icp_oracle.inc: CRegistration::icp_reg / ptplicp_reg /
calOverlap / invTransform, reference src/common_reg.cpp:45-199, 294-370) against independent implementations:
scipy's KD-tree for the neighbour queries, numpy for normals and the closed-form inverse quirk."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from conftest import rot_err, trans_err


def small_pair(synth, n=6000, seed=3, deg=2.0, t=(0.10, -0.05, 0.02), keep=0.8, noise=0.003):
    """Target = down-sampled ray-cast scene; source = a rigidly displaced, sub-sampled, noisy copy (T = R S + t)."""
    tgt = synth.tls_pair(40_000, config_id=2, pair_id=seed).target[:, :3]
    rng = np.random.default_rng(seed)
    tgt = tgt[rng.permutation(len(tgt))[:n]].astype(np.float32)
    a = np.deg2rad(deg)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    src = ((tgt.astype(np.float64) - np.array(t)) @ R).astype(np.float32)
    src = src[rng.permutation(len(src))[: int(keep * n)]]
    src = (src + rng.normal(0, noise, src.shape)).astype(np.float32)
    gt = np.eye(4)
    gt[:3, :3] = R
    gt[:3, 3] = t
    return src, tgt, gt


def test_nn1_matches_kdtree(oracle, synth):
    src, tgt, _ = small_pair(synth)
    far = np.array([[500.0, -300.0, 80.0], [-1000.0, 0.0, 0.0]], np.float32)  # queries far outside the target's box
    q = np.vstack([src, far])
    idx, d2 = oracle.nn1(q, tgt)
    dist, ref = cKDTree(tgt.astype(np.float64)).query(q.astype(np.float64))
    diff = q - tgt[idx]
    np.testing.assert_array_equal(d2, (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2])
    same = idx == ref
    # where the KD-tree disagrees the two candidates must be equidistant in float arithmetic
    assert same.mean() > 0.999
    np.testing.assert_allclose(np.sqrt(d2[~same].astype(np.float64)), dist[~same], rtol=1e-6)


def test_cal_overlap_matches_ball_query(oracle, synth):
    src, tgt, _ = small_pair(synth)
    for r in (0.05, 0.3):
        cnt = sum(1 for nb in cKDTree(tgt.astype(np.float64)).query_ball_point(src.astype(np.float64), r * (1 - 1e-7)) if nb)
        got = oracle.cal_overlap(src, tgt, r)
        assert abs(got - np.float32((0.01 + cnt) / len(src))) <= 2.0 / len(src)  # boundary d == r may round either way
    assert oracle.cal_overlap(src, src, 0.01) == np.float32((0.01 + len(src)) / len(src))  # every point finds itself


def test_inv_transform_is_the_reference_quirk(oracle):
    T = np.eye(4, dtype=np.float32)
    a = 0.3
    T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    T[:3, 3] = [1, 2, 3]
    inv = oracle.inv_transform(T)
    np.testing.assert_array_equal(inv[:3, :3], T[:3, :3].T)
    np.testing.assert_array_equal(inv[:3, 3], -T[:3, 3])  # common_reg.cpp:361-363: NOT -R^T t
    np.testing.assert_array_equal(inv[3], [0, 0, 0, 1])


def test_knn_normals(oracle, synth):
    _, tgt, _ = small_pair(synth, n=3000)
    k = 12
    nrm = oracle.knn_normals(tgt, k)
    _, nb = cKDTree(tgt.astype(np.float64)).query(tgt.astype(np.float64), k)
    bad = 0
    for i in range(0, len(tgt), 7):
        P = tgt[nb[i]].astype(np.float64)
        w, v = np.linalg.eigh(np.cov(P.T, bias=True))
        if w[1] - w[0] < 1e-3 * w[2]:
            continue  # ambiguous smallest eigenvector
        n = v[:, 0]
        if np.dot(n, -tgt[i]) < 0:
            n = -n
        if np.abs(nrm[i] - n).max() > 2e-3:
            bad += 1
    assert bad <= 2
    np.testing.assert_allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-5)
    np.testing.assert_array_equal(oracle.knn_normals(tgt[:2], 5), np.full((2, 3), 0.577, np.float32))  # CheckNormals


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("reciprocal,trimmed", [(False, False), (False, True), (True, True)])
def test_icp_recovers_the_displacement(oracle, synth, metric, reciprocal, trimmed):
    src, tgt, gt = small_pair(synth)
    r = oracle.icp(src, tgt, oracle.icp_params(40, reciprocal, trimmed, metric, 0.2, 0.1, 12), want_trace=True)
    assert r["done"] == 1 and r["converged"] == 1 and 1 <= r["iterations"] <= 40
    assert rot_err(r["T"].astype(np.float64), gt) < 2e-3 and trans_err(r["T"].astype(np.float64), gt) < 0.02
    # final_transformation_ is the ordered product of the per-iteration transforms
    acc = np.eye(4, dtype=np.float32)
    for T in r["trace"]:
        acc = (T.astype(np.float32) @ acc).astype(np.float32)
    np.testing.assert_allclose(acc, r["T"], atol=2e-6)
    np.testing.assert_allclose(r["transformed"], src @ r["T"][:3, :3].T + r["T"][:3, 3], atol=1e-4)
    if trimmed:
        assert 0 < r["overlap"] <= 1.01 and r["correspondences"] <= int(np.floor(np.float32(r["overlap"]) * np.float32(len(src)))) + 0
    else:
        assert r["correspondences"] == len(src)
    assert r["fitness"] < 1e-3


def test_icp_refuses_low_overlap_and_counts_iterations(oracle, synth):
    src, tgt, _ = small_pair(synth)
    r = oracle.icp(src + np.float32(500.0), tgt, oracle.icp_params(10, False, True, 0, 0.2, 0.5))
    assert r["done"] == 0 and r["overlap"] < 0.01
    r = oracle.icp(src, tgt, oracle.icp_params(2, False, False, 0))
    assert r["iterations"] == 2 and r["reason"] == 1  # CONVERGENCE_CRITERIA_ITERATIONS


def test_adaptive_keypoints_restatement(oracle, synth):
    """keypointDetectionBasedOnCurvature_adaptive (keypoint_detect.hpp:53-111): the loop only moves the ratio threshold, so
    its result must equal the plain detector run at the reported threshold, and the thresholds follow the reference's
    float arithmetic (-= 0.05 per round, += 0.025 once when the count drops below the lower bound)."""
    p = synth.tls_pair(60_000)
    ds = p.target[oracle.voxel_filter(p.target, 0.1)]
    plain, _ = oracle.keypoints(ds, 0.5, 0.6, 0.9)
    assert plain.size > 120
    # never enters the loop: same as the plain detector
    kp, ru, nr = oracle.keypoints_adaptive(ds, 0.5, 0.6, 0.9, upper=10 ** 6, lower=10)
    np.testing.assert_array_equal(kp, plain)
    assert nr == 0 and ru == np.float32(0.9)
    # enters the loop
    kp, ru, nr = oracle.keypoints_adaptive(ds, 0.5, 0.6, 0.9, upper=plain.size - 1, lower=plain.size // 2)
    assert nr >= 1
    expect = np.float32(0.9)
    path = []
    for _ in range(nr):
        path.append(expect)
        expect = np.float32(np.float64(expect) - 0.05)
    assert ru in (expect, np.float32(np.float64(path[-1]) + 0.025)) or ru == expect
    again, _ = oracle.keypoints(ds, 0.5, 0.6, float(ru))
    np.testing.assert_array_equal(kp, again)
    assert kp.size <= plain.size
