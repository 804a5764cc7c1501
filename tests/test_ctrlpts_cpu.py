"""Closed-form control-point solvers of the reference's CRegistration (src/common_reg.cpp:425-888) as provided by
include/common_reg.h (host arithmetic, no GPU): compared with numpy least squares / Kabsch on seeded control points, plus
the host-side rigid solve that shares its source with the kernels (gh-icp_amd/csrc/devmath.h) against the CPU restatement."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("ctrl") / "test_ctrlpts"
    libdir = os.path.join(ROOT, "gh-icp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_ctrlpts.cpp"),
                           "-L", libdir, "-lghicp_hip", "-Wl,-rpath," + libdir, "-o", str(out)])
    return str(out)


def run(exe, tmp_path, A, B, cp, theta0):
    path = tmp_path / "pts.txt"
    with open(path, "w") as f:
        f.write("%d %d %.17g\n" % (len(A), cp, theta0))
        for a, b in zip(A, B):
            f.write(" ".join("%.17g" % v for v in list(a) + list(b)) + "\n")
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l.split() and l.split()[0] in ("C4", "C7", "LLS", "SVD", "FEW")}


def test_control_point_solvers(exe, tmp_path):
    rng = np.random.default_rng(11)
    n, cp = 12, 8
    A = rng.uniform(-50, 50, (n, 3))
    yaw, s = np.deg2rad(17.0), 1.0003
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    t = np.array([12.5, -3.25, 0.75])
    B = s * A @ R.T + t + rng.normal(0, 0.002, (n, 3))
    r = run(exe, tmp_path, A, B, cp, 15.0)
    # --- CSTRAN_4DOF: B_xy = [[a,-b],[b,a]] A_xy + t (common_reg.cpp:452-470)
    M = np.zeros((2 * cp, 4))
    M[0::2] = np.c_[np.ones(cp), np.zeros(cp), A[:cp, 0], -A[:cp, 1]]
    M[1::2] = np.c_[np.zeros(cp), np.ones(cp), A[:cp, 1], A[:cp, 0]]
    x = np.linalg.lstsq(M, B[:cp, :2].reshape(-1), rcond=None)[0]
    sc = np.hypot(x[2], x[3])
    got = np.array([float(v) for v in r["C4"][1:6]])
    assert r["C4"][0] == "1"
    np.testing.assert_allclose(got, [x[0], x[1], sc, x[3] / sc, x[2] / sc], rtol=1e-9, atol=1e-9)
    pred = sc * (A[cp:, :2] @ np.array([[x[2], x[3]], [-x[3], x[2]]]) / sc) + x[:2]
    np.testing.assert_allclose(float(r["C4"][7]), np.sqrt(((pred - B[cp:, :2]) ** 2).sum(1).mean()), rtol=1e-8)
    assert abs(sc - s) < 1e-3 and abs(np.arctan2(x[3], x[2]) - yaw) < 1e-4
    # --- CSTRAN_7DOF: small-angle similarity (common_reg.cpp:538-575)
    M = np.zeros((3 * cp, 7))
    ax, ay, az = A[:cp].T
    M[0::3] = np.c_[np.ones(cp), np.zeros(cp), np.zeros(cp), np.zeros(cp), -az, ay, ax]
    M[1::3] = np.c_[np.zeros(cp), np.ones(cp), np.zeros(cp), az, np.zeros(cp), -ax, ay]
    M[2::3] = np.c_[np.zeros(cp), np.zeros(cp), np.ones(cp), -ay, ax, np.zeros(cp), az]
    x7 = np.linalg.lstsq(M, B[:cp].reshape(-1), rcond=None)[0]
    np.testing.assert_allclose([float(v) for v in r["C7"][1:8]], x7, rtol=1e-8, atol=1e-8)
    # --- LLS_4DOF: Gauss-Newton on the yaw (common_reg.cpp:660-700) ends at the least-squares yaw / translation
    T = np.array([float(v) for v in r["LLS"][1:17]]).reshape(4, 4)
    th = np.arctan2(T[1, 0], T[0, 0])
    assert r["LLS"][0] == "1" and abs(th - yaw) < 2e-4  # the unit-scale model absorbs s = 1.0003 into a small residual
    np.testing.assert_allclose(T[:3, :3], [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], atol=1e-12)
    Rz = T[:3, :3]
    res = B[:cp] - A[:cp] @ Rz.T
    np.testing.assert_allclose(T[:3, 3], res.mean(0), atol=1e-6)  # stationary point: t = mean residual
    g = ((A[:cp, :2] @ np.array([[-np.sin(th), np.cos(th)], [-np.cos(th), -np.sin(th)]])) * (res[:, :2] - T[:2, 3])).sum()
    assert abs(g) < 1e-5  # d(cost)/d(theta) = 0
    # --- SVD_6DOF: float Umeyama == numpy Kabsch to float rounding
    T6 = np.array([float(v) for v in r["SVD"][1:17]]).reshape(4, 4)
    Ac, Bc = A[:cp] - A[:cp].mean(0), B[:cp] - B[:cp].mean(0)
    U, _, Vt = np.linalg.svd(Bc.T @ Ac)
    D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
    Rk = U @ D @ Vt
    assert r["SVD"][0] == "1"
    np.testing.assert_allclose(T6[:3, :3], Rk, atol=5e-6)
    np.testing.assert_allclose(T6[:3, 3], B[:cp].mean(0) - Rk @ A[:cp].mean(0), atol=5e-4)
    assert r["FEW"] == ["0", "0", "0", "0"]  # too few control points: every solver refuses (common_reg.cpp:440, 531, 645, 803)


def test_host_rigid_solve_shares_the_kernel_arithmetic(api, oracle):
    """ghicp_rigid_svd_host compiles gh_quant_grid / gh_kabsch / gh_jacobi3 of csrc/devmath.h for the host: on the CPU it must
    reproduce the restatement bit for bit -- the same functions run inside k_solve / k_rigid_svd / k_icp_step on the GPU."""
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 7, 200, 5000):
        S = rng.normal(0, 20, (n, 3))
        a, b = rng.uniform(-3, 3, 2)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) @ np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        T = S @ R.T + rng.uniform(-5, 5, 3) + rng.normal(0, 0.05, (n, 3))
        np.testing.assert_array_equal(api.rigid_svd_host(S, T), oracle.rigid_svd(S, T))
    # degenerate input: coincident points (zero covariance) and reflections still give a proper rotation
    P = np.tile([[1.0, 2.0, 3.0]], (5, 1))
    g = api.rigid_svd_host(P, P + 1.0)
    np.testing.assert_array_equal(g, oracle.rigid_svd(P, P + 1.0))
    assert abs(np.linalg.det(g[:3, :3]) - 1) < 1e-6
    with pytest.raises(api.GhicpError):
        api.rigid_svd_host(np.zeros((0, 3)), np.zeros((0, 3)))
