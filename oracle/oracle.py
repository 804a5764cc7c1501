"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
the product package.  See oracle/ghicp_oracle.cpp for what each entry restates (file:line).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BSC, ROPS, FPFH, NONE = 0, 1, 2, 3  # utility.h:51-57
NN, NNR, KM = 0, 1, 2  # utility.h:59-64


class Params(C.Structure):
    _fields_ = [("feature", C.c_int), ("corr", C.c_int), ("dof", C.c_int), ("max_iter", C.c_int),
                ("radius_nonmax", C.c_float), ("adjust_ratio", C.c_float), ("adjust_step", C.c_float),
                ("est_iou", C.c_float), ("converge_t", C.c_float), ("converge_r", C.c_float),
                ("bbx_magnitude", C.c_float), ("pad_", C.c_float),
                ("penalty_initial", C.c_double), ("para1", C.c_double), ("para2", C.c_double),
                ("km_eps", C.c_double), ("min_cor", C.c_int), ("weight_changing_rate", C.c_int)]


class Iter(C.Structure):
    _fields_ = [("cor", C.c_int), ("converged", C.c_int)] + [
        (k, C.c_double) for k in ("penalty", "cdmean", "cdstd", "rmse", "rmse_after", "fdm", "fdstd",
                                  "iou", "para1", "para2", "energy")] + [("Rt", C.c_double * 16)]


def default_params(feature=NONE, corr=NN, dof=6, est_iou=0.6, radius_nonmax=1.5, bbx_magnitude=100.0,
                   adjust_ratio=1.1, adjust_step=0.1, max_iter=200) -> Params:
    """Reference defaults: README.md:79-85, ghicp_reg.h:32-40,80."""
    return Params(feature, corr, dof, max_iter, radius_nonmax, adjust_ratio, adjust_step, est_iou,
                  0.02, 0.02, bbx_magnitude, 0.0, 2.0, 1.0, 1.0, 0.01, 10, 6)


def build(force: bool = False) -> None:
    so = os.path.join(_HERE, "libghicp_oracle.so")
    src = os.path.join(_HERE, "ghicp_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ghicp_oracle.cpp", "icp_oracle.inc", "km_model.inc", "km4_model.inc", "bsc_exp_table.inc")):
        subprocess.check_call(["make", "-C", _HERE, "libghicp_oracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(_HERE, "_ref", "libkm_ref.so")
    if os.path.exists("/root/reference/src/km.cpp") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.path.join(_HERE, "libghicp_oracle.so"))
        _lib.orc_km.restype = C.c_longlong
        _lib.orc_bbx_magnitude.restype = C.c_float
    return _lib


def ref_lib():
    """The reference's own km.cpp (oracle/_ref/libkm_ref.so) or None when it was not built."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libkm_ref.so")
        if not os.path.exists(p):
            try:
                build()
            except Exception:
                pass
        if os.path.exists(p):
            _ref = C.CDLL(p)
    return _ref


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def voxel_filter(xyz, voxel):
    xyz = _f32(xyz)
    keep = np.empty(xyz.shape[0] + 1, dtype=np.int32)
    m = lib().orc_voxel_filter(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], C.c_float(voxel), _p(keep, C.c_int))
    return keep[:m].copy()


def pca(xyz, radius):
    xyz = _f32(xyz)
    m = xyz.shape[0]
    lam = np.zeros((m, 3), np.float32)
    curv = np.zeros(m, np.float64)
    cnt = np.zeros(m, np.int32)
    lib().orc_pca(_p(xyz, C.c_float), m, xyz.shape[1], C.c_float(radius), _p(lam, C.c_float), _p(curv, C.c_double), _p(cnt, C.c_int))
    return lam, curv, cnt


def prune(lam, cnt, ratio_max=0.65, min_n=20):
    lam = _f32(lam)
    cnt = np.ascontiguousarray(cnt, np.int32)
    cand = np.empty(lam.shape[0], np.int32)
    c = lib().orc_prune(_p(lam, C.c_float), _p(cnt, C.c_int), lam.shape[0], C.c_float(ratio_max), min_n, _p(cand, C.c_int))
    return cand[:c].copy()


def nms(xyz, curv, cand, R):
    xyz = _f32(xyz)
    curv = np.ascontiguousarray(curv, np.float64)
    cand = np.ascontiguousarray(cand, np.int32)
    kp = np.empty(max(1, cand.size), np.int32)
    k = lib().orc_nms(_p(xyz, C.c_float), xyz.shape[1], _p(curv, C.c_double), _p(cand, C.c_int), cand.size, C.c_float(R), _p(kp, C.c_int))
    return kp[:k].copy()


def keypoints(xyz, radius, R_nms, ratio_max=0.65, min_n=20):
    xyz = _f32(xyz)
    kp = np.empty(max(1, xyz.shape[0]), np.int32)
    mean_nb = C.c_double(0)
    k = lib().orc_keypoints(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], C.c_float(radius), C.c_float(ratio_max), min_n,
                            C.c_float(R_nms), _p(kp, C.c_int), C.byref(mean_nb))
    return kp[:k].copy(), mean_nb.value


def bsc(xyz, kp, R, dof, pattern):
    xyz = _f32(xyz)
    kp = np.ascontiguousarray(kp, np.int32)
    pattern = np.ascontiguousarray(pattern, np.int32)
    K = kp.size
    feat = np.zeros((4, K, 56), np.uint8)
    lcs = np.zeros((K, 12), np.float32)
    mean_nb = C.c_double(0)
    lib().orc_bsc(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], _p(kp, C.c_int), K, C.c_float(R), dof, _p(pattern, C.c_int),
                  _p(feat, C.c_ubyte), _p(lcs, C.c_float), C.byref(mean_nb))
    return feat, lcs, mean_nb.value


def bsc_binarize(weight, depth, pattern):
    weight, depth = _f32(weight), _f32(depth)
    pattern = np.ascontiguousarray(pattern, np.int32)
    out = np.zeros(56, np.uint8)
    lib().orc_bsc_binarize(_p(weight, C.c_float), _p(depth, C.c_float), weight.size, _p(pattern, C.c_int), _p(out, C.c_ubyte))
    return out


def fpfh(xyz, k=20):
    """pcl NormalEstimation(k) + FPFHEstimation(k) restated: returns (normals (m,3), hist (m,33))."""
    xyz = _f32(xyz)
    m = xyz.shape[0]
    nrm = np.zeros((m, 3), np.float32)
    hist = np.zeros((m, 33), np.float32)
    lib().orc_fpfh(_p(xyz, C.c_float), m, xyz.shape[1], k, _p(nrm, C.c_float), _p(hist, C.c_float))
    return nrm, hist


def set_bsc_exp_libm(on):
    """Test switch: the BSC encoder's Gaussian weight through the host libm's f64 exp rounded once to f32 (the correctly rounded expf up to
    double rounding) instead of the contract's table x polynomial.  Always switch it back off."""
    lib().orc_set_bsc_exp_libm(1 if on else 0)


def bsc_expf(x):
    """N4 of the numerics contract: expf(x), x in [-4.5, 0], of the BSC Gaussian cell weight (orc::contract_bsc_expf)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.float32)
    lib().orc_bsc_expf(_p(x, C.c_float), int(x.size), _p(out, C.c_float))
    return out


def atan2f(y, x):
    """N7 of the numerics contract: the atan2f of the SPFH angle feature (orc::contract_atan2f)."""
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(y.shape, np.float32)
    lib().orc_atan2f(_p(y, C.c_float), _p(x, C.c_float), int(y.size), _p(out, C.c_float))
    return out


def fd_bsc(fS, fT):
    """fS: (V,ks,56) u8, fT: (kt,56) u8 -> (ks,kt) f64"""
    fS = np.ascontiguousarray(fS, np.uint8)
    fT = np.ascontiguousarray(fT, np.uint8)
    V, ks, _ = fS.shape
    kt = fT.shape[0]
    FD = np.zeros((ks, kt), np.float64)
    lib().orc_fd_bsc(_p(fS, C.c_ubyte), ks, V, _p(fT, C.c_ubyte), kt, _p(FD, C.c_double))
    return FD


def fd_fpfh(hS, hT):
    hS, hT = _f32(hS), _f32(hT)
    FD = np.zeros((hS.shape[0], hT.shape[0]), np.float64)
    lib().orc_fd_fpfh(_p(hS, C.c_float), hS.shape[0], _p(hT, C.c_float), hT.shape[0], _p(FD, C.c_double))
    return FD


def km(w, eps=0.01):
    w = np.ascontiguousarray(w, np.float64)
    n = w.shape[0]
    match = np.empty(n, np.int32)
    steps = lib().orc_km(_p(w, C.c_double), n, C.c_double(eps), _p(match, C.c_int))
    return match, steps


def km_reference(w, eps=0.01, penalty=0.0):
    """The reference's own Km::kmsolve (compiled from /root/reference/src/km.cpp)."""
    r = ref_lib()
    if r is None:
        return None
    w = np.ascontiguousarray(w, np.float64)
    n = w.shape[0]
    match = np.empty(n, np.int32)
    r.ref_km_solve(_p(w, C.c_double), n, C.c_double(eps), C.c_double(penalty), _p(match, C.c_int))
    return match


def jacobi3(a):
    a = np.ascontiguousarray(a, np.float64)
    ev = np.zeros(3)
    evec = np.zeros((3, 3))
    lib().orc_jacobi3(_p(a, C.c_double), _p(ev, C.c_double), _p(evec, C.c_double))
    return ev, evec


def rigid_svd(src, tgt):
    src = np.ascontiguousarray(src, np.float64)
    tgt = np.ascontiguousarray(tgt, np.float64)
    Rt = np.zeros(16)
    lib().orc_rigid_svd(_p(src, C.c_double), _p(tgt, C.c_double), src.shape[0], _p(Rt, C.c_double))
    return Rt.reshape(4, 4)


def register(params: Params, kpS, kpT, FD=None, want_matchlist=False):
    """Returns dict(Rt, iters, trace[list of dict], matchlist, km_seconds)."""
    kpS = np.ascontiguousarray(kpS, np.float64)
    kpT = np.ascontiguousarray(kpT, np.float64)
    ks, kt = kpS.shape[0], kpT.shape[0]
    Rt = np.zeros(16)
    trace = (Iter * params.max_iter)()
    ml = np.full((params.max_iter, ks), -2, np.int32) if want_matchlist else None
    fdp = None
    if FD is not None:
        FD = np.ascontiguousarray(FD, np.float64)
        assert FD.shape == (ks, kt)
        fdp = _p(FD, C.c_double)
    kms = C.c_double(0)
    it = lib().orc_register(C.byref(params), _p(kpS, C.c_double), ks, _p(kpT, C.c_double), kt, fdp, _p(Rt, C.c_double), trace,
                            _p(ml, C.c_int) if ml is not None else None, C.byref(kms))
    tr = []
    for i in range(it):
        r = trace[i]
        d = {k: getattr(r, k) for k, _ in Iter._fields_ if k != "Rt"}
        d["Rt"] = np.array(r.Rt[:]).reshape(4, 4)
        tr.append(d)
    return dict(Rt=Rt.reshape(4, 4), iters=it, trace=tr, matchlist=None if ml is None else ml[:it], km_seconds=kms.value)


def transform_cloud(xyz, Rt):
    xyz = _f32(xyz)
    Rt = np.ascontiguousarray(Rt, np.float64)
    out = np.empty((xyz.shape[0], 3), np.float32)
    lib().orc_transform_cloud(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], _p(Rt, C.c_double), _p(out, C.c_float))
    return out


def bbx_magnitude(xyz):
    xyz = _f32(xyz)
    return float(lib().orc_bbx_magnitude(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1]))


# ---------------------------------------------------------------- fine registration (common_reg.cpp), see icp_oracle.inc
P2P, P2PLANE = 0, 1


class IcpParams(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("use_reciprocal", C.c_int), ("use_trimmed", C.c_int), ("metric", C.c_int),
                ("thre_dis", C.c_float), ("min_overlap", C.c_float), ("covariance_k", C.c_int), ("pad_", C.c_int),
                ("transformation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double)]


class IcpStats(C.Structure):
    _fields_ = [("done", C.c_int), ("iterations", C.c_int), ("converged", C.c_int), ("reason", C.c_int),
                ("correspondences", C.c_longlong), ("overlap", C.c_float), ("pad_", C.c_float),
                ("mse", C.c_double), ("fitness", C.c_double)]


def icp_params(max_iter=50, reciprocal=False, trimmed=False, metric=P2P, thre_dis=0.5, min_overlap=0.1, covariance_k=15):
    """common_reg.cpp:78-85: transformation epsilon 1e-8, Euclidean fitness epsilon 1e-5."""
    return IcpParams(max_iter, int(reciprocal), int(trimmed), metric, thre_dis, min_overlap, covariance_k, 0, 1e-8, 1e-5)


def cal_overlap(c1, c2, thre_dis):
    c1, c2 = _f32(c1), _f32(c2)
    f = lib().orc_cal_overlap
    f.restype = C.c_float
    return float(f(_p(c1, C.c_float), c1.shape[0], c1.shape[1], _p(c2, C.c_float), c2.shape[0], c2.shape[1], C.c_float(thre_dis)))


def inv_transform(T):
    T = np.ascontiguousarray(T, np.float32)
    out = np.zeros(16, np.float32)
    lib().orc_inv_transform(_p(T, C.c_float), _p(out, C.c_float))
    return out.reshape(4, 4)


def knn_normals(xyz, k):
    xyz = _f32(xyz)
    out = np.zeros((xyz.shape[0], 3), np.float32)
    lib().orc_knn_normals(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], int(k), _p(out, C.c_float))
    return out


def nn1(q, tgt):
    q, tgt = _f32(q), _f32(tgt)
    idx = np.zeros(q.shape[0], np.int32)
    d2 = np.zeros(q.shape[0], np.float32)
    lib().orc_nn1(_p(q, C.c_float), q.shape[0], q.shape[1], _p(tgt, C.c_float), tgt.shape[0], tgt.shape[1], _p(idx, C.c_int), _p(d2, C.c_float))
    return idx, d2


def icp(src, tgt, params: IcpParams, want_trace=False):
    """Returns dict(done, T (4,4) f32, transformed, stats fields, corr0, trace)."""
    src, tgt = _f32(src), _f32(tgt)
    ns = src.shape[0]
    T = np.zeros(16, np.float32)
    out = np.zeros((ns, 3), np.float32)
    st = IcpStats()
    corr0 = np.full(ns, -2, np.int32)
    tr = np.zeros((max(params.max_iter, 1), 16), np.float32) if want_trace else None
    done = lib().orc_icp(_p(src, C.c_float), ns, src.shape[1], _p(tgt, C.c_float), tgt.shape[0], tgt.shape[1], C.byref(params),
                         _p(T, C.c_float), _p(out, C.c_float), C.byref(st), _p(corr0, C.c_int),
                         _p(tr, C.c_float) if tr is not None else None)
    d = {k: getattr(st, k) for k, _ in IcpStats._fields_ if k != "pad_"}
    d.update(done=int(done), T=T.reshape(4, 4), transformed=out, corr0=corr0,
             trace=None if tr is None else tr[:st.iterations].reshape(-1, 4, 4))
    return d


def keypoints_adaptive(xyz, radius, R_nms, ratio_max=0.65, min_n=20, upper=50000, lower=5000):
    """keypointDetectionBasedOnCurvature_adaptive (keypoint_detect.hpp:53-111). Returns (kp, ratio_used, rounds)."""
    xyz = _f32(xyz)
    kp = np.empty(max(1, xyz.shape[0]), np.int32)
    ru, nr = C.c_float(0), C.c_int(0)
    k = lib().orc_keypoints_adaptive(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], C.c_float(radius), C.c_float(ratio_max), min_n,
                                     C.c_float(R_nms), C.c_longlong(upper), C.c_longlong(lower), _p(kp, C.c_int), C.byref(ru), C.byref(nr))
    return kp[:k].copy(), ru.value, nr.value


def km_model(w, eps=0.01, march=True, sweep_first=False, full=False, flood_dead=False):
    """Sequential model of the GPU solver's state machine (oracle/km_model.inc).  Returns (match, steps, marched,
    failed_phases); with full=True also (rows swept in failed phases, rows swept before an aborted sweep).
    sweep_first=True: rule E10 (an order-free sweep decides the fate of the phase before any DFS); flood_dead=True (with
    sweep_first): prototype E12 (a dead child's reachable set is flooded instead of stepped through); full=True appends
    (rows swept in failed phases, rows swept before an aborted sweep, rows in dead floods, rows in probes of live children)."""
    w = np.ascontiguousarray(w, np.float64)
    n = w.shape[0]
    match = np.empty(n, np.int32)
    st = np.zeros(7, np.int64)
    rc = lib().orc_km_model(_p(w, C.c_double), n, C.c_double(eps), _p(match, C.c_int), _p(st, C.c_longlong),
                            int(march) | (2 if sweep_first else 0) | (4 if flood_dead else 0))
    if rc != 0:
        raise RuntimeError("km_model failed (status %d)" % rc)
    out = (match, int(st[0]), int(st[1]), int(st[2]))
    return out + (int(st[3]), int(st[4]), int(st[5]), int(st[6])) if full else out


def km4_model(w, eps=0.01, cap=3, prune=True, hint=0, exact_rest=False, exact_s=False, seed=False, lazy=False):
    """Rule-level model of the flood-first Kuhn-Munkres kernel k_km4 (oracle/km4_model.inc).  Returns (match, stats) with
    stats = dict(phases, failed, flood_rows, push_rows, rebuild_rows, pull_rounds, dfs_steps, dfs_pops, overflow_rows, seeded, unseeded,
    lazy_none, lazy_unwound); match is None when the model reports the slack hazard (rule R4) -- the kernel then falls back to its
    literal solver.  The kernel's configuration is cap=3, hint=6, exact_rest=True, seed=True, lazy=True (rules R5, R3', R5'); the defaults
    are round 2's."""
    w = np.ascontiguousarray(w, np.float64)
    n = w.shape[0]
    match = np.empty(n, np.int32)
    st = np.zeros(13, np.int64)
    rc = lib().orc_km4_model(_p(w, C.c_double), n, C.c_double(eps), _p(match, C.c_int), _p(st, C.c_longlong),
                             int(cap) | (0 if prune else 0x100) | (0x200 if exact_s else 0) | ((int(hint) & 0x3f) << 10) | (0x10000 if exact_rest else 0) | (0x20000 if seed else 0)
                             | (0x40000 if lazy else 0))
    names = ("phases", "failed", "flood_rows", "push_rows", "rebuild_rows", "pull_rounds", "dfs_steps", "dfs_pops", "overflow_rows", "seeded", "unseeded",
             "lazy_none", "lazy_unwound")
    stats = dict(zip(names, (int(v) for v in st)))
    if rc == 4:
        return None, stats
    if rc != 0:
        raise RuntimeError("km4_model failed (status %d)" % rc)
    return match, stats


# ---------------------------------------------------------------- whole pair (the per-pair half of test/ghicp_main.cpp:86-151)
def build_native() -> str:
    """Compiles the same restatement with -O3 -march=native for the CPU timing legs of bench.py into
    oracle/_native/ (git-ignored; built on the machine that runs it because of -march=native).  Returns the path."""
    out_dir = os.path.join(_HERE, "_native")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libghicp_oracle_native.so")
    srcs = [os.path.join(_HERE, f) for f in ("ghicp_oracle.cpp", "icp_oracle.inc", "km_model.inc", "km4_model.inc", "bsc_exp_table.inc")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w", "-shared",
                               "-o", so, srcs[0]])
    return so


def use_library(path: str) -> None:
    """Routes every wrapper of this module through another build of the same restatement (see build_native)."""
    global _lib
    _lib = C.CDLL(path)
    _lib.orc_km.restype = C.c_longlong
    _lib.orc_bbx_magnitude.restype = C.c_float


def register_pair(S, T, voxel, r_pca, R_nms, dof, feature, corr, est_iou, pattern=None, max_iter=200):
    """main() of the reference for one (source, target) pair: voxel filter -> curvature keypoints -> BSC / FPFH ->
    feature distance -> GH-ICP loop.  Returns dict(Rt, iters, m_s, m_t, k_s, k_t, k_bar, m_bar, seconds{stage: s}, trace)."""
    import time
    sec = {}
    t = time.perf_counter()
    ds = {n: c[voxel_filter(c, voxel)] for n, c in (("S", S), ("T", T))}
    sec["voxel"] = time.perf_counter() - t
    t = time.perf_counter()
    kp, kbar = {}, {}
    for n in ("T", "S"):
        kp[n], kbar[n] = keypoints(ds[n], r_pca, R_nms)
    sec["keypoints"] = time.perf_counter() - t
    t = time.perf_counter()
    mbar = 0.0
    if feature == BSC:
        fT, _, mT = bsc(ds["T"], kp["T"], R_nms, 0, pattern)
        fS, _, mS = bsc(ds["S"], kp["S"], R_nms, dof, pattern)
        mbar = 0.5 * (mT + mS)
        sec["feature"] = time.perf_counter() - t
        t = time.perf_counter()
        FD = fd_bsc(fS[:(4 if dof > 4 else (2 if dof > 0 else 1))], fT[0])
    elif feature == FPFH:
        hT = fpfh(ds["T"])[1][kp["T"]]
        hS = fpfh(ds["S"])[1][kp["S"]]
        sec["feature"] = time.perf_counter() - t
        t = time.perf_counter()
        FD = fd_fpfh(hS, hT)
    else:
        sec["feature"] = 0.0
        FD = None
    sec["fd"] = time.perf_counter() - t
    t = time.perf_counter()
    P = default_params(feature, corr, dof, est_iou, R_nms, bbx_magnitude(ds["S"]), max_iter=max_iter)
    ro = register(P, ds["S"][kp["S"]].astype(np.float64), ds["T"][kp["T"]].astype(np.float64), FD)
    sec["loop"] = time.perf_counter() - t
    sec["total"] = sum(sec.values())
    return dict(Rt=ro["Rt"], iters=ro["iters"], m_s=int(ds["S"].shape[0]), m_t=int(ds["T"].shape[0]), k_s=int(kp["S"].size), k_t=int(kp["T"].size),
                k_bar=0.5 * (kbar["S"] + kbar["T"]), m_bar=mbar, seconds=sec, km_seconds=ro["km_seconds"],
                cor=[tr["cor"] for tr in ro["trace"]], **registration_verdict(ro["trace"], R_nms))


def registration_verdict(trace, R_nms):
    """src/ghicp_reg.cpp:918-924: at convergence the reference prints "Registration Succeed." iff RMSEafter < 1.5 * nonmax."""
    if not trace:
        return dict(converged=0, rmse_after=float("nan"), registered_ok=0)
    last = trace[-1]
    conv = int(last["converged"])
    return dict(converged=conv, rmse_after=float(last["rmse_after"]),
                registered_ok=int(bool(conv) and last["rmse_after"] < 1.5 * float(np.float32(R_nms))))


# ---------------------------------------------------------------- the reference's own code (oracle/_ref/libghicp_ref.so)
_ref2 = None


def ref2_lib():
    """oracle/_ref/libghicp_ref.so: the reference's stereo_binary_feature.cpp, fpfh.hpp distance and the plain-C++ members of
    ghicp_reg.cpp compiled from where they lie (oracle/ghicp_ref_shim.cpp), or None when it was not built."""
    global _ref2
    if _ref2 is None:
        p = os.path.join(_HERE, "_ref", "libghicp_ref.so")
        if not os.path.exists(p) and os.path.exists("/root/reference/src/ghicp_reg.cpp"):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
        if os.path.exists(p):
            _ref2 = C.CDLL(p)
            _ref2.ref_fpfh_distance.restype = C.c_float
            _ref2.ref_bbx_magnitude.restype = C.c_float
    return _ref2


def ref_iter_step(kpS, kpT, FD, feature, corr, bbx, it, RMS, FDM, FDstd, para1, para2):
    """calED -> calCD_* -> findcorrespondence* of the reference itself.  Returns dict(penalty, CD, SP, TP, rmse, fdm, fdstd, energy)."""
    kpS = np.ascontiguousarray(kpS, np.float64)
    kpT = np.ascontiguousarray(kpT, np.float64)
    ks, kt = kpS.shape[0], kpT.shape[0]
    fdp = None
    if FD is not None:
        FD = np.ascontiguousarray(FD, np.float64)
        fdp = _p(FD, C.c_double)
    n = max(ks, kt)
    SP, TP = np.zeros(n, np.int32), np.zeros(n, np.int32)
    CD = np.zeros((ks, kt))
    pen, rmse, fdm, fdstd, en = (C.c_double(0) for _ in range(5))
    cor = ref2_lib().ref_iter_step(_p(kpS, C.c_double), ks, _p(kpT, C.c_double), kt, fdp, int(feature), int(corr), C.c_float(bbx), int(it),
                                   C.c_double(RMS), C.c_double(FDM), C.c_double(FDstd), C.c_double(para1), C.c_double(para2), C.byref(pen),
                                   _p(CD, C.c_double), _p(SP, C.c_int), _p(TP, C.c_int), C.byref(rmse), C.byref(fdm), C.byref(fdstd), C.byref(en))
    return dict(penalty=pen.value, CD=CD, SP=SP[:cor].copy(), TP=TP[:cor].copy(), rmse=rmse.value, fdm=fdm.value, fdstd=fdstd.value, energy=en.value)


def ref_adjustweight(est_iou, iou, ratio, step, para1, para2):
    a, b = C.c_double(para1), C.c_double(para2)
    ref2_lib().ref_adjustweight(C.c_float(est_iou), C.c_double(iou), C.c_float(ratio), C.c_float(step), C.byref(a), C.byref(b))
    return a.value, b.value


def ref_fd_bsc(fS, fT, dof):
    fS = np.ascontiguousarray(fS, np.uint8)
    fT = np.ascontiguousarray(fT, np.uint8)
    V, ks, _ = fS.shape
    kt = fT.shape[0]
    FD = np.zeros((ks, kt))
    ref2_lib().ref_fd_bsc(_p(fS, C.c_ubyte), ks, V, _p(fT, C.c_ubyte), kt, int(dof), _p(FD, C.c_double))
    return FD


def ref_hamming(a, b, nbits=441):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(ref2_lib().ref_hamming(_p(a, C.c_ubyte), _p(b, C.c_ubyte), int(nbits)))


def ref_fpfh_distance(h1, h2):
    h1, h2 = _f32(h1), _f32(h2)
    return float(ref2_lib().ref_fpfh_distance(_p(h1, C.c_float), _p(h2, C.c_float)))


def ref_sbf_write(path, feat):
    feat = np.ascontiguousarray(feat, np.uint8).reshape(-1, 56)
    ref2_lib().ref_sbf_write(str(path).encode(), _p(feat, C.c_ubyte), feat.shape[0])


_ref3 = None


def ref3_lib():
    """oracle/_ref/libfrontend_ref.so: plain-C++ members of the reference's front end (BSC binarisation / re-arrangement / sample
    pattern, CFilter::voxelfilter, pruneUnstablePoints) compiled from where they lie (oracle/frontend_ref_shim.cpp), or None."""
    global _ref3
    if _ref3 is None:
        p = os.path.join(_HERE, "_ref", "libfrontend_ref.so")
        if not os.path.exists(p) and os.path.exists("/root/reference/include/filter.hpp"):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
        if os.path.exists(p):
            _ref3 = C.CDLL(p)
    return _ref3


def bsc_strings(weight147, depth147, dof, pattern):
    """The oracle's strings of one keypoint from its 147 cells -> (4, 56) u8 (orc::bsc_strings, the tail of the BSC encoder)."""
    w, d = _f32(weight147), _f32(depth147)
    pattern = np.ascontiguousarray(pattern, np.int32)
    out = np.zeros((4, 56), np.uint8)
    lib().orc_bsc_strings(_p(w, C.c_float), _p(d, C.c_float), int(dof), _p(pattern, C.c_int), _p(out, C.c_ubyte))
    return out


def ref_bsc_strings(weight147, depth147, dof, pattern):
    w, d = _f32(weight147), _f32(depth147)
    pattern = np.ascontiguousarray(pattern, np.int32)
    out = np.zeros((4, 56), np.uint8)
    ref3_lib().ref_bsc_strings(_p(w, C.c_float), _p(d, C.c_float), int(dof), _p(pattern, C.c_int), _p(out, C.c_ubyte))
    return out


def ref_bsc_pattern():
    out = np.zeros(98, np.int32)
    ref3_lib().ref_bsc_pattern(_p(out, C.c_int))
    return out.reshape(49, 2)


def ref_voxelfilter(xyz, voxel):
    xyz = _f32(xyz)
    out = np.zeros((xyz.shape[0] + 1, 3), np.float32)
    m = ref3_lib().ref_voxelfilter(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], C.c_float(voxel), _p(out, C.c_float))
    return out[:m].copy()


def ref_prune(lam, cnt, ratio_max=0.65, min_n=20):
    lam = _f32(lam)
    cnt = np.ascontiguousarray(cnt, np.int32)
    out = np.zeros(max(1, lam.shape[0]), np.int32)
    k = ref3_lib().ref_prune(_p(lam, C.c_float), _p(cnt, C.c_int), lam.shape[0], C.c_float(ratio_max), int(min_n), _p(out, C.c_int))
    return out[:k].copy()


def ref_nms(xyz, curv, cand, R):
    """The reference's own nonMaximaSuppression over the candidates `cand` of cloud `xyz` -> kept point ids in its output order."""
    xyz = _f32(xyz)
    cand = np.ascontiguousarray(cand, np.int32)
    pts = np.ascontiguousarray(xyz[cand][:, :3], np.float32)
    cv = np.ascontiguousarray(np.asarray(curv, np.float64)[cand])
    out = np.zeros(max(1, cand.size), np.int32)
    k = ref3_lib().ref_nms(_p(pts, C.c_float), _p(cv, C.c_double), _p(cand, C.c_int), cand.size, C.c_float(R), _p(out, C.c_int))
    return out[:k].copy()


def ref_bbx_magnitude(xyz):
    xyz = _f32(xyz)
    return float(ref2_lib().ref_bbx_magnitude(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1]))


def ref_te_tail(Rt, SpA, Tp, ks, kt, min_cor=10, conv_t=0.02, conv_r=0.02):
    """Scalar tail of the reference's transformestimation: returns (IoU, RMSE after, converged)."""
    Rt = np.ascontiguousarray(Rt, np.float64).reshape(16)
    SpA = np.ascontiguousarray(SpA, np.float64).reshape(-1, 3)
    Tp = np.ascontiguousarray(Tp, np.float64).reshape(-1, 3)
    out = np.zeros(3)
    ref2_lib().ref_te_tail(_p(Rt, C.c_double), _p(SpA, C.c_double), _p(Tp, C.c_double), SpA.shape[0], int(ks), int(kt), int(min_cor),
                           C.c_float(conv_t), C.c_float(conv_r), _p(out, C.c_double))
    return float(out[0]), float(out[1]), int(out[2])


def bsc_cells(xyz, point_id, R, pattern):
    """(tests) one keypoint of the oracle's BSC encoder: (neighbourhood in the LCS (n,3) f32, weight (147,), depth (147,))."""
    xyz = _f32(xyz)
    pattern = np.ascontiguousarray(pattern, np.int32)
    loc = np.zeros((xyz.shape[0], 3), np.float32)
    cells = np.zeros(294, np.float32)
    n = lib().orc_bsc_cells(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], int(point_id), C.c_float(R), _p(pattern, C.c_int), _p(loc, C.c_float),
                            xyz.shape[0], _p(cells, C.c_float))
    return loc[:n].copy(), cells[:147].copy(), cells[147:].copy()


def ref_cubic_grid(loc, R):
    loc = _f32(loc)
    cells = np.zeros(294, np.float32)
    ref3_lib().ref_cubic_grid(_p(loc, C.c_float), loc.shape[0], C.c_float(R), _p(cells, C.c_float))
    return cells[:147].copy(), cells[147:].copy()


# ---- round 3: more of the reference's plain C++ behind the same shim (oracle/frontend_ref_shim.cpp)
def weighted_cov(xyz, idx, test_index, R):
    """The contract's weighted covariance of one keypoint neighbourhood (orc::weighted_covariance) -> (3, 3) f32."""
    xyz, idx = _f32(xyz), np.ascontiguousarray(idx, np.int32)
    out = np.zeros(9, np.float32)
    lib().orc_weighted_cov(_p(xyz, C.c_float), xyz.shape[1], _p(idx, C.c_int), int(idx.size), int(test_index), C.c_float(R), _p(out, C.c_float))
    return out.reshape(3, 3)


def ref_weighted_cov(xyz, idx, test_index, R):
    """binary_feature_extraction.hpp:947-989 itself (float running sums in the given order) -> (3, 3) f32, or None for < 3 neighbours."""
    xyz = np.ascontiguousarray(_f32(xyz)[:, :3])
    idx = np.ascontiguousarray(idx, np.int32)
    out = np.zeros(9, np.float32)
    ok = ref3_lib().ref_weighted_cov(_p(xyz, C.c_float), xyz.shape[0], _p(idx, C.c_int), int(idx.size), int(test_index), C.c_float(R), _p(out, C.c_float))
    return out.reshape(3, 3) if ok else None


def ref_pca_curvature(lam):
    """pca.h:228-239 itself for every eigenvalue triplet (m, 3) f32 -> curvature f64."""
    L = ref3_lib()
    L.ref_pca_curvature.restype = C.c_double
    lam = _f32(lam)
    return np.array([L.ref_pca_curvature(C.c_float(a), C.c_float(b), C.c_float(c)) for a, b, c in lam])


def ref_adaptive_tail(xyz, lam, cnt, curv, R_nms, ratio_max=0.65, min_n=20, upper=50000, lower=5000):
    """keypoint_detect.hpp:60-107 itself (the threshold loop of keypointDetectionBasedOnCurvature_adaptive) on given PCA features."""
    xyz = np.ascontiguousarray(_f32(xyz)[:, :3])
    lam, cnt, curv = _f32(lam), np.ascontiguousarray(cnt, np.int32), np.ascontiguousarray(curv, np.float64)
    out = np.empty(max(1, xyz.shape[0]), np.int32)
    k = ref3_lib().ref_adaptive_tail(_p(xyz, C.c_float), _p(lam, C.c_float), _p(cnt, C.c_int), _p(curv, C.c_double), xyz.shape[0], int(min_n),
                                     C.c_float(R_nms), C.c_float(ratio_max), int(upper), int(lower), _p(out, C.c_int))
    return out[:k].copy()


def ref_keyfpfh(hist, kp):
    """fpfh.hpp:93-115 itself: the histogram rows of the keypoints."""
    hist, kp = _f32(hist), np.ascontiguousarray(kp, np.int32)
    out = np.empty((kp.size, 33), np.float32)
    ref3_lib().ref_keyfpfh(_p(hist, C.c_float), hist.shape[0], _p(kp, C.c_int), int(kp.size), _p(out, C.c_float))
    return out


def ref_cal_overlap(c1, c2, thre_dis):
    """common_reg.cpp:302-313 itself (counting loop and ratio; exact stand-in radius search)."""
    L = ref3_lib()
    L.ref_cal_overlap.restype = C.c_float
    c1, c2 = np.ascontiguousarray(_f32(c1)[:, :3]), np.ascontiguousarray(_f32(c2)[:, :3])
    return float(L.ref_cal_overlap(_p(c1, C.c_float), c1.shape[0], _p(c2, C.c_float), c2.shape[0], C.c_float(thre_dis)))


def adaptive_tail(xyz, lam, cnt, curv, R_nms, ratio_max=0.65, min_n=20, upper=50000, lower=5000):
    """The restatement's threshold loop of keypointDetectionBasedOnCurvature_adaptive on given PCA features (orc_adaptive_tail)."""
    xyz = _f32(xyz)
    lam, cnt, curv = _f32(lam), np.ascontiguousarray(cnt, np.int32), np.ascontiguousarray(curv, np.float64)
    out = np.empty(max(1, xyz.shape[0]), np.int32)
    k = lib().orc_adaptive_tail(_p(xyz, C.c_float), xyz.shape[0], xyz.shape[1], _p(lam, C.c_float), _p(curv, C.c_double), _p(cnt, C.c_int),
                                C.c_float(ratio_max), int(min_n), C.c_float(R_nms), C.c_longlong(upper), C.c_longlong(lower), _p(out, C.c_int))
    return out[:k].copy()


def lcs_from_cov(cov):
    """The contract's LCS of a 3 x 3 weighted covariance: (axes (3, 3) f32 rows x, y, z; eigenvalues (3,) f32; eigenvectors (3, 3) f32 in
    columns, sign convention applied) -- what the stand-in for Eigen::EigenSolver 'returns' and what the reference's steps make of it."""
    cov = np.ascontiguousarray(cov, np.float32).reshape(9)
    axes, vals, vecs = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32)
    lib().orc_lcs_from_cov(_p(cov, C.c_float), _p(axes, C.c_float), _p(vals, C.c_float), _p(vecs, C.c_float))
    return axes.reshape(3, 3), vals, vecs.reshape(3, 3)


def ref_lcs(xyz, idx, test_index, R, values, vectors):
    """binary_feature_extraction.hpp:939-1035 + 119-155 themselves over a stand-in Eigen::EigenSolver that returns (values, vectors):
    (4, 3) f32 rows x axis, y axis, z axis, origin."""
    xyz = np.ascontiguousarray(_f32(xyz)[:, :3])
    idx = np.ascontiguousarray(idx, np.int32)
    values, vectors = np.ascontiguousarray(values, np.float32), np.ascontiguousarray(vectors, np.float32).reshape(9)
    out = np.zeros(12, np.float32)
    ok = ref3_lib().ref_lcs(_p(xyz, C.c_float), xyz.shape[0], _p(idx, C.c_int), int(idx.size), int(test_index), C.c_float(R), _p(values, C.c_float),
                            _p(vectors, C.c_float), _p(out, C.c_float))
    return out.reshape(4, 3) if ok else None
