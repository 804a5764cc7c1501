"""ctypes binding of libghicp_hip.so (include/ghicp_c.h) for tests and bench.py.

PyTorch is used only as the owner of device memory and streams; every compute call goes through
the C ABI into the HIP kernels.  There is NO CPU fallback: a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GHICP_LIB: another in-tree build of the same library (variants of a kernel timed side by side, scripts/km_variant_lib.sh)
LIB_PATH = os.environ.get("GHICP_LIB") or os.path.join(_HERE, "libghicp_hip.so")

FEATURE_BSC, FEATURE_ROPS, FEATURE_FPFH, FEATURE_NONE = 0, 1, 2, 3
CORR_NN, CORR_NNR, CORR_KM = 0, 1, 2


class Params(C.Structure):
    _fields_ = [("feature", C.c_int32), ("corr", C.c_int32), ("dof", C.c_int32), ("max_iter", C.c_int32),
                ("radius_nonmax", C.c_float), ("adjust_ratio", C.c_float), ("adjust_step", C.c_float),
                ("est_iou", C.c_float), ("converge_t", C.c_float), ("converge_r", C.c_float),
                ("bbx_magnitude", C.c_float), ("pad_", C.c_float),
                ("penalty_initial", C.c_double), ("para1", C.c_double), ("para2", C.c_double),
                ("km_eps", C.c_double), ("min_cor", C.c_int32), ("weight_changing_rate", C.c_int32)]


class Iter(C.Structure):
    _fields_ = [("cor", C.c_int32), ("converged", C.c_int32)] + [
        (k, C.c_double) for k in ("penalty", "cdmean", "cdstd", "rmse", "rmse_after", "fdm", "fdstd",
                                  "iou", "para1", "para2", "energy")] + [("Rt", C.c_double * 16)]


class PairConfig(C.Structure):
    _fields_ = [("reg", Params), ("voxel", C.c_float), ("neighborhood_radius", C.c_float),
                ("ratio_max", C.c_float), ("min_neighbors", C.c_int32), ("pattern", C.c_int32 * 98)]


class PairStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_s", "n_t", "m_s", "m_t", "k_s", "k_t")] + [
        ("iterations", C.c_int32), ("converged", C.c_int32), ("Rt", C.c_double * 16), ("bbx_magnitude", C.c_float)] + [
        (k, C.c_float) for k in ("ms_voxel", "ms_keypoints", "ms_feature", "ms_fd", "ms_loop", "ms_total", "pad_")] + [
        ("rmse_after", C.c_double), ("registered_ok", C.c_int32), ("pad2_", C.c_int32)]


EXPORTS = [
    "ghicp_ctx_create", "ghicp_ctx_destroy", "ghicp_ctx_set_stream", "ghicp_ctx_set_host_pointers", "ghicp_ctx_stage_stats", "ghicp_ctx_stage_clear", "ghicp_ctx_set_cu_mask",
    "ghicp_ctx_synchronize", "ghicp_ctx_kernel_timing", "ghicp_ctx_kernel_time", "ghicp_ctx_km_launch_stats", "ghicp_ctx_pair_loop_stats", "ghicp_ctx_loop_timeline", "ghicp_ctx_set_loop_cost_hints", "ghicp_ctx_loop_progress", "ghicp_ctx_loop_progress_reset", "ghicp_ctx_loop_hazards", "ghicp_last_error", "ghicp_version", "ghicp_params_default",
    "ghicp_voxel_filter", "ghicp_sort_pairs", "ghicp_gather_points", "ghicp_bbx_magnitude", "ghicp_cloud_bounds", "ghicp_pca_curvature", "ghicp_prune",
    "ghicp_nms", "ghicp_keypoints", "ghicp_keypoints_adaptive", "ghicp_bsc_encode", "ghicp_fpfh", "ghicp_fpfh_keypoints", "ghicp_fd_bsc", "ghicp_fd_fpfh", "ghicp_km_solve",
    "ghicp_rigid_svd", "ghicp_rigid_svd_host", "ghicp_register", "ghicp_loop_create", "ghicp_iterate", "ghicp_loop_result", "ghicp_loop_destroy", "ghicp_transform_cloud", "ghicp_transform_clouds", "ghicp_register_pair",
    "ghicp_register_pairs",
    "ghicp_icp_params_default", "ghicp_cal_overlap", "ghicp_icp", "ghicp_knn_normals", "ghicp_nn_search", "ghicp_inv_transform",
    "ghicp_transform_cloud_f32",
    "ghicp_cloud_create", "ghicp_cloud_recompute", "ghicp_clouds_recompute", "ghicp_cloud_from_features", "ghicp_cloud_destroy", "ghicp_cloud_get_info", "ghicp_cloud_download",
    "ghicp_register_clouds", "ghicp_sbf_write", "ghicp_sbf_read",
    "ghicp_pairqueue_create", "ghicp_pairqueue_destroy", "ghicp_pairqueue_last_error", "ghicp_pairqueue_info", "ghicp_pairqueue_broadcast",
    "ghicp_pairqueue_barrier", "ghicp_pairqueue_static_share", "ghicp_pairqueue_claim", "ghicp_pairqueue_counter_reset",
    "ghicp_pairqueue_gather_records", "ghicp_pairqueue_pack_records", "ghicp_pairqueue_register_pairs",
]

_lib = None


class GhicpError(RuntimeError):
    pass


def load():
    """Load the HIP library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GhicpError("libghicp_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(or make -C gh-icp_amd/csrc). There is no CPU fallback.")
        # torch BEFORE the library: libtorch_hip links the unversioned name "libamdhip64.so", which does not match the soname (libamdhip64.so.7) of a
        # runtime that is already loaded, so with the library first the process ends up with TWO HIP runtimes and ghicp_ctx_create reports no GPU
        # (build() followed by smoke() in one process did, round 6 call 25); with torch first the library's libamdhip64.so.7 resolves to torch's copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.ghicp_last_error.restype = C.c_char_p
        _lib.ghicp_version.restype = C.c_char_p
        _lib.ghicp_loop_destroy.restype = None
        _lib.ghicp_pairqueue_last_error.restype = C.c_char_p
    return _lib


def default_params(feature=FEATURE_BSC, corr=CORR_KM, dof=6, est_iou=0.6, radius_nonmax=1.5, bbx_magnitude=0.0,
                   max_iter=200) -> Params:
    p = Params()
    load().ghicp_params_default(C.byref(p))
    p.feature, p.corr, p.dof, p.est_iou, p.radius_nonmax = feature, corr, dof, est_iou, radius_nonmax
    p.bbx_magnitude, p.max_iter = bbx_magnitude, max_iter
    return p


ICP_POINT_TO_POINT, ICP_POINT_TO_PLANE = 0, 1


class IcpParams(C.Structure):
    """ghicp_icp_params: the arguments of CRegistration::icp_reg / ptplicp_reg (common_reg.cpp:45-55, 122-133)."""
    _fields_ = [("max_iter", C.c_int32), ("use_reciprocal", C.c_int32), ("use_trimmed", C.c_int32), ("metric", C.c_int32),
                ("thre_dis", C.c_float), ("min_overlap", C.c_float), ("covariance_k", C.c_int32), ("pad_", C.c_int32),
                ("transformation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double)]


class IcpStats(C.Structure):
    _fields_ = [("done", C.c_int32), ("iterations", C.c_int32), ("converged", C.c_int32), ("reason", C.c_int32),
                ("correspondences", C.c_int64), ("overlap", C.c_float), ("pad_", C.c_float),
                ("mse", C.c_double), ("fitness", C.c_double)]


def icp_params(max_iter=50, reciprocal=False, trimmed=False, metric=ICP_POINT_TO_POINT, thre_dis=0.5, min_overlap=0.1,
               covariance_k=15) -> IcpParams:
    p = IcpParams()
    load().ghicp_icp_params_default(C.byref(p))
    p.max_iter, p.use_reciprocal, p.use_trimmed, p.metric = max_iter, int(reciprocal), int(trimmed), metric
    p.thre_dis, p.min_overlap, p.covariance_k = thre_dis, min_overlap, covariance_k
    return p


def rigid_svd_host(src, tgt):
    """ghicp_rigid_svd_host: float Umeyama on the host with the kernels' numerics contract (no GPU needed)."""
    src = np.ascontiguousarray(src, np.float64)
    tgt = np.ascontiguousarray(tgt, np.float64)
    out = np.zeros(16)
    rc = load().ghicp_rigid_svd_host(src.ctypes.data_as(C.POINTER(C.c_double)), tgt.ctypes.data_as(C.POINTER(C.c_double)),
                                     C.c_int64(src.shape[0]), out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise GhicpError("ghicp_rigid_svd_host failed with code %d" % rc)
    return out.reshape(4, 4)


def inv_transform(T):
    """CRegistration::invTransform (common_reg.cpp:357-370), host-side."""
    T = np.ascontiguousarray(T, np.float32)
    out = np.zeros(16, np.float32)
    load().ghicp_inv_transform(T.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(4, 4)


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


class Context:
    """One ghicp_ctx bound to a device (and optionally a torch stream)."""

    def __init__(self, device: int = 0, stream=None):
        import torch

        if not torch.cuda.is_available():
            raise GhicpError("no GPU visible: the GH-ICP hot path has no CPU fallback")
        self.torch = torch
        self.lib = load()
        self.device = device
        h = C.c_void_p()
        rc = self.lib.ghicp_ctx_create(device, C.byref(h))
        if rc != 0:
            raise GhicpError("ghicp_ctx_create failed with code %d" % rc)
        self.h = h
        if stream is not None:
            self.set_stream(stream)
        self.dev = torch.device("cuda", device)

    def set_stream(self, stream):
        self._check(self.lib.ghicp_ctx_set_stream(self.h, C.c_void_p(stream.cuda_stream)))

    def set_cu_mask(self, mask_words):
        """Own stream restricted to the CUs whose bit is set (list of uint32 words, 32 CUs each)."""
        m = (C.c_uint32 * len(mask_words))(*[int(w) & 0xFFFFFFFF for w in mask_words])
        self._check(self.lib.ghicp_ctx_set_cu_mask(self.h, m, len(mask_words)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.ghicp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise GhicpError("ghicp error %d: %s" % (rc, self.lib.ghicp_last_error(self.h).decode()))

    def sync(self):
        self._check(self.lib.ghicp_ctx_synchronize(self.h))

    def kernel_timing(self, on=True):
        self._check(self.lib.ghicp_ctx_kernel_timing(self.h, 1 if on else 0))

    def km_launch_stats(self):
        """dict of the Kuhn-Munkres launch records collected since kernel_timing(True) (ghicp_ctx_km_launch_stats)."""
        out = (C.c_double * 8)()
        self._check(self.lib.ghicp_ctx_km_launch_stats(self.h, out))
        keys = ("launches", "solves", "mean_solve_ms", "mean_longest_solve_ms", "mean_launch_span_ms", "slots", "idle_slot_fraction", "worst_longest_over_mean")
        return dict(zip(keys, (float(v) for v in out)))

    def pair_loop_stats(self):
        """dict of the persistent pair loop's launch records collected since kernel_timing(True) (ghicp_ctx_pair_loop_stats)."""
        out = (C.c_double * 8)()
        self._check(self.lib.ghicp_ctx_pair_loop_stats(self.h, out))
        keys = ("launches", "slots", "solves", "mean_solve_ms", "longest_solve_ms", "mean_launch_span_ms", "idle_slot_fraction", "solve_share_of_slot_time")
        return dict(zip(keys, (float(v) for v in out)))

    def loop_progress(self):
        """(pairs still iterating, pairs of the batch) of the batched loop running on this context; callable from another thread."""
        a, t = C.c_int64(0), C.c_int64(0)
        self.lib.ghicp_ctx_loop_progress(self.h, C.byref(a), C.byref(t))
        return a.value, t.value

    def loop_progress_reset(self, total):
        """declare a batch of `total` pairs as about to start (so that other threads never read the previous batch's finished state)"""
        self._check(self.lib.ghicp_ctx_loop_progress_reset(self.h, C.c_int64(int(total))))

    def set_loop_cost_hints(self, cost):
        """queue order of the next batched Kuhn-Munkres registration with len(cost) pairs: costliest first (results do not depend on it)"""
        c = np.ascontiguousarray(cost, dtype=np.float32)
        self._check(self.lib.ghicp_ctx_set_loop_cost_hints(self.h, C.c_int32(c.size), c.ctypes.data_as(C.POINTER(C.c_float))))

    def loop_timeline(self, detail=False):
        """(n, 3) int64: per pair of the last persistent batch (kernel timing on) slot begin / end in 100 MHz ticks, iterations.
        detail=True: (n, 7) -- + the pair's longest solve (ticks), its iteration, the compute unit (die * 64 + engine * 8 + array * ... packed as
        die, engine, array, cu in one number: die << 8 | engine << 5 | array << 4 | cu) the slot ran on, 0 (ghicp_c.h: ghicp_ctx_loop_timeline)."""
        n = C.c_int64(0)
        self._check(self.lib.ghicp_ctx_loop_timeline(self.h, None, C.c_int64(0), C.byref(n)))
        raw = np.zeros((max(1, n.value), 3), np.uint64)
        self._check(self.lib.ghicp_ctx_loop_timeline(self.h, raw.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int64(n.value), C.byref(n)))
        raw = raw[: n.value]
        mask = np.uint64(0x000FFFFFFFFFFFFF)
        out = np.zeros((raw.shape[0], 7 if detail else 3), np.int64)
        out[:, 0] = (raw[:, 0] & mask).astype(np.int64)
        out[:, 1] = (raw[:, 1] & mask).astype(np.int64)
        out[:, 2] = (raw[:, 2] & np.uint64(0xFFFF)).astype(np.int64)
        if detail:
            out[:, 3] = ((raw[:, 2] >> np.uint64(32)) << np.uint64(4)).astype(np.int64)
            out[:, 4] = ((raw[:, 2] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
            out[:, 5] = (raw[:, 0] >> np.uint64(52)).astype(np.int64)
        return out

    def loop_hazards(self):
        """Kuhn-Munkres solves of this context's batched loops that took the solver's literal fallback (diagnostics; expected 0)."""
        n = C.c_int64(0)
        self._check(self.lib.ghicp_ctx_loop_hazards(self.h, C.byref(n)))
        return n.value

    def kernel_time(self, name):
        """(total_ms, launches) of a named kernel since kernel_timing(True)."""
        ms, n = C.c_double(0), C.c_int64(0)
        self._check(self.lib.ghicp_ctx_kernel_time(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def _dev(self, a, dtype):
        t = self.torch
        if isinstance(a, np.ndarray):
            a = t.from_numpy(np.ascontiguousarray(a))
        return a.to(device=self.dev, dtype=dtype).contiguous()

    # ---------------------------------------------------------------- per pair
    def fd_bsc(self, featS, featT):
        """featS (V,ks,56) u8, featT (kt,56) u8 -> FD (ks,kt) int16 tensor (values are u16 <= 441)."""
        t = self.torch
        fS, fT = self._dev(featS, t.uint8), self._dev(featT, t.uint8)
        V, ks, _ = fS.shape
        kt = fT.shape[0]
        FD = t.empty((ks, kt), dtype=t.int16, device=self.dev)
        self._check(self.lib.ghicp_fd_bsc(self.h, _ptr(fS), C.c_int64(ks), V, _ptr(fT), C.c_int64(kt), _ptr(FD)))
        return FD

    def fd_fpfh(self, hS, hT):
        t = self.torch
        hS, hT = self._dev(hS, t.float32), self._dev(hT, t.float32)
        FD = t.empty((hS.shape[0], hT.shape[0]), dtype=t.float32, device=self.dev)
        self._check(self.lib.ghicp_fd_fpfh(self.h, _ptr(hS), C.c_int64(hS.shape[0]), _ptr(hT), C.c_int64(hT.shape[0]), _ptr(FD)))
        return FD

    def km_solve(self, w, eps=0.01):
        t = self.torch
        w = self._dev(w, t.float64)
        n = w.shape[0]
        match = t.empty(n, dtype=t.int32, device=self.dev)
        self._check(self.lib.ghicp_km_solve(self.h, _ptr(w), C.c_int64(n), C.c_double(eps), _ptr(match)))
        return match

    def rigid_svd(self, src, tgt):
        t = self.torch
        src, tgt = self._dev(src, t.float64), self._dev(tgt, t.float64)
        Rt = (C.c_double * 16)()
        self._check(self.lib.ghicp_rigid_svd(self.h, _ptr(src), _ptr(tgt), C.c_int64(src.shape[0]), Rt))
        return np.array(Rt[:]).reshape(4, 4)

    def register(self, params: Params, kpS, kpT, FD=None, want_matchlist=False):
        """GHRegistration::ghicp_reg. kpS/kpT (k,3) f64; FD int16(u16)/f32 (ks,kt) or None."""
        t = self.torch
        kpS, kpT = self._dev(kpS, t.float64), self._dev(kpT, t.float64)
        ks, kt = kpS.shape[0], kpT.shape[0]
        if FD is not None:
            assert tuple(FD.shape) == (ks, kt) and FD.is_contiguous()
        Rt = (C.c_double * 16)()
        trace = (Iter * params.max_iter)()
        n_iter = C.c_int32(0)
        ml = t.full((params.max_iter, ks), -2, dtype=t.int32, device=self.dev) if want_matchlist else None
        self._check(self.lib.ghicp_register(self.h, C.byref(params), _ptr(kpS), C.c_int64(ks), _ptr(kpT), C.c_int64(kt), _ptr(FD), Rt,
                                            trace, C.byref(n_iter), _ptr(ml)))
        it = n_iter.value
        tr = []
        for i in range(it):
            r = trace[i]
            d = {k: getattr(r, k) for k, _ in Iter._fields_ if k != "Rt"}
            d["Rt"] = np.array(r.Rt[:]).reshape(4, 4)
            tr.append(d)
        return dict(Rt=np.array(Rt[:]).reshape(4, 4), iters=it, trace=tr,
                    matchlist=None if ml is None else ml[:it].cpu().numpy())

    def register_stepwise(self, params: Params, kpS, kpT, FD=None, max_steps=None, extra_calls=0):
        """The same registration through ghicp_loop_create + ghicp_iterate (one iteration of ghicp_reg.cpp:49-103 per call) +
        ghicp_loop_result.  Same return value as register(..., want_matchlist=True)."""
        t = self.torch
        kpS, kpT = self._dev(kpS, t.float64), self._dev(kpT, t.float64)
        ks, kt = kpS.shape[0], kpT.shape[0]
        if FD is not None:
            assert tuple(FD.shape) == (ks, kt) and FD.is_contiguous()
        loop = C.c_void_p()
        self._check(self.lib.ghicp_loop_create(self.h, C.byref(params), _ptr(kpS), C.c_int64(ks), _ptr(kpT), C.c_int64(kt), _ptr(FD), C.byref(loop)))
        try:
            tr, rows = [], []
            for _ in range(max_steps or params.max_iter):
                rec = Iter()
                row = t.full((max(ks, 1),), -2, dtype=t.int32, device=self.dev)
                self._check(self.lib.ghicp_iterate(self.h, loop, C.byref(rec), _ptr(row)))
                d = {k: getattr(rec, k) for k, _ in Iter._fields_ if k != "Rt"}
                d["Rt"] = np.array(rec.Rt[:]).reshape(4, 4)
                tr.append(d)
                rows.append(row[:ks].cpu().numpy())
                if rec.converged:
                    break
            for _ in range(extra_calls):  # (tests: iterating a converged loop is an argument error)
                self._check(self.lib.ghicp_iterate(self.h, loop, C.byref(Iter()), None))
            Rt = (C.c_double * 16)()
            n_iter, conv, ra = C.c_int32(0), C.c_int32(0), C.c_double(0)
            self._check(self.lib.ghicp_loop_result(loop, Rt, C.byref(n_iter), C.byref(conv), C.byref(ra)))
        finally:
            self.lib.ghicp_loop_destroy(loop)
        return dict(Rt=np.array(Rt[:]).reshape(4, 4), iters=n_iter.value, trace=tr, converged=conv.value, rmse_after=ra.value,
                    matchlist=np.stack(rows) if rows else np.zeros((0, ks), np.int32))

    # ---------------------------------------------------------------- front end
    def _xyz(self, xyz):
        t = self.torch
        x = self._dev(xyz, t.float32)
        assert x.dim() == 2 and x.shape[1] >= 3
        return x

    def voxel_filter(self, xyz, voxel):
        t = self.torch
        x = self._xyz(xyz)
        n = x.shape[0]
        keep = t.empty(n + 1, dtype=t.int32, device=self.dev)
        m = C.c_int64(0)
        self._check(self.lib.ghicp_voxel_filter(self.h, _ptr(x), C.c_int64(n), x.shape[1], C.c_float(voxel), _ptr(keep), C.byref(m)))
        return keep[: m.value]

    def sort_pairs(self, keys, vals=None, bit_begin=0, bit_end=None):
        """stable ascending sort on the key bits [bit_begin, bit_end) (prims.hip); keys: int32 / int64 tensor or array holding u32 / u64 bit patterns"""
        t = self.torch
        k = self._dev(keys, None)
        assert k.dtype in (t.int32, t.int64) and k.dim() == 1
        kb = 4 if k.dtype == t.int32 else 8
        n = k.shape[0]
        ko = t.empty_like(k)
        v = vo = None
        if vals is not None:
            v = self._dev(vals, t.int32)
            vo = t.empty_like(v)
        self._check(self.lib.ghicp_sort_pairs(self.h, kb, _ptr(k), _ptr(ko), _ptr(v) if v is not None else None, _ptr(vo) if vo is not None else None,
                                              C.c_int64(n), int(bit_begin), int(8 * kb if bit_end is None else bit_end)))
        return (ko, vo) if vals is not None else ko

    def bbx_magnitude(self, xyz):
        x = self._xyz(xyz)
        out = C.c_float(0)
        self._check(self.lib.ghicp_bbx_magnitude(self.h, _ptr(x), C.c_int64(x.shape[0]), x.shape[1], C.byref(out)))
        return out.value

    def cloud_bounds(self, xyz):
        """CloudUtility::getCloudBound (utility.h:153-183): (min_x, min_y, min_z, max_x, max_y, max_z)"""
        x = self._xyz(xyz)
        out = (C.c_double * 6)()
        self._check(self.lib.ghicp_cloud_bounds(self.h, _ptr(x), C.c_int64(x.shape[0]), x.shape[1], out))
        return np.array(out[:], np.float64)

    def pca_curvature(self, xyz, radius):
        t = self.torch
        x = self._xyz(xyz)
        m = x.shape[0]
        lam = t.empty((m, 3), dtype=t.float32, device=self.dev)
        curv = t.empty(m, dtype=t.float64, device=self.dev)
        cnt = t.empty(m, dtype=t.int32, device=self.dev)
        self._check(self.lib.ghicp_pca_curvature(self.h, _ptr(x), C.c_int64(m), x.shape[1], C.c_float(radius), _ptr(lam), _ptr(curv), _ptr(cnt)))
        return lam, curv, cnt

    def prune(self, lam, cnt, ratio_max=0.65, min_n=20):
        t = self.torch
        lam, cnt = self._dev(lam, t.float32), self._dev(cnt, t.int32)
        m = lam.shape[0]
        cand = t.empty(max(m, 1), dtype=t.int32, device=self.dev)
        c = C.c_int64(0)
        self._check(self.lib.ghicp_prune(self.h, _ptr(lam), _ptr(cnt), C.c_int64(m), C.c_float(ratio_max), min_n, _ptr(cand), C.byref(c)))
        return cand[: c.value]

    def nms(self, xyz, curv, cand, radius):
        t = self.torch
        x = self._xyz(xyz)
        curv, cand = self._dev(curv, t.float64), self._dev(cand, t.int32)
        c = cand.shape[0]
        kp = t.empty(max(c, 1), dtype=t.int32, device=self.dev)
        k = C.c_int64(0)
        self._check(self.lib.ghicp_nms(self.h, _ptr(x), x.shape[1], _ptr(curv), _ptr(cand), C.c_int64(c), C.c_float(radius), _ptr(kp), C.byref(k)))
        return kp[: k.value]

    def keypoints(self, xyz, radius, nms_radius, ratio_max=0.65, min_n=20):
        t = self.torch
        x = self._xyz(xyz)
        m = x.shape[0]
        kp = t.empty(max(m, 1), dtype=t.int32, device=self.dev)
        k = C.c_int64(0)
        self._check(self.lib.ghicp_keypoints(self.h, _ptr(x), C.c_int64(m), x.shape[1], C.c_float(radius), C.c_float(ratio_max), min_n,
                                             C.c_float(nms_radius), _ptr(kp), C.byref(k)))
        return kp[: k.value]

    def keypoints_adaptive(self, xyz, radius, nms_radius, ratio_max=0.65, min_n=20, upper=50000, lower=5000):
        """keypointDetectionBasedOnCurvature_adaptive: returns (kp tensor, ratio_used, rounds)."""
        t = self.torch
        x = self._xyz(xyz)
        m = x.shape[0]
        kp = t.empty(max(m, 1), dtype=t.int32, device=self.dev)
        k, ru, nr = C.c_int64(0), C.c_float(0), C.c_int32(0)
        self._check(self.lib.ghicp_keypoints_adaptive(self.h, _ptr(x), C.c_int64(m), x.shape[1], C.c_float(radius), C.c_float(ratio_max), min_n,
                                                      C.c_float(nms_radius), C.c_int64(upper), C.c_int64(lower), _ptr(kp), C.byref(k),
                                                      C.byref(ru), C.byref(nr)))
        return kp[: k.value], ru.value, nr.value

    def bsc_encode(self, xyz, kp, radius, dof, pattern):
        t = self.torch
        x = self._xyz(xyz)
        kp = self._dev(kp, t.int32)
        K = kp.shape[0]
        pat = np.ascontiguousarray(pattern, dtype=np.int32).reshape(-1)
        assert pat.size == 98
        feat = t.zeros((4, K, 56), dtype=t.uint8, device=self.dev)
        lcs = t.zeros((K, 12), dtype=t.float32, device=self.dev)
        self._check(self.lib.ghicp_bsc_encode(self.h, _ptr(x), C.c_int64(x.shape[0]), x.shape[1], _ptr(kp), C.c_int64(K), C.c_float(radius), dof,
                                              pat.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(feat), _ptr(lcs)))
        return feat, lcs

    def fpfh(self, xyz):
        """pcl NormalEstimation(k=20) + FPFHEstimation(k=20): returns (normals (m,3), hist (m,33)) device tensors."""
        t = self.torch
        x = self._xyz(xyz)
        m = x.shape[0]
        nrm = t.empty((m, 3), dtype=t.float32, device=self.dev)
        hist = t.empty((m, 33), dtype=t.float32, device=self.dev)
        self._check(self.lib.ghicp_fpfh(self.h, _ptr(x), C.c_int64(m), x.shape[1], 20, 20, _ptr(nrm), _ptr(hist)))
        return nrm, hist

    def transform_cloud(self, xyz, Rt):
        t = self.torch
        x = self._xyz(xyz)
        Rt = np.ascontiguousarray(Rt, dtype=np.float64)
        out = t.empty((x.shape[0], 3), dtype=t.float32, device=self.dev)
        self._check(self.lib.ghicp_transform_cloud(self.h, _ptr(x), C.c_int64(x.shape[0]), x.shape[1], Rt.ctypes.data_as(C.POINTER(C.c_double)), _ptr(out)))
        return out

    def transform_clouds(self, clouds, Rts, outs):
        """S7 (main:153) of a batch in one launch: device clouds (n_i, stride) under Rts[i] (4x4) into the device tensors outs[i] (n_i, 3) f32."""
        n = len(clouds)
        if n == 0:
            return
        stride = clouds[0].shape[1]
        assert all(c.shape[1] == stride and c.is_contiguous() and c.dtype == self.torch.float32 for c in clouds)
        assert all(o.is_contiguous() and o.dtype == self.torch.float32 and o.shape == (c.shape[0], 3) for c, o in zip(clouds, outs))
        xs = (C.c_void_p * n)(*[c.data_ptr() for c in clouds])
        os_ = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        ns = (C.c_int64 * n)(*[c.shape[0] for c in clouds])
        R = np.ascontiguousarray(np.asarray(Rts, dtype=np.float64).reshape(n, 16))
        self._check(self.lib.ghicp_transform_clouds(self.h, C.c_int32(n), xs, ns, C.c_int(stride), R.ctypes.data_as(C.POINTER(C.c_double)), os_))

    # ---------------------------------------------------------------- fine registration (common_reg)
    def cal_overlap(self, c1, c2, thre_dis):
        a, b = self._xyz(c1), self._xyz(c2)
        r = C.c_float(0)
        self._check(self.lib.ghicp_cal_overlap(self.h, _ptr(a), C.c_int64(a.shape[0]), a.shape[1], _ptr(b), C.c_int64(b.shape[0]), b.shape[1],
                                               C.c_float(thre_dis), C.byref(r)))
        return r.value

    def knn_normals(self, xyz, k):
        t = self.torch
        x = self._xyz(xyz)
        out = t.empty((x.shape[0], 3), dtype=t.float32, device=self.dev)
        self._check(self.lib.ghicp_knn_normals(self.h, _ptr(x), C.c_int64(x.shape[0]), x.shape[1], int(k), _ptr(out)))
        return out

    def nn_search(self, query, target):
        t = self.torch
        q, x = self._xyz(query), self._xyz(target)
        idx = t.empty((q.shape[0],), dtype=t.int32, device=self.dev)
        d2 = t.empty((q.shape[0],), dtype=t.float32, device=self.dev)
        self._check(self.lib.ghicp_nn_search(self.h, _ptr(q), C.c_int64(q.shape[0]), q.shape[1], _ptr(x), C.c_int64(x.shape[0]), x.shape[1],
                                             _ptr(idx), _ptr(d2)))
        return idx, d2

    def transform_cloud_f32(self, xyz, T):
        t = self.torch
        x = self._xyz(xyz)
        T = np.ascontiguousarray(T, dtype=np.float32)
        out = t.empty((x.shape[0], 3), dtype=t.float32, device=self.dev)
        self._check(self.lib.ghicp_transform_cloud_f32(self.h, _ptr(x), C.c_int64(x.shape[0]), x.shape[1], T.ctypes.data_as(C.POINTER(C.c_float)),
                                                       _ptr(out)))
        return out

    def icp(self, xyzS, xyzT, params: IcpParams, want_transformed=True):
        """icp_reg / ptplicp_reg.  Returns dict(done, T (4,4) f32, transformed tensor, + ghicp_icp_stats fields)."""
        t = self.torch
        xS, xT = self._xyz(xyzS), self._xyz(xyzT)
        T = np.zeros(16, np.float32)
        out = t.empty((xS.shape[0], 3), dtype=t.float32, device=self.dev) if want_transformed else None
        st = IcpStats()
        self._check(self.lib.ghicp_icp(self.h, _ptr(xS), C.c_int64(xS.shape[0]), xS.shape[1], _ptr(xT), C.c_int64(xT.shape[0]), xT.shape[1],
                                       C.byref(params), T.ctypes.data_as(C.POINTER(C.c_float)), _ptr(out), C.byref(st)))
        d = {k: getattr(st, k) for k, _ in IcpStats._fields_ if k != "pad_"}
        d.update(T=T.reshape(4, 4), transformed=out)
        return d

    def register_pair(self, cfg: PairConfig, xyzS, xyzT, want_trace=True):
        xS, xT = self._xyz(xyzS), self._xyz(xyzT)
        assert xS.shape[1] == xT.shape[1]
        stats = PairStats()
        trace = (Iter * cfg.reg.max_iter)() if want_trace else None
        self._check(self.lib.ghicp_register_pair(self.h, C.byref(cfg), _ptr(xS), C.c_int64(xS.shape[0]), _ptr(xT), C.c_int64(xT.shape[0]),
                                                 xS.shape[1], C.byref(stats), trace))
        tr = []
        if want_trace:
            for i in range(stats.iterations):
                r = trace[i]
                d = {k: getattr(r, k) for k, _ in Iter._fields_ if k != "Rt"}
                d["Rt"] = np.array(r.Rt[:]).reshape(4, 4)
                tr.append(d)
        return stats, tr


def _register_pairs(self, cfg, pairs):
    """pairs: list of (xyzS, xyzT) device tensors (float32, same column count). Returns list[PairStats]."""
    n = len(pairs)
    if n == 0:
        return []
    xs = [self._xyz(a) for a, _ in pairs]
    xt = [self._xyz(b) for _, b in pairs]
    stride = xs[0].shape[1]
    assert all(t.shape[1] == stride for t in xs + xt)
    PS = (C.c_void_p * n)(*[t.data_ptr() for t in xs])
    PT = (C.c_void_p * n)(*[t.data_ptr() for t in xt])
    NS = (C.c_int64 * n)(*[t.shape[0] for t in xs])
    NT = (C.c_int64 * n)(*[t.shape[0] for t in xt])
    stats = (PairStats * n)()
    self._check(self.lib.ghicp_register_pairs(self.h, C.byref(cfg), n, PS, NS, PT, NT, stride, stats))
    return list(stats)


Context.register_pairs = _register_pairs


# ---------------------------------------------------------------- per-cloud front-end cache
class CloudInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("k", C.c_int64), ("variants", C.c_int32), ("feature", C.c_int32),
                ("bbx_magnitude", C.c_float), ("candidates", C.c_int32), ("feature_bytes", C.c_int64)]


class Cloud:
    """A ghicp_cloud handle: the down-sampled points, keypoints and descriptors of one cloud, resident in HBM."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def info(self) -> CloudInfo:
        i = CloudInfo()
        self.ctx._check(self.ctx.lib.ghicp_cloud_get_info(self.h, C.byref(i)))
        return i

    def download(self, points=True):
        """Returns dict(ds (m,3) f32 | None, kp (k,) i32 | None, kp_xyz (k,3) f64, feat: (V,k,56) u8 | (k,33) f32 | None) as device tensors."""
        t, c, i = self.ctx.torch, self.ctx, self.info()
        ds = t.empty((i.m, 3), dtype=t.float32, device=c.dev) if points and i.m > 0 else None
        kp = t.empty((i.k,), dtype=t.int32, device=c.dev) if points and i.m > 0 else None
        kpx = t.empty((i.k, 3), dtype=t.float64, device=c.dev)
        feat = None
        if i.feature == FEATURE_BSC:
            feat = t.empty((i.variants, i.k, 56), dtype=t.uint8, device=c.dev)
        elif i.feature == FEATURE_FPFH:
            feat = t.empty((i.k, 33), dtype=t.float32, device=c.dev)
        c._check(c.lib.ghicp_cloud_download(self.h, _ptr(ds), _ptr(kp), _ptr(kpx), _ptr(feat)))
        return dict(ds=ds, kp=kp, kp_xyz=kpx, feat=feat)

    def recompute(self, xyz):
        """Front end of another raw cloud into this handle (buffers reused)."""
        x = self.ctx._xyz(xyz)
        self.ctx._check(self.ctx.lib.ghicp_cloud_recompute(self.h, _ptr(x), C.c_int64(x.shape[0]), x.shape[1]))
        return self

    def close(self):
        if self.h:
            self.ctx.lib.ghicp_cloud_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _cloud_create(self, cfg, xyz) -> Cloud:
    x = self._xyz(xyz)
    h = C.c_void_p()
    self._check(self.lib.ghicp_cloud_create(self.h, C.byref(cfg), _ptr(x), C.c_int64(x.shape[0]), x.shape[1], C.byref(h)))
    return Cloud(self, h)


def _cloud_from_features(self, cfg, kp_xyz, feat, bbx_magnitude) -> Cloud:
    t = self.torch
    kpx = self._dev(kp_xyz, t.float64)
    if feat is not None:
        feat = self._dev(feat, t.uint8 if cfg.reg.feature == FEATURE_BSC else t.float32)
    h = C.c_void_p()
    self._check(self.lib.ghicp_cloud_from_features(self.h, C.byref(cfg), _ptr(kpx), C.c_int64(kpx.shape[0]), _ptr(feat), C.c_float(bbx_magnitude),
                                                   C.byref(h)))
    return Cloud(self, h)


def _register_clouds(self, cfg, pairs):
    """pairs: list of (Cloud S, Cloud T).  Returns list[PairStats]."""
    n = len(pairs)
    if n == 0:
        return []
    HS = (C.c_void_p * n)(*[a.h.value for a, _ in pairs])
    HT = (C.c_void_p * n)(*[b.h.value for _, b in pairs])
    stats = (PairStats * n)()
    self._check(self.lib.ghicp_register_clouds(self.h, C.byref(cfg), n, HS, HT, stats))
    return list(stats)


def _clouds_recompute(self, clouds, xyzs):
    """ghicp_clouds_recompute: the front ends of several raw clouds into existing handles (same configuration, this context) with one
    launch sequence for the whole batch; same results as Cloud.recompute() one by one."""
    n = len(clouds)
    if n == 0:
        return clouds
    xs = [self._xyz(x) for x in xyzs]
    assert len(xs) == n and all(x.shape[1] == xs[0].shape[1] for x in xs)
    H = (C.c_void_p * n)(*[c.h.value for c in clouds])
    P = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    N = (C.c_int64 * n)(*[x.shape[0] for x in xs])
    self._check(self.lib.ghicp_clouds_recompute(self.h, n, H, P, N, xs[0].shape[1]))
    return clouds


Context.cloud_create = _cloud_create
Context.clouds_recompute = _clouds_recompute
Context.cloud_from_features = _cloud_from_features
Context.register_clouds = _register_clouds


def sbf_write(path, feat):
    """StereoBinaryFeature::writeFeatures format (stereo_binary_feature.cpp:107-124); feat (k,56) u8 host array."""
    f = np.ascontiguousarray(feat, np.uint8).reshape(-1, 56)
    rc = load().ghicp_sbf_write(str(path).encode(), f.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(f.shape[0]))
    if rc != 0:
        raise GhicpError("ghicp_sbf_write(%s) failed" % path)


def sbf_read(path):
    k = C.c_int64(0)
    if load().ghicp_sbf_read(str(path).encode(), None, C.c_int64(0), C.byref(k)) != 0:
        raise GhicpError("ghicp_sbf_read(%s): not a 441-bit feature dump" % path)
    f = np.zeros((k.value, 56), np.uint8)
    if load().ghicp_sbf_read(str(path).encode(), f.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(k.value), C.byref(k)) != 0:
        raise GhicpError("ghicp_sbf_read(%s) failed" % path)
    return f


def pair_config(feature=FEATURE_BSC, corr=CORR_KM, dof=6, est_iou=0.6, voxel=0.1, neighborhood_radius=0.5, radius_nonmax=1.5,
                pattern=None, max_iter=200) -> PairConfig:
    cfg = PairConfig()
    cfg.reg = default_params(feature, corr, dof, est_iou, radius_nonmax, 0.0, max_iter)
    cfg.voxel, cfg.neighborhood_radius, cfg.ratio_max, cfg.min_neighbors = voxel, neighborhood_radius, 0.65, 20
    pat = np.zeros(98, np.int32) if pattern is None else np.ascontiguousarray(pattern, np.int32).reshape(-1)
    for i in range(98):
        cfg.pattern[i] = int(pat[i])
    return cfg
