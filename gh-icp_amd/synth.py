"""Deterministic synthetic inputs for the GH-ICP hot path.

Everything here is a pure function of an integer seed; the RNG is SplitMix64 used
counter-style (value i of stream `seed` = mix(seed + (i+1)*GAMMA)), so the same
arrays can be regenerated anywhere (numpy only, no GPU, no files).

Configs (BASELINE.json `configs`).  cfg1, cfg2 and cfg5 have the geometry SURVEY.md §8d fixes.  cfg3 and cfg4 exist in TWO variants since
round 5 (round-5 advisor: the change must be explicit, BASELINE.md §4 reports both):
  cfg1  gauss_pair        two 50k-pt Gaussian blobs, explicit keypoints, N/N, 6-DoF
  cfg2  tls_pair(1M)      ray-cast TLS scene 120x120 m, stations 13 m / 30 deg apart
  cfg3  tls_pair(5M)      same scene, finer angular grid.  variant "registering" (default; bench.py --config 3): station B = (5,-2.5,0), yaw -12 --
                          inside the basin of the reference's FPFH + reciprocal-NN loop.  variant "surveyed" (bench.py --config 13): SURVEY.md
                          §8d's station B = (15,-8,0), yaw -40 (rounds 1-4) -- the reference algorithm does NOT register it, on either side
  cfg4  indoor_pair       3DMatch-like fragments.  variant "registering" (default; --config 4): three fused depth frustums, cluttered room,
                          CENTIMETRES, poses 0.4 m / 12 deg apart.  variant "surveyed" (--config 14): SURVEY.md §8d's single frustum in metres,
                          poses 0.8 m / 25 deg apart, voxel 0.025 / r_pca 0.10 / R_nms 0.30 (rounds 1-4: 19-53 keypoints, 1 of 64 pairs accepted)
  cfg5  tls_pair(10M)     200x200 m scene, station B = (38,12,0), yaw 55, levelled
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def seed_for(config_id: int, pair_id: int = 0) -> int:
    """seed = 0x5EED0000 + 256*config_id + pair_id."""
    return 0x5EED0000 + 256 * int(config_id) + int(pair_id)


class SplitMix64:
    """Counter-based SplitMix64 stream; draws are vectorised."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        self.ctr = 0

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(self.ctr + 1, self.ctr + n + 1, dtype=np.uint64)
            z = self.seed + idx * _GAMMA
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        self.ctr += n
        return z

    def uniform(self, n: int) -> np.ndarray:
        """f64 in [0,1) with 53 random bits."""
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, n: int) -> np.ndarray:
        """Box-Muller (both branches used), f64."""
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)  # (0,1]
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return z[:n]

    def permutation(self, n: int) -> np.ndarray:
        return np.argsort(self.u64(n), kind="stable")


def rot_zyx(yaw_deg: float, pitch_deg: float, roll_deg: float) -> np.ndarray:
    """R = Rz(yaw) Ry(pitch) Rx(roll), f64."""
    a, b, c = (math.radians(v) for v in (yaw_deg, pitch_deg, roll_deg))
    rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1.0]])
    ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1.0, 0], [-math.sin(b), 0, math.cos(b)]])
    rx = np.array([[1.0, 0, 0], [0, math.cos(c), -math.sin(c)], [0, math.sin(c), math.cos(c)]])
    return rz @ ry @ rx


def rt44(R: np.ndarray, t) -> np.ndarray:
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = np.asarray(t, dtype=np.float64)
    return M


@dataclass
class Pair:
    """One registration job. `gt` maps Source -> Target (same sense as Rt_final, main:145)."""

    source: np.ndarray  # (N,3) f32
    target: np.ndarray  # (N,3) f32
    gt: np.ndarray  # (4,4) f64
    name: str = ""
    kp_source: np.ndarray | None = None  # explicit keypoint rows (cfg1 only)
    kp_target: np.ndarray | None = None


# --------------------------------------------------------------------------- cfg1
def gauss_pair(n: int = 50_000, seed: int | None = None, n_kp: int = 2000) -> Pair:
    """cfg1: T ~ N(0, diag(10,6,2)^2); S = GT^-1(T) + N(0, 0.01^2), row-permuted."""
    rng = SplitMix64(seed_for(1) if seed is None else seed)
    T = rng.normal(3 * n).reshape(n, 3) * np.array([10.0, 6.0, 2.0])
    R = rot_zyx(10.0, 3.0, 2.0)
    t = np.array([0.8, -0.5, 0.2])
    S = (T - t) @ R  # R^T (T - t), row-vector form
    S = S + 0.01 * rng.normal(3 * n).reshape(n, 3)
    perm = rng.permutation(n)
    S = S[perm]
    # Explicit keypoints (Gaussian blobs have no edges, the curvature detector rejects them):
    # target rows 0..n_kp-1 and the source rows holding the same physical points, in source order.
    n_kp = min(n_kp, n)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    kps = np.sort(inv[:n_kp]).astype(np.int32)
    kpt = np.arange(n_kp, dtype=np.int32)
    return Pair(S.astype(np.float32), T.astype(np.float32), rt44(R, t), "gauss%d" % n, kps, kpt)


# --------------------------------------------------------------------------- TLS scene
@dataclass
class Scene:
    half: float  # ground plane is [-half, half]^2 at z = 0
    boxes: np.ndarray  # (B,6): x0,y0,x1,y1,z0,z1   (z0 = 0)
    cyls: np.ndarray  # (C,4): cx,cy,r,h


def make_scene(rng: SplitMix64, half: float, n_box: int, n_cyl: int, keepout) -> Scene:
    """Axis-aligned boxes (5-20 m footprint, 3-15 m high) and vertical cylinders
    (r 0.1-0.4 m, h 3-8 m); nothing within 4 m of a scanner station."""
    keepout = np.asarray(keepout, dtype=np.float64).reshape(-1, 2)
    boxes = []
    while len(boxes) < n_box:
        u = rng.uniform(5)
        sx, sy = 5 + 15 * u[0], 5 + 15 * u[1]
        cx = (2 * u[2] - 1) * (half - sx / 2 - 1)
        cy = (2 * u[3] - 1) * (half - sy / 2 - 1)
        h = 3 + 12 * u[4]
        x0, x1, y0, y1 = cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2
        dx = np.maximum(np.maximum(x0 - keepout[:, 0], keepout[:, 0] - x1), 0)
        dy = np.maximum(np.maximum(y0 - keepout[:, 1], keepout[:, 1] - y1), 0)
        if np.any(np.hypot(dx, dy) < 4.0):
            continue
        boxes.append([x0, y0, x1, y1, 0.0, h])
    cyls = []
    while len(cyls) < n_cyl:
        u = rng.uniform(4)
        cx, cy = (2 * u[0] - 1) * (half - 1), (2 * u[1] - 1) * (half - 1)
        if np.any(np.hypot(cx - keepout[:, 0], cy - keepout[:, 1]) < 4.0):
            continue
        cyls.append([cx, cy, 0.1 + 0.3 * u[2], 3 + 5 * u[3]])
    return Scene(half, np.array(boxes), np.array(cyls))


def _raycast(scene: Scene, o: np.ndarray, d: np.ndarray, tmax: float) -> np.ndarray:
    """Nearest hit distance along unit rays o + t d (inf = miss). o:(3,), d:(n,3)."""
    n = d.shape[0]
    best = np.full(n, np.inf)
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground z = 0
        t = -o[2] / dz
        px, py = o[0] + t * dx, o[1] + t * dy
        ok = (t > 0) & (np.abs(px) <= scene.half) & (np.abs(py) <= scene.half)
        best = np.where(ok, t, best)
        ix, iy, iz = 1.0 / dx, 1.0 / dy, 1.0 / dz
        for b in scene.boxes:  # slab test
            t0x, t1x = (b[0] - o[0]) * ix, (b[2] - o[0]) * ix
            t0y, t1y = (b[1] - o[1]) * iy, (b[3] - o[1]) * iy
            t0z, t1z = (b[4] - o[2]) * iz, (b[5] - o[2]) * iz
            tn = np.maximum(np.maximum(np.minimum(t0x, t1x), np.minimum(t0y, t1y)), np.minimum(t0z, t1z))
            tf = np.minimum(np.minimum(np.maximum(t0x, t1x), np.maximum(t0y, t1y)), np.maximum(t0z, t1z))
            ok = (tf >= tn) & (tn > 0) & (tn < best)
            best = np.where(ok, tn, best)
        a = dx * dx + dy * dy
        for c in scene.cyls:  # vertical cylinder side surface
            ox, oy = o[0] - c[0], o[1] - c[1]
            bq = ox * dx + oy * dy
            cq = ox * ox + oy * oy - c[2] * c[2]
            disc = bq * bq - a * cq
            t = (-bq - np.sqrt(np.maximum(disc, 0))) / a
            z = o[2] + t * dz
            ok = (disc > 0) & (t > 0) & (z >= 0) & (z <= c[3]) & (t < best)
            best = np.where(ok, t, best)
    best[best > tmax] = np.inf
    return best


def _scan(scene: Scene, rng: SplitMix64, pos, R: np.ndarray, n_hits: int, tmax: float,
          sigma: float, chunk: int = 1 << 19) -> np.ndarray:
    """Scanner at `pos` with attitude R (scanner->world): az in [0,2pi), el in [-30,60] deg,
    uniform angular grid + jitter, range noise sigma. Returns (n_hits,3) f32 in the SCANNER frame."""
    pos = np.asarray(pos, dtype=np.float64)
    el0, el1 = math.radians(-30.0), math.radians(60.0)
    # a first estimate of the hit fraction fixes the angular grid; rows are then cast until n_hits
    n_az = int(math.sqrt(n_hits * 1.35 * 2 * math.pi / (el1 - el0)))
    n_el = max(8, int(n_az * (el1 - el0) / (2 * math.pi)))
    d_az, d_el = 2 * math.pi / n_az, (el1 - el0) / n_el
    out = []
    got = 0
    rows_per_chunk = max(1, chunk // n_az)
    row = 0
    # rows are visited in a bit-reversed-like stride so a truncated scan still covers all elevations
    order = np.argsort(SplitMix64(int(rng.seed) ^ 0xABCDEF).u64(n_el), kind="stable")
    while got < n_hits:
        if row >= n_el:  # grid exhausted: refine with a fresh jittered pass
            row = 0
        rows = order[row:row + rows_per_chunk]
        row += rows_per_chunk
        m = rows.size * n_az
        j = rng.uniform(2 * m)
        az = (np.tile(np.arange(n_az), rows.size) + j[:m]) * d_az
        el = el0 + (np.repeat(rows, n_az) + j[m:]) * d_el
        ce = np.cos(el)
        d_s = np.stack([ce * np.cos(az), ce * np.sin(az), np.sin(el)], axis=1)
        d_w = d_s @ R.T
        t = _raycast(scene, pos, d_w, tmax)
        hit = np.isfinite(t)
        t = t[hit] + sigma * rng.normal(m)[hit]
        p = d_s[hit] * t[:, None]
        out.append(p.astype(np.float32))
        got += p.shape[0]
    return np.concatenate(out)[:n_hits]


def tls_pair(n_hits: int = 1_000_000, config_id: int = 2, pair_id: int = 0, variant: str = "registering") -> Pair:
    """cfg2/3/5. Target = station A (identity attitude), Source = station B.  `variant` only matters for cfg3."""
    if variant not in ("registering", "surveyed"):
        raise ValueError("variant must be 'registering' or 'surveyed'")
    rng = SplitMix64(seed_for(config_id, pair_id))
    if config_id == 3 and variant == "surveyed":
        half, tmax, b_xy, yaw, pr = 60.0, 60.0, (15.0, -8.0), -40.0, 1.0  # SURVEY.md §8d as written (rounds 1-4)
    elif config_id == 3:
        # FPFH + reciprocal NN has no global stage: its energy is CD = ED / FD^(1/k), i.e. the Euclidean distance as soon as the histograms
        # agree (they do on man-made surfaces: median |correlation| 0.9998), so the pair must start inside the basin of a reciprocal-NN ICP.
        # 17 m / 40 deg (rounds 1-4) ended 0.9 rad / 16 m from the truth on BOTH sides; 5.6 m / 12 deg registers (oracle at full size, seeds
        # 0..3: 22-27 iterations, within 0.009 rad / 0.12 m of ground truth)
        half, tmax, b_xy, yaw, pr = 60.0, 60.0, (5.0, -2.5), -12.0, 1.0
    elif config_id == 5:
        half, tmax, b_xy, yaw, pr = 100.0, 100.0, (38.0, 12.0), 55.0, 0.0
    else:
        half, tmax, b_xy, yaw, pr = 60.0, 60.0, (12.0, 5.0), 30.0, 1.0
    scale = (half / 60.0) ** 2
    scene = make_scene(rng, half, int(40 * scale), int(60 * scale), [(0.0, 0.0), b_xy])
    hs = 1.6
    pa = np.array([0.0, 0.0, hs])
    pb = np.array([b_xy[0], b_xy[1], hs])
    Ra = np.eye(3)
    Rb = rot_zyx(yaw, pr, pr)
    T = _scan(scene, rng, pa, Ra, n_hits, tmax, 0.003)
    S = _scan(scene, rng, pb, Rb, n_hits, tmax, 0.003)
    # p_A = Ra^T (Rb p_B + pb - pa)
    gt = rt44(Ra.T @ Rb, Ra.T @ (pb - pa))
    return Pair(S, T, gt, "tls%d_cfg%d" % (n_hits, config_id))


# --------------------------------------------------------------------------- cfg4
INDOOR_UNIT_M = 0.01  # cfg4 coordinates are CENTIMETRES (see indoor_pair)


def indoor_pair_surveyed(pair_id: int = 0, n_pts: int = 100_000) -> Pair:
    """cfg4 as SURVEY.md §8d wrote it (rounds 1-4): room 6 x 5 x 3 m + 8-15 furniture boxes, ONE pin-hole depth frustum 58 x 45 deg,
    0.4-3.5 m, two poses ~0.8 m / 25 deg apart, sigma 2 mm, METRES in each camera frame (x fwd, y left, z up)."""
    rng = SplitMix64(seed_for(4, pair_id))
    nb = 8 + int(rng.uniform(1)[0] * 8)
    boxes = []
    for _ in range(nb):
        u = rng.uniform(5)
        sx, sy, h = 0.3 + 1.2 * u[0], 0.3 + 1.2 * u[1], 0.3 + 1.5 * u[2]
        cx, cy = (2 * u[3] - 1) * (3.0 - sx / 2), (2 * u[4] - 1) * (2.5 - sy / 2)
        if math.hypot(cx, cy) < 1.0 + max(sx, sy) / 2:
            continue
        boxes.append([cx - sx / 2, cy - sy / 2, cx + sx / 2, cy + sy / 2, 0.0, h])
    walls = np.array([[-3.2, -2.5, -3.0, 2.5, 0, 3.0], [3.0, -2.5, 3.2, 2.5, 0, 3.0],
                      [-3.0, -2.7, 3.0, -2.5, 0, 3.0], [-3.0, 2.5, 3.0, 2.7, 0, 3.0],
                      [-3.2, -2.7, 3.2, 2.7, 3.0, 3.2]])
    scene = Scene(3.0, np.concatenate([np.array(boxes).reshape(-1, 6), walls]), np.zeros((0, 4)))
    u = rng.uniform(6)
    yaw0 = 360.0 * u[0]
    pa = np.array([0.4 * (2 * u[1] - 1), 0.4 * (2 * u[2] - 1), 1.3])
    Ra = rot_zyx(yaw0, 8.0, 0.0)
    pb = pa + rot_zyx(yaw0 + 90.0, 0, 0) @ np.array([0.8, 0.0, 0.0]) * (0.8 + 0.4 * u[3])
    Rb = rot_zyx(yaw0 + 25.0 * (1 if u[4] > 0.5 else -1), 8.0 + 4 * (u[5] - 0.5), 2.0)

    def frame(pos, R):
        hx, hy = math.tan(math.radians(29.0)), math.tan(math.radians(22.5))
        side = int(math.sqrt(n_pts * 1.6))
        out, got = [], 0
        while got < n_pts:
            j = rng.uniform(2 * side * side)
            gx = (np.tile(np.arange(side), side) + j[: side * side]) / side * 2 - 1
            gy = (np.repeat(np.arange(side), side) + j[side * side:]) / side * 2 - 1
            d = np.stack([np.ones_like(gx), gx * hx, gy * hy], axis=1)
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            t = _raycast(scene, pos, d @ R.T, 3.5)
            hit = np.isfinite(t) & (t > 0.4)
            t = t[hit] + 0.002 * rng.normal(t.size)[hit]
            out.append((d[hit] * t[:, None]).astype(np.float32))
            got += out[-1].shape[0]
        return np.concatenate(out)[:n_pts]

    T = frame(pa, Ra)
    S = frame(pb, Rb)
    return Pair(S, T, rt44(Ra.T @ Rb, Ra.T @ (pb - pa)), "indoor_surveyed%d" % pair_id)


def indoor_pair(pair_id: int = 0, n_pts: int = 100_000) -> Pair:
    """cfg4 (variant "registering"; `indoor_pair_surveyed` is SURVEY.md §8d's): 3DMatch-like fragment pairs.  Room 6 x 5 x 3 m with 8-15 furniture boxes, ~30 small objects (5-40 cm boxes on the floor or on
    shelves) and ~12 thin posts / lamp stands (vertical cylinders, r 3-15 cm).  A fragment is what a 3DMatch fragment is: SEVERAL depth frames
    fused in the frame of the middle one -- three pin-hole frustums of 58 x 45 deg, 0.4-3.5 m, from one position, 20 deg of yaw apart, range
    noise sigma 2 mm.  The two fragments are ~0.4 m / 12 deg apart.  Coordinates are in CENTIMETRES in each fragment's frame (x fwd, y left,
    z up): the reference's energy balances Euclidean and feature distance with constants made for TLS scenes -- ED is scaled by 0.005 x the
    bounding-box magnitude and the penalty has a floor of 5 (ghicp_reg.h:40, ghicp_reg.cpp:275) -- so a 6 m room in metres has ED x 0.05
    against Hamming distances of 50-200 and the loop never uses geometry (rounds 1-4: 19-53 keypoints, 1 of 64 pairs accepted).  In
    centimetres (bounding-box magnitude ~1000, voxel 1.25, r_pca 5, R_nms 15) the same scenes give 96-312 keypoints per fragment and the
    reference's verdict accepts 56 of 64 pairs, 39 of them within 0.05 rad / 0.5 m of ground truth (oracle run)."""
    rng = SplitMix64(seed_for(4, pair_id))
    nb = 8 + int(rng.uniform(1)[0] * 8)
    boxes = []
    for _ in range(nb):  # furniture
        u = rng.uniform(5)
        sx, sy, h = 0.3 + 1.2 * u[0], 0.3 + 1.2 * u[1], 0.3 + 1.5 * u[2]
        cx, cy = (2 * u[3] - 1) * (3.0 - sx / 2), (2 * u[4] - 1) * (2.5 - sy / 2)
        if math.hypot(cx, cy) < 1.0 + max(sx, sy) / 2:
            continue
        boxes.append([cx - sx / 2, cy - sy / 2, cx + sx / 2, cy + sy / 2, 0.0, h])
    for _ in range(30):  # clutter: small boxes on the floor or at shelf height
        u = rng.uniform(6)
        sx, sy, h = 0.05 + 0.35 * u[0], 0.05 + 0.35 * u[1], 0.05 + 0.5 * u[2]
        cx, cy = (2 * u[3] - 1) * 2.8, (2 * u[4] - 1) * 2.3
        if math.hypot(cx, cy) < 0.8:
            continue
        z0 = 0.0 if u[5] < 0.5 else 0.4 + 1.6 * (u[5] - 0.5) * 2
        boxes.append([cx - sx / 2, cy - sy / 2, cx + sx / 2, cy + sy / 2, z0, z0 + h])
    cyls = []
    for _ in range(12):  # posts, lamp stands, table legs
        u = rng.uniform(4)
        cx, cy = (2 * u[0] - 1) * 2.8, (2 * u[1] - 1) * 2.3
        if math.hypot(cx, cy) < 0.8:
            continue
        cyls.append([cx, cy, 0.03 + 0.12 * u[2], 0.3 + 2.0 * u[3]])
    walls = np.array([[-3.2, -2.5, -3.0, 2.5, 0, 3.0], [3.0, -2.5, 3.2, 2.5, 0, 3.0],
                      [-3.0, -2.7, 3.0, -2.5, 0, 3.0], [-3.0, 2.5, 3.0, 2.7, 0, 3.0],
                      [-3.2, -2.7, 3.2, 2.7, 3.0, 3.2]])
    scene = Scene(3.0, np.concatenate([np.array(boxes).reshape(-1, 6), walls]), np.array(cyls).reshape(-1, 4))
    u = rng.uniform(6)
    yaw0 = 360.0 * u[0]
    pa = np.array([0.4 * (2 * u[1] - 1), 0.4 * (2 * u[2] - 1), 1.3])
    Ra = rot_zyx(yaw0, 8.0, 0.0)
    pb = pa + rot_zyx(yaw0 + 90.0, 0, 0) @ np.array([0.4, 0.0, 0.0]) * (0.8 + 0.4 * u[3])
    Rb = rot_zyx(yaw0 + 12.0 * (1 if u[4] > 0.5 else -1), 8.0 + 4 * (u[5] - 0.5), 2.0)
    views, view_step = 3, 20.0

    def fragment(pos, R):
        hx, hy = math.tan(math.radians(29.0)), math.tan(math.radians(22.5))
        per = n_pts // views
        side = max(2, int(math.sqrt(per * 1.6)))
        parts = []
        for v in range(views):
            Rv = rot_zyx((v - (views - 1) / 2) * view_step, 0, 0)  # this frame's camera in the fragment (middle camera) frame
            want = per if v < views - 1 else n_pts - per * (views - 1)
            out, got = [], 0
            while got < want:
                j = rng.uniform(2 * side * side)
                gx = (np.tile(np.arange(side), side) + j[: side * side]) / side * 2 - 1
                gy = (np.repeat(np.arange(side), side) + j[side * side:]) / side * 2 - 1
                d = np.stack([np.ones_like(gx), gx * hx, gy * hy], axis=1)
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                df = d @ Rv.T
                t = _raycast(scene, pos, df @ R.T, 3.5)
                hit = np.isfinite(t) & (t > 0.4)
                t = t[hit] + 0.002 * rng.normal(t.size)[hit]
                out.append(df[hit] * t[:, None])
                got += out[-1].shape[0]
            parts.append(np.concatenate(out)[:want])
        return (np.concatenate(parts) / INDOOR_UNIT_M).astype(np.float32)

    T = fragment(pa, Ra)
    S = fragment(pb, Rb)
    gt = rt44(Ra.T @ Rb, Ra.T @ (pb - pa) / INDOOR_UNIT_M)
    return Pair(S, T, gt, "indoor%d" % pair_id)


# --------------------------------------------------------------------------- misc
def bsc_pattern_glibc() -> np.ndarray:
    """The 49 (a,b) cell pairs `BSCEncoder(R, 7, build_sample_pattern=true)` draws from glibc's
    unseeded rand() (binary_feature_extraction.hpp:75-88; values probed in SURVEY.md §8a Q2)."""
    flat = [15, 39, 37, 5, 29, 10, 24, 23, 8, 17, 9, 47, 6, 18, 20, 21, 17, 22, 3, 21, 30, 47, 33, 48,
            29, 5, 5, 0, 45, 47, 10, 30, 8, 35, 9, 16, 8, 18, 19, 14, 41, 45, 41, 9, 23, 15, 38, 26,
            42, 43, 46, 4, 22, 31, 9, 27, 32, 5, 31, 47, 40, 39, 38, 0, 12, 3, 23, 31, 17, 16, 32, 14,
            30, 7, 35, 24, 33, 28, 18, 4, 7, 16, 8, 29, 47, 18, 12, 35, 28, 43, 39, 20, 39, 28, 25, 7, 31, 0]
    return np.array(flat, dtype=np.int32).reshape(49, 2)


def bsc_pattern_zero() -> np.ndarray:
    """What a fresh checkout gets: sample_pattern.txt is missing so every pair reads as (0,0)
    (binary_feature_extraction.hpp:107-116)."""
    return np.zeros((49, 2), dtype=np.int32)


def rot_err(Ra: np.ndarray, Rb: np.ndarray) -> float:
    """||Ra Rb^T - I||_F  (rotation parity metric)."""
    return float(np.linalg.norm(Ra[:3, :3] @ Rb[:3, :3].T - np.eye(3)))


def trans_err(Ma: np.ndarray, Mb: np.ndarray) -> float:
    return float(np.linalg.norm(Ma[:3, 3] - Mb[:3, 3]))
