"""Deterministic synthetic inputs for the scan-to-map path (numpy only).

The reference ships no data (its test/data is git-ignored and the harnesses read
developer-local PCDs: reference test/align.cpp:100-108, config/params.json:10),
so tests and bench.py use this planar "city block" world: a ground plane,
box buildings on a regular pitch and a perimeter wall -- every surface planar, so
the LOAM plane model (reference PCR/src/LoamRegister.cpp:29-45) is valid.

  make_map(n_points, seed)      -> World, float32 (n,4) [x y z intensity], map frame
  make_scan(world, k, seed)     -> float32 (n,4) in the sensor frame, true pose 4x4
  perturb(pose, seed)           -> initial guess = truth o exp(xi)
"""
from dataclasses import dataclass

import numpy as np

PITCH = 50.0       # m, one block
BUILDING = 30.0    # m, footprint side (streets are PITCH-BUILDING = 20 m wide)
HEIGHT = 12.0      # m, building and perimeter-wall height
SENSOR_Z = 1.8     # m
MAX_RANGE = 120.0  # m


@dataclass
class World:
    blocks: int          # blocks per side
    side: float          # world side length, m
    spacing: float       # map sample spacing, m
    boxes: np.ndarray    # (B,4) x0 y0 x1 y1 of the buildings (z in [0,HEIGHT])


def map_origin(blocks):
    """World coordinates of the map frame's origin: a street point SENSOR_Z + 0.2 m above the
    ground, like a SLAM map whose origin is the first key-frame's lidar pose (the reference's
    tf.lidar_height is 2.0, config/params.json:20).  LOAM writes planes as x.p + 1 = 0
    (LoamRegister.cpp:29-35), which cannot represent a plane through the origin, so the
    synthetic surfaces must stay away from it as real ones do."""
    mid = (blocks // 2) * PITCH
    if blocks == 1:
        return np.array([6.0, 4.0, SENSOR_Z + 0.2])
    return np.array([mid + 7.0, mid - 4.0, SENSOR_Z + 0.2])


def _world(blocks):
    c = np.arange(blocks) * PITCH + (PITCH - BUILDING) / 2
    x0, y0 = np.meshgrid(c, c, indexing="ij")
    boxes = np.stack([x0.ravel(), y0.ravel(), x0.ravel() + BUILDING, y0.ravel() + BUILDING], 1)
    return boxes


def _surface_count(blocks, s):
    side = blocks * PITCH
    ground = (side * side - blocks * blocks * BUILDING * BUILDING) / (s * s)
    walls = blocks * blocks * 4 * BUILDING * HEIGHT / (s * s)
    perim = 4 * side * HEIGHT / (s * s)
    return ground + walls + perim


def _lattice(u0, u1, v0, v1, s, rng):
    nu = max(1, int(round((u1 - u0) / s)))
    nv = max(1, int(round((v1 - v0) / s)))
    su, sv = (u1 - u0) / nu, (v1 - v0) / nv
    u, v = np.meshgrid((np.arange(nu) + 0.5) * su + u0, (np.arange(nv) + 0.5) * sv + v0, indexing="ij")
    u = u.ravel() + rng.uniform(-0.4, 0.4, u.size) * su
    v = v.ravel() + rng.uniform(-0.4, 0.4, v.size) * sv
    return u, v


def make_map(n_points, seed=0, noise=0.01, spacing=0.5):
    """Map of exactly n_points surface samples at ~`spacing` m (0.5: the reference's sub-map is
    voxel-filtered at downSampleVoxelGridSize = 0.5, config/params.json:8; NDT wants a denser map --
    "use ndt maybe no need to downsample", config/params.json:7 -- so its tests pass 0.2)."""
    rng = np.random.default_rng(seed)
    blocks = max(1, int(round(np.sqrt(n_points / _surface_count(1, spacing)))))
    while _surface_count(blocks, spacing) < n_points * 0.75:
        blocks += 1
    s = spacing * np.sqrt(_surface_count(blocks, spacing) / (1.04 * n_points))
    side = blocks * PITCH
    boxes = _world(blocks)
    parts = []
    # ground, minus building footprints
    gx, gy = _lattice(0, side, 0, side, s, rng)
    bx = np.floor(gx / PITCH) * PITCH + (PITCH - BUILDING) / 2
    by = np.floor(gy / PITCH) * PITCH + (PITCH - BUILDING) / 2
    keep = ~((gx > bx) & (gx < bx + BUILDING) & (gy > by) & (gy < by + BUILDING))
    parts.append(np.stack([gx[keep], gy[keep], np.zeros(keep.sum())], 1))
    # building walls
    for (x0, y0, x1, y1) in boxes:
        for (fixed, val, a0, a1) in ((0, x0, y0, y1), (0, x1, y0, y1), (1, y0, x0, x1), (1, y1, x0, x1)):
            u, z = _lattice(a0, a1, 0, HEIGHT, s, rng)
            p = np.empty((u.size, 3))
            p[:, fixed] = val
            p[:, 1 - fixed] = u
            p[:, 2] = z
            parts.append(p)
    # perimeter walls
    for (fixed, val) in ((0, 0.0), (0, side), (1, 0.0), (1, side)):
        u, z = _lattice(0, side, 0, HEIGHT, s, rng)
        p = np.empty((u.size, 3))
        p[:, fixed] = val
        p[:, 1 - fixed] = u
        p[:, 2] = z
        parts.append(p)
    pts = np.concatenate(parts, 0)
    pts += rng.normal(0, noise, pts.shape)
    if pts.shape[0] < n_points:  # top up by jittered duplicates (rare: only at tiny n)
        extra = pts[rng.integers(0, pts.shape[0], n_points - pts.shape[0])] + rng.normal(0, 0.05, (n_points - pts.shape[0], 3))
        pts = np.concatenate([pts, extra], 0)
    sel = rng.permutation(pts.shape[0])[:n_points]
    sel.sort()
    out = np.empty((n_points, 4), np.float32)
    out[:, :3] = pts[sel] - map_origin(blocks)
    out[:, 3] = rng.uniform(0, 255, n_points)
    return World(blocks, side, s, boxes), out


def _raycast(world, o, d):
    """Nearest hit distance of rays o + t d (world frame); inf when nothing within MAX_RANGE."""
    n = d.shape[0]
    t = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(d[:, 2] < 0, -o[2] / d[:, 2], np.inf)
        # ground hit must not be under a building
        gx, gy = o[0] + tg * d[:, 0], o[1] + tg * d[:, 1]
        bx = np.floor(gx / PITCH) * PITCH + (PITCH - BUILDING) / 2
        by = np.floor(gy / PITCH) * PITCH + (PITCH - BUILDING) / 2
        under = (gx > bx) & (gx < bx + BUILDING) & (gy > by) & (gy < by + BUILDING)
        inside = (gx >= 0) & (gx <= world.side) & (gy >= 0) & (gy <= world.side)
        t = np.minimum(t, np.where(under | ~inside, np.inf, tg))
        # perimeter
        for (ax, val) in ((0, 0.0), (0, world.side), (1, 0.0), (1, world.side)):
            tp = (val - o[ax]) / d[:, ax]
            z = o[2] + tp * d[:, 2]
            other = o[1 - ax] + tp * d[:, 1 - ax]
            ok = (tp > 0) & (z >= 0) & (z <= HEIGHT) & (other >= 0) & (other <= world.side)
            t = np.minimum(t, np.where(ok, tp, np.inf))
        # buildings near the sensor (slab test, chunked over rays)
        near = world.boxes[(np.abs((world.boxes[:, 0] + world.boxes[:, 2]) / 2 - o[0]) < MAX_RANGE + BUILDING) &
                           (np.abs((world.boxes[:, 1] + world.boxes[:, 3]) / 2 - o[1]) < MAX_RANGE + BUILDING)]
        for c0 in range(0, n, 16384):
            dd = d[c0:c0 + 16384]
            inv = 1.0 / dd[:, None, :2]
            t0 = (near[None, :, 0:2] - o[None, None, :2]) * inv
            t1 = (near[None, :, 2:4] - o[None, None, :2]) * inv
            tmin = np.minimum(t0, t1).max(2)
            tmax = np.maximum(t0, t1).min(2)
            z = o[2] + tmin * dd[:, None, 2]
            ok = (tmax >= tmin) & (tmin > 0) & (z >= 0) & (z <= HEIGHT)
            tb = np.where(ok, tmin, np.inf).min(1) if near.shape[0] else np.full(dd.shape[0], np.inf)
            t[c0:c0 + 16384] = np.minimum(t[c0:c0 + 16384], tb)
    t[t > MAX_RANGE] = np.inf
    return t


def scan_pose(world, k=0, seed=0):
    """True sensor pose of the k-th scan: a drive along a street near the centre."""
    rng = np.random.default_rng([seed, k, 7])
    mid = (world.blocks // 2) * PITCH
    x = mid + 3.0 + 1.5 * k + rng.uniform(-0.2, 0.2)
    y = mid - 2.0 + rng.uniform(-0.5, 0.5)
    if world.blocks == 1:  # single block: stay in the street ring around the building
        x, y = 5.0 + 0.5 * k, 5.0
    yaw = rng.uniform(-np.pi, np.pi)
    roll, pitch = rng.normal(0, 0.01, 2)
    cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = np.array([x, y, SENSOR_Z]) - map_origin(world.blocks)
    return T


def trajectory_pose(world, k, step=0.5):
    """True sensor pose of scan k of a DRIVE: `step` metres per scan along the street through the middle of the world (x grows),
    a gentle weave in y, a slowly changing heading -- what the reference's front end tracks scan after scan
    (frontend/src/LidarOdometry.cpp:160-200), where scan_pose() above draws unrelated poses."""
    mid = (world.blocks // 2) * PITCH
    x = mid + 3.0 + step * k
    y = mid - 2.0 + 0.8 * np.sin(0.11 * k)
    if world.blocks == 1:
        x, y = 5.0 + 0.25 * k, 5.0
    yaw = 0.06 * np.sin(0.07 * k) + 0.02 * np.cos(0.23 * k)
    pitch, roll = 0.004 * np.sin(0.31 * k), 0.003 * np.cos(0.19 * k)
    cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = np.array([x, y, SENSOR_Z]) - map_origin(world.blocks)
    return T


def make_scan(world, k=0, seed=0, beams=64, azimuths=1024, noise=0.02, pose=None):
    """beams*azimuths returns of a spinning lidar (elevation -24.8..+2 deg), in the
    sensor frame.  Rays with no return within MAX_RANGE are re-drawn so the scan has
    exactly beams*azimuths points (BASELINE fixes N_s = 65 536).  pose: the true sensor pose (default: scan_pose(world, k, seed))."""
    rng = np.random.default_rng([seed, k, 11])
    T = scan_pose(world, k, seed) if pose is None else np.array(pose, float)
    n = beams * azimuths
    el = np.deg2rad(np.linspace(-24.8, 2.0, beams))
    az = np.arange(azimuths) * (2 * np.pi / azimuths)
    el_g, az_g = np.meshgrid(el, az, indexing="ij")
    el_g = el_g.ravel()
    az_g = az_g.ravel() + rng.uniform(0, 2 * np.pi / azimuths)
    pending = np.arange(n)
    rng_out = np.zeros(n)
    dirs = np.zeros((n, 3))
    for _ in range(64):
        ds = np.stack([np.cos(el_g[pending]) * np.cos(az_g[pending]), np.cos(el_g[pending]) * np.sin(az_g[pending]),
                       np.sin(el_g[pending])], 1)
        t = _raycast(world, T[:3, 3] + map_origin(world.blocks), ds @ T[:3, :3].T)
        hit = np.isfinite(t)
        rng_out[pending[hit]] = t[hit]
        dirs[pending[hit]] = ds[hit]
        pending = pending[~hit]
        if pending.size == 0:
            break
        el_g[pending] = np.deg2rad(rng.uniform(-24.8, 0.0, pending.size))
        az_g[pending] = rng.uniform(0, 2 * np.pi, pending.size)
    assert pending.size == 0, "scan generation failed to fill all rays"
    r = rng_out + rng.normal(0, noise, n)
    out = np.empty((n, 4), np.float32)
    out[:, :3] = dirs * r[:, None]
    out[:, 3] = rng.uniform(0, 255, n)
    return out, T


def se3_exp(xi):
    """exp of [rho; omega] (same closed form as the reference, common/geometry/manifolds.hpp:33-60)."""
    xi = np.asarray(xi, float)
    rho, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    T = np.eye(4)
    if th < 1e-6:
        T[:3, 3] = rho
        return T
    a = w / th
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    aa = np.outer(a, a)
    T[:3, :3] = np.cos(th) * np.eye(3) + (1 - np.cos(th)) * aa + np.sin(th) * K
    V = np.sin(th) / th * np.eye(3) + (1 - np.sin(th) / th) * aa + (1 - np.cos(th)) / th * K
    T[:3, 3] = V @ rho
    return T


def perturb(T, seed=0, trans=0.3, rot_deg=2.0):
    """Initial guess = truth o exp(xi), |xi_t| <= trans (m), |xi_r| <= rot_deg."""
    rng = np.random.default_rng([seed, 13])
    xi = np.concatenate([rng.uniform(-trans, trans, 3), np.deg2rad(rng.uniform(-rot_deg, rot_deg, 3))])
    return T @ se3_exp(xi)


def pose_error(Ta, Tb):
    """(translation distance m, rotation angle rad) between two 4x4 poses."""
    dt = float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
    Rr = Ta[:3, :3].T @ Tb[:3, :3]
    c = (np.trace(Rr) - 1) / 2
    # angle from the skew part: accurate near zero where acos is not
    s = 0.5 * np.linalg.norm([Rr[2, 1] - Rr[1, 2], Rr[0, 2] - Rr[2, 0], Rr[1, 0] - Rr[0, 1]])
    return dt, float(np.arctan2(s, c))
