"""Seeded synthetic rotating-lidar streams (S64 / S128).

A *stream* is a sequence of firings (one vertical set of ``num_rows`` laser returns each, sensor frame,
NaN x = no return) plus one ``odom_from_sensor`` pose per firing, i.e. exactly what the reference's
``ContinuousClustering::addFiring`` consumes (src/clustering/continuous_clustering.cpp:88-93; firing shape as
built by src/tools/kitti_demo.cpp:123-159).

Scene (world = odom frame, sensor starts at the origin, ground 1.73 m below it): ground plane, ``n_objects``
vertical cylinders, an enclosing wall ring (optionally with gaps). Rays are cast per laser; range noise +-1 cm,
2 % dropouts. The generator is written against an array namespace so that the same code runs with numpy (tests,
golden fixtures) and with torch on the GPU (bench.py generates its inputs directly in HBM).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np


@dataclass
class SensorModel:
    num_rows: int = 64
    num_columns: int = 2200
    incl_top_deg: float = 2.0
    incl_bottom_deg: float = -24.8
    clockwise: bool = True
    # per-laser azimuth offsets (degrees), repeated cyclically over the rows; () = all lasers share one azimuth
    azimuth_offsets_deg: tuple = ()
    firing_period_s: float = 1.0 / 22000.0

    @staticmethod
    def s64() -> "SensorModel":
        return SensorModel()

    @staticmethod
    def s128(with_offsets: bool = True) -> "SensorModel":
        offs = (-6.4, -4.5, -2.7, -0.9, 0.9, 2.7, 4.5, 6.4) if with_offsets else ()
        return SensorModel(num_rows=128, num_columns=1700, incl_top_deg=15.0, incl_bottom_deg=-25.0,
                           azimuth_offsets_deg=offs, firing_period_s=1.0 / 17000.0)


@dataclass
class SceneModel:
    n_objects: int = 60
    object_range: tuple = (8.0, 35.0)
    object_radius: tuple = (0.3, 1.1)
    object_top_z: float = 1.0
    ground_z: float = -1.73
    max_range: float = 120.0
    wall_radius: float = 45.0
    wall_top_z: float = 3.0
    # list of (start_deg, end_deg) world azimuth intervals where the wall is missing; () = unbroken ring,
    # which drives the reference's "cluster exceeding one rotation" forced-finish path (cc.cpp:913-919)
    wall_gaps_deg: tuple = ((20.0, 32.0), (140.0, 155.0), (250.0, 262.0))
    range_noise: float = 0.01
    dropout: float = 0.02
    # vegetation / fences: (start_deg, end_deg, r0, r1, density) — inside the world azimuth interval a ray returns, with probability `density`,
    # from a random range in [r0, r1] instead of from what lies behind (thin returns at jittered ranges: many small trees that stay unfinished
    # side by side, and neighbours that join them a few columns later). () = none
    clutter: tuple = ()

    @staticmethod
    def cluttered(density: float = 0.1) -> "SceneModel":
        """The scene of the bench with vegetation in front of it over half the circle: sparse returns at 3 .. 30 m (isolated leaves: more
        unfinished trees side by side than the batch-parallel association has lanes for) in two sectors and a dense shell at 2 .. 8 m in a third
        (neighbours a few columns apart keep joining trees: points with more link candidates than the window scan records)."""
        return SceneModel(clutter=((30.0, 120.0, 3.0, 30.0, density), (200.0, 260.0, 3.0, 30.0, density), (300.0, 330.0, 2.0, 8.0, density)))

    @staticmethod
    def sparse_clutter(density: float = 0.1) -> "SceneModel":
        return SceneModel(clutter=((0.0, 360.0, 3.0, 30.0, density),))

    @staticmethod
    def near_clutter(density: float = 0.1) -> "SceneModel":
        return SceneModel(clutter=((0.0, 360.0, 2.0, 8.0, density),))


@dataclass
class Motion:
    velocity: tuple = (0.0, 0.0, 0.0)   # m/s in the odom frame
    yaw_rate: float = 0.0               # rad/s
    kind: str = "static"

    @staticmethod
    def static() -> "Motion":
        return Motion()

    @staticmethod
    def translate(v: float = 10.0) -> "Motion":
        return Motion(velocity=(v, 0.0, 0.0), kind="translate")

    @staticmethod
    def turn(v: float = 10.0, yaw_rate: float = 0.2) -> "Motion":
        return Motion(velocity=(v, 0.0, 0.0), yaw_rate=yaw_rate, kind="turn")


@dataclass
class Stream:
    xyz: np.ndarray        # [F, rows, 3] float32, sensor frame
    intensity: np.ndarray  # [F, rows] uint8
    poses: np.ndarray      # [F, 12] float64, odom_from_sensor as row-major 3x4 [R|t]
    sensor: SensorModel = field(default_factory=SensorModel)
    hit: np.ndarray | None = None  # [F, rows] uint16: 0 no return, 1 ground, 2 wall, 3 + i object i (ground truth for evaluation)

    @property
    def n_firings(self) -> int:
        return int(self.xyz.shape[0])


def _scene_params(scene: SceneModel, seed: int):
    rng = np.random.default_rng(seed)
    n = scene.n_objects
    rr = rng.uniform(scene.object_range[0], scene.object_range[1], n)
    aa = rng.uniform(0.0, 2 * math.pi, n)
    rad = rng.uniform(scene.object_radius[0], scene.object_radius[1], n)
    cx, cy = rr * np.cos(aa), rr * np.sin(aa)
    return cx, cy, rad


def _cast(xp, o, d, cx, cy, rad, scene: SceneModel):
    """o: [F,1,3] ray origins, d: [F,R,3] unit directions (odom frame). Returns (range t [F,R] (inf = no hit), hit id [F,R])."""
    inf = float("inf")
    ox, oy, oz = o[..., 0], o[..., 1], o[..., 2]
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    # ground plane
    tg = xp.where(dz < 0, (scene.ground_z - oz) / xp.where(dz < 0, dz, -1.0), inf)
    t = tg
    hit = xp.where(tg < inf, 1, 0)
    dxy2 = dx * dx + dy * dy
    dxy2 = xp.where(dxy2 > 1e-12, dxy2, 1e-12)
    # cylinders: [F,R,N]
    if cx.shape[0] > 0:
        ex = ox[..., None] - cx
        ey = oy[..., None] - cy
        b = ex * dx[..., None] + ey * dy[..., None]
        c = ex * ex + ey * ey - rad * rad
        disc = b * b - dxy2[..., None] * c
        ok = disc > 0
        sq = xp.sqrt(xp.where(ok, disc, 0.0))
        tc = (-b - sq) / dxy2[..., None]
        zc = oz[..., None] + tc * dz[..., None]
        ok = ok & (tc > 0.05) & (zc <= scene.object_top_z) & (zc >= scene.ground_z)
        tc = xp.where(ok, tc, inf)
        tmin = tc.min(-1) if xp is np else tc.min(-1).values
        imin = tc.argmin(-1)
        hit = xp.where(tmin < t, imin + 3, hit)
        t = xp.minimum(t, tmin)
    # wall ring centred on the world origin, hit from the inside (far root)
    if scene.wall_radius > 0:
        b = ox * dx + oy * dy
        c = ox * ox + oy * oy - scene.wall_radius ** 2
        disc = b * b - dxy2 * c
        ok = disc > 0
        sq = xp.sqrt(xp.where(ok, disc, 0.0))
        tw = (-b + sq) / dxy2
        hx, hy, hz = ox + tw * dx, oy + tw * dy, oz + tw * dz
        ok = ok & (tw > 0.05) & (hz <= scene.wall_top_z) & (hz >= scene.ground_z)
        if scene.wall_gaps_deg:
            ang = xp.arctan2(hy, hx) * (180.0 / math.pi)
            ang = xp.where(ang < 0, ang + 360.0, ang)
            for g0, g1 in scene.wall_gaps_deg:
                ok = ok & ~((ang >= g0) & (ang <= g1))
        tw = xp.where(ok, tw, inf)
        hit = xp.where(tw < t, 2, hit)
        t = xp.minimum(t, tw)
    return t, hit


def make_stream(n_firings: int, seed: int = 1234, sensor: SensorModel | None = None, scene: SceneModel | None = None,
                motion: Motion | None = None, start_column: int = 0, xp=np, device=None, chunk: int = 2200) -> Stream:
    """Generate ``n_firings`` consecutive firings. With ``xp=torch`` the arrays are torch tensors on ``device``."""
    sensor = sensor or SensorModel.s64()
    scene = scene or SceneModel()
    motion = motion or Motion.static()
    is_np = xp is np
    R, C = sensor.num_rows, sensor.num_columns
    cx, cy, rad = _scene_params(scene, seed)
    incl = np.deg2rad(np.linspace(sensor.incl_top_deg, sensor.incl_bottom_deg, R))
    if sensor.azimuth_offsets_deg:
        offs = np.deg2rad(np.array([sensor.azimuth_offsets_deg[r % len(sensor.azimuth_offsets_deg)] for r in range(R)]))
    else:
        offs = np.zeros(R)
    width = 2 * math.pi / C

    def A(a, dtype=None):
        if is_np:
            return np.asarray(a, dtype=dtype or np.float64)
        import torch
        return torch.as_tensor(np.asarray(a, dtype=dtype or np.float64), device=device)

    cxx, cyy, radd, incl_x, offs_x = A(cx), A(cy), A(rad), A(incl), A(offs)
    rng = np.random.default_rng(seed + 7919)
    if not is_np:
        import torch
    xyz_parts, int_parts, pose_parts, hit_parts = [], [], [], []
    for f0 in range(0, n_firings, chunk):
        f1 = min(n_firings, f0 + chunk)
        k = A(np.arange(f0, f1, dtype=np.float64))
        tsec = k * sensor.firing_period_s
        base = (k + start_column + 0.5) * width
        # clockwise sensor: azimuth decreases from +pi (cc.cpp:146-148); counter-clockwise: increases from -pi
        az0 = (math.pi - base) if sensor.clockwise else (-math.pi + base)
        az = az0[:, None] + (offs_x[None, :] if sensor.clockwise else -offs_x[None, :])
        ce, se = xp.cos(incl_x)[None, :], xp.sin(incl_x)[None, :]
        ds = xp.stack([ce * xp.cos(az), ce * xp.sin(az), se * xp.ones_like(az)], -1)  # sensor-frame directions [F,R,3]
        yaw = tsec * motion.yaw_rate
        cyw, syw = xp.cos(yaw), xp.sin(yaw)
        if motion.yaw_rate != 0.0:
            # integrate the velocity in the body frame
            wv = motion.yaw_rate
            px = motion.velocity[0] * syw / wv
            py = motion.velocity[0] * (1.0 - cyw) / wv
        else:
            px = tsec * motion.velocity[0]
            py = tsec * motion.velocity[1]
        pz = tsec * motion.velocity[2]
        # world directions
        dwx = cyw[:, None] * ds[..., 0] - syw[:, None] * ds[..., 1]
        dwy = syw[:, None] * ds[..., 0] + cyw[:, None] * ds[..., 1]
        dw = xp.stack([dwx, dwy, ds[..., 2]], -1)
        o = xp.stack([px, py, pz], -1)[:, None, :]
        t, hit = _cast(xp, o, dw, cxx, cyy, radd, scene)
        F = f1 - f0
        if is_np:
            noise = rng.uniform(-scene.range_noise, scene.range_noise, (F, R))
            drop = rng.uniform(0, 1, (F, R)) < scene.dropout
            inten = rng.integers(0, 256, (F, R), dtype=np.uint8)
        else:
            g = torch.Generator(device=device)
            g.manual_seed(seed * 1000003 + f0)
            noise = (torch.rand((F, R), generator=g, device=device, dtype=torch.float64) * 2 - 1) * scene.range_noise
            drop = torch.rand((F, R), generator=g, device=device) < scene.dropout
            inten = torch.randint(0, 256, (F, R), generator=g, device=device, dtype=torch.uint8)
        if scene.clutter:
            ang = xp.arctan2(dw[..., 1], dw[..., 0]) * (180.0 / math.pi)
            ang = xp.where(ang < 0, ang + 360.0, ang)
            for ci, (a0, a1, r0, r1, dens) in enumerate(scene.clutter):
                if is_np:
                    u1, u2 = rng.uniform(0, 1, (F, R)), rng.uniform(0, 1, (F, R))
                else:
                    u1 = torch.rand((F, R), generator=g, device=device, dtype=torch.float64)
                    u2 = torch.rand((F, R), generator=g, device=device, dtype=torch.float64)
                tl = r0 + (r1 - r0) * u2
                zl = o[..., 2] + tl * dw[..., 2]
                ok = (ang >= a0) & (ang <= a1) & (u1 < dens) & (zl >= scene.ground_z + 0.25) & (zl <= scene.wall_top_z) & (tl < t)
                t = xp.where(ok, tl, t)
                hit = xp.where(ok, 3 + scene.n_objects + ci, hit)
        valid = (t < scene.max_range) & ~drop
        hit = xp.where(valid, hit, 0)
        tt = xp.where(valid, t + noise, float("nan"))
        pts = ds * tt[..., None]
        pose = xp.stack([cyw, -syw, xp.zeros_like(cyw), px,
                         syw, cyw, xp.zeros_like(cyw), py,
                         xp.zeros_like(cyw), xp.zeros_like(cyw), xp.ones_like(cyw), pz], -1)
        if is_np:
            xyz_parts.append(pts.astype(np.float32))
        else:
            xyz_parts.append(pts.to(torch.float32))
        int_parts.append(inten)
        pose_parts.append(pose)
        hit_parts.append(hit.astype(np.uint16) if is_np else hit.to(torch.int16))
    if is_np:
        return Stream(np.concatenate(xyz_parts), np.concatenate(int_parts), np.concatenate(pose_parts), sensor, np.concatenate(hit_parts))
    import torch
    return Stream(torch.cat(xyz_parts), torch.cat(int_parts), torch.cat(pose_parts), sensor, torch.cat(hit_parts))
