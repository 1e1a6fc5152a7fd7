"""ctypes host mirror of include/cc_kitti.h — the KITTI replay path upstream of insertion.

`KittiConverter` turns .bin clouds into pseudo-firings on the GPU (recoverLaserIndices, undoEgoMotionCorrection,
generateRangeImage, makePseudoFiringFromRangeImageColumn of the reference: kitti_loader.cpp:48-210, kitti_demo.cpp:123-159);
the module-level functions are the host pose arithmetic of the same call sites. `synthetic_frame` /
`write_synthetic_sequence` produce KITTI-format inputs (no dataset is available offline). No CPU variant of the per-point work.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import EngineError, load_library

ROWS, COLS = 64, 2200
RECOVER_ROWS, UNDO_EGO_MOTION, RANGE_IMAGE, SHIFT_OCCUPIED, FIRINGS = 1, 2, 4, 8, 16
ALL_STAGES = 31


class Frame(C.Structure):
    _fields_ = [("points", C.c_void_p), ("n_points", C.c_int64), ("laser_index", C.c_void_p), ("stages", C.c_uint32),
                ("num_bins", C.c_int32), ("rotation_start_stamp", C.c_uint64), ("rotation_end_stamp", C.c_uint64),
                ("bin_transforms", C.c_void_p), ("d_xyz", C.c_void_p), ("d_intensity", C.c_void_p), ("d_original_index", C.c_void_p)]


class FrameInfo(C.Structure):
    _fields_ = [("rows_found", C.c_int32), ("max_columns", C.c_int32), ("break_index", C.c_int64), ("skipped", C.c_int64)]


_bound = False


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
        L.cc_kitti_create.argtypes = [C.POINTER(vp), i32, i32, i64, vp]
        L.cc_kitti_destroy.argtypes = [vp]
        L.cc_kitti_destroy.restype = None
        L.cc_kitti_last_error.restype = C.c_char_p
        L.cc_kitti_convert_frames.argtypes = [vp, i32, C.POINTER(Frame)]
        L.cc_kitti_sync.argtypes = [vp]
        L.cc_kitti_hip_stream.argtypes = [vp]
        L.cc_kitti_hip_stream.restype = vp
        L.cc_kitti_frame_result.argtypes = [vp, i32, C.POINTER(FrameInfo), vp, vp, vp]
        L.cc_kitti_pose_interpolate.argtypes = [i64, vp, vp, u64, vp]
        L.cc_kitti_bin_transforms.argtypes = [i64, vp, vp, u64, u64, vp, vp, C.c_int32, C.POINTER(C.c_int32)]
        L.cc_kitti_firing_stamps_and_poses.argtypes = [i64, vp, vp, u64, u64, vp, vp]
        L.cc_kitti_start_end_stamps.argtypes = [i64, vp, vp, vp]
        L.cc_kitti_pose_from_line.argtypes = [vp, vp, vp]
        _bound = True
    return L


def _check(rc: int):
    if rc != 0:
        raise EngineError(rc, _lib().cc_kitti_last_error().decode())


# ---- host pose arithmetic ---------------------------------------------------------------------------------------------------

def _poses_args(stamps, poses):
    stamps = np.ascontiguousarray(stamps, dtype=np.uint64)
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 12)
    assert stamps.shape[0] == poses.shape[0]
    return stamps, poses


def pose_interpolate(stamps, poses, stamp: int) -> np.ndarray:
    stamps, poses = _poses_args(stamps, poses)
    out = np.zeros(12, dtype=np.float64)
    _check(_lib().cc_kitti_pose_interpolate(stamps.shape[0], stamps.ctypes.data, poses.ctypes.data, int(stamp), out.ctypes.data))
    return out


def bin_transforms(stamps, poses, start: int, end: int, mid_pose) -> np.ndarray:
    stamps, poses = _poses_args(stamps, poses)
    mid = np.ascontiguousarray(mid_pose, dtype=np.float64).reshape(12)
    out = np.zeros((512, 12), dtype=np.float64)
    nb = C.c_int32(0)
    _check(_lib().cc_kitti_bin_transforms(stamps.shape[0], stamps.ctypes.data, poses.ctypes.data, int(start), int(end), mid.ctypes.data,
                                          out.ctypes.data, 512, C.byref(nb)))
    return out[: nb.value].copy()


def firing_stamps_and_poses(stamps, poses, start: int, end: int):
    stamps, poses = _poses_args(stamps, poses)
    out_s = np.zeros(COLS, dtype=np.uint64)
    out_p = np.zeros((COLS, 12), dtype=np.float64)
    _check(_lib().cc_kitti_firing_stamps_and_poses(stamps.shape[0], stamps.ctypes.data, poses.ctypes.data, int(start), int(end),
                                                   out_s.ctypes.data, out_p.ctypes.data))
    return out_s, out_p


def start_end_stamps(middle):
    middle = np.ascontiguousarray(middle, dtype=np.uint64)
    start = np.zeros_like(middle)
    end = np.zeros_like(middle)
    _check(_lib().cc_kitti_start_end_stamps(middle.shape[0], middle.ctypes.data, start.ctypes.data, end.ctypes.data))
    return start, end


def pose_from_line(row12, cam0_from_x) -> np.ndarray:
    row = np.ascontiguousarray(row12, dtype=np.float64).reshape(12)
    cam = np.ascontiguousarray(cam0_from_x, dtype=np.float64).reshape(12)
    out = np.zeros(12, dtype=np.float64)
    _check(_lib().cc_kitti_pose_from_line(row.ctypes.data, cam.ctypes.data, out.ctypes.data))
    return out


# ---- device conversion ------------------------------------------------------------------------------------------------------

class KittiConverter:
    """One cc_kitti handle: up to `max_frames` frames per call, each in its own device slot."""

    def __init__(self, max_frames: int = 1, max_points: int = 140000, device: int = 0, hip_stream: int | None = None):
        self.L = _lib()
        self.h = C.c_void_p()
        _check(self.L.cc_kitti_create(C.byref(self.h), device, max_frames, max_points, hip_stream))
        self._keep = []

    def close(self):
        if self.h:
            self.L.cc_kitti_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def convert(self, frames):
        """frames: list of dicts with keys points (n x 4 f32), stages, and optionally laser_index, start, end, bins (num_bins x 12),
        d_xyz, d_intensity, d_original_index (device pointers as ints)."""
        arr = (Frame * len(frames))()
        self._keep = []
        for f, d in zip(arr, frames):
            pts = np.ascontiguousarray(d["points"], dtype=np.float32).reshape(-1, 4)
            self._keep.append(pts)
            f.points = pts.ctypes.data if pts.shape[0] else None
            f.n_points = pts.shape[0]
            f.stages = int(d.get("stages", ALL_STAGES))
            if d.get("laser_index") is not None:
                li = np.ascontiguousarray(d["laser_index"], dtype=np.uint8)
                assert li.shape[0] == pts.shape[0]
                self._keep.append(li)
                f.laser_index = li.ctypes.data
            if d.get("bins") is not None:
                b = np.ascontiguousarray(d["bins"], dtype=np.float64).reshape(-1, 12)
                self._keep.append(b)
                f.bin_transforms = b.ctypes.data
                f.num_bins = b.shape[0]
            f.rotation_start_stamp = int(d.get("start", 0))
            f.rotation_end_stamp = int(d.get("end", 0))
            f.d_xyz = d.get("d_xyz")
            f.d_intensity = d.get("d_intensity")
            f.d_original_index = d.get("d_original_index")
        _check(self.L.cc_kitti_convert_frames(self.h, len(frames), arr))

    def sync(self):
        _check(self.L.cc_kitti_sync(self.h))

    def hip_stream(self) -> int:
        return self.L.cc_kitti_hip_stream(self.h)

    def result(self, slot: int, n_points: int, points: bool = True, laser: bool = True, cells: bool = True) -> dict:
        info = FrameInfo()
        out = {}
        p = np.zeros((n_points, 4), dtype=np.float32) if points else None
        l = np.zeros(n_points, dtype=np.uint8) if laser else None
        c = np.zeros((ROWS, COLS), dtype=np.int32) if cells else None
        _check(self.L.cc_kitti_frame_result(self.h, slot, C.byref(info), p.ctypes.data if points and n_points else None,
                                            l.ctypes.data if laser and n_points else None, c.ctypes.data if cells else None))
        out.update(points=p, laser_index=l, cell_source=c, rows_found=info.rows_found, max_columns=info.max_columns,
                   break_index=info.break_index, skipped=info.skipped)
        return out


# ---- synthetic KITTI-format data (seeded; stands in for the dataset that cannot be downloaded here) --------------------------

def synthetic_frame(seed: int = 0, n_rows: int = 64, cols_per_row: int = 2083, dropout: float = 0.12, motion=(8.0, 0.0, 0.0, 0.05),
                    duplicate: float = 0.02):
    """One velodyne frame in KITTI .bin order: rows top to bottom, inside a row by azimuth 0 -> pi -> -pi -> 0 (kitti_loader.cpp:50-54),
    NaN returns omitted, ego-motion-corrected to the middle of the rotation. Returns (points n x 4 f32, true_row n u8).
    motion = (vx, vy, vz [m/s], yaw rate [rad/s]) during the 0.1-s rotation; `duplicate` = fraction of returns whose azimuth is
    jittered by up to 1.5 columns (cell collisions for the shift rule)."""
    rng = np.random.default_rng(seed)
    incl = np.deg2rad(np.concatenate([np.linspace(2.0, -8.33, 32), np.linspace(-8.87, -24.8, 32)]))[:n_rows]
    pts, rows = [], []
    for r in range(n_rows):
        k = np.arange(cols_per_row)
        # firing k of the rotation has sensor azimuth pi - (k + 0.5) * 2 pi / cols (clockwise, starting at the -x axis)
        az = np.pi - (k + 0.5) * (2 * np.pi / cols_per_row)
        az = az + rng.uniform(-1.5, 1.5, k.shape) * (2 * np.pi / COLS) * (rng.random(k.shape) < duplicate)
        t = k / cols_per_row - 0.5  # fraction of the rotation relative to its middle
        keep = rng.random(k.shape) >= dropout
        if r < 6:  # the top rows look over the horizon: sparse (kitti_loader.cpp:69-71)
            keep &= rng.random(k.shape) < 0.35
        # ranges: ground plane below the sensor, walls elsewhere
        ground = 1.73 / np.maximum(np.sin(-incl[r]), 1e-3)
        wall = 18.0 + 10.0 * np.sin(3 * az) + 4.0 * np.cos(7 * az + r * 0.1)
        rng_m = np.minimum(ground, wall) * (1 + rng.uniform(-0.002, 0.002, k.shape))
        keep &= rng_m < 110.0
        x = rng_m * np.cos(incl[r]) * np.cos(az)
        y = rng_m * np.cos(incl[r]) * np.sin(az)
        z = rng_m * np.sin(incl[r])
        # ego-motion correction to the middle of the rotation: the sensor was at pose(t) when it measured the point
        yaw = motion[3] * 0.1 * t
        cx, sx = np.cos(yaw), np.sin(yaw)
        xc = cx * x - sx * y + motion[0] * 0.1 * t
        yc = sx * x + cx * y + motion[1] * 0.1 * t
        zc = z + motion[2] * 0.1 * t
        inten = rng.random(k.shape)
        p = np.stack([xc, yc, zc, inten], axis=1)[keep]
        # file order inside a row: azimuth of the corrected point, 0 -> pi -> -pi -> 0
        a = np.arctan2(p[:, 1], p[:, 0])
        order = np.argsort(np.where(a < 0, a + 2 * np.pi, a), kind="stable")
        pts.append(p[order])
        rows.append(np.full(order.shape[0], r, dtype=np.uint8))
    return np.concatenate(pts).astype(np.float32), np.concatenate(rows)


def synthetic_poses(n_frames: int, motion=(8.0, 0.0, 0.0, 0.05), dt: float = 0.1):
    """poses.txt rows (first_cam0_from_cam0, 12 numbers per frame) and times.txt seconds for a constant-twist drive. The velodyne
    moves along its +x = cam0's +z; yaw about velodyne z = cam0's -y."""
    rows, times = [], []
    for f in range(n_frames):
        yaw = motion[3] * dt * f
        # velodyne-frame pose of frame f relative to frame 0
        c, s = np.cos(yaw), np.sin(yaw)
        Rv = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        tv = np.array([motion[0] * dt * f, motion[1] * dt * f, motion[2] * dt * f])
        # cam0 axes: x_c = -y_v, y_c = -z_v, z_c = x_v
        P = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0.0]])
        Rc = P @ Rv @ P.T
        tc = P @ tv
        rows.append(np.concatenate([np.concatenate([Rc[i], tc[i:i + 1]]) for i in range(3)]))
        times.append(f * dt)
    return np.array(rows), np.array(times)


# Tr of calib.txt that makes cam0_from_velodyne the pure axis permutation used above
CALIB_TR = np.array([0, -1, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0], dtype=np.float64)


def write_synthetic_sequence(root: str, sequence: int, n_frames: int, seed: int = 0, motion=(8.0, 0.0, 0.0, 0.05), labels: bool = True,
                             cols_per_row: int = 2083):
    """Lay out <root>/sequences/<ss>/{velodyne/*.bin, labels/*.label, labels_euclidean_clustering/*.label, poses.txt, times.txt,
    calib.txt} the way kitti_demo.cpp:243-262 expects them. Labels are synthetic: ground = road (40), the rest building (50) with
    instance = euclidean label = 1 + 8-sector index of the azimuth."""
    seq = os.path.join(root, "sequences", f"{sequence:02d}")
    os.makedirs(os.path.join(seq, "velodyne"), exist_ok=True)
    if labels:
        os.makedirs(os.path.join(seq, "labels"), exist_ok=True)
        os.makedirs(os.path.join(seq, "labels_euclidean_clustering"), exist_ok=True)
    rows, times = synthetic_poses(n_frames, motion)
    with open(os.path.join(seq, "poses.txt"), "w") as f:
        for r in rows:
            f.write(" ".join(repr(float(v)) for v in r) + "\n")
    with open(os.path.join(seq, "times.txt"), "w") as f:
        for t in times:
            f.write(f"{t:.6e}\n")
    with open(os.path.join(seq, "calib.txt"), "w") as f:
        for name in ("P0", "P1", "P2", "P3"):
            f.write(name + ": " + " ".join(repr(float(v)) for v in [700, 0, 600, 0, 0, 700, 180, 0, 0, 0, 1, 0]) + "\n")
        f.write("Tr: " + " ".join(repr(float(v)) for v in CALIB_TR) + "\n")
    sizes = []
    for fi in range(n_frames):
        pts, _ = synthetic_frame(seed + fi, motion=motion, cols_per_row=cols_per_row)
        pts.tofile(os.path.join(seq, "velodyne", f"{fi:06d}.bin"))
        sizes.append(pts.shape[0])
        if labels:
            is_ground = pts[:, 2] < -1.55
            sector = ((np.arctan2(pts[:, 1], pts[:, 0]) + np.pi) / (2 * np.pi) * 8).astype(np.int64).clip(0, 7)
            sem = np.where(is_ground, 40, 50).astype(np.uint16)
            inst = np.where(is_ground, 0, 1 + sector).astype(np.uint16)
            np.stack([sem, inst], axis=1).astype(np.uint16).tofile(os.path.join(seq, "labels", f"{fi:06d}.label"))
            eu = np.where(is_ground, 0, 1 + sector).astype(np.uint16)
            eu.tofile(os.path.join(seq, "labels_euclidean_clustering", f"{fi:06d}.label"))
    return seq, sizes
