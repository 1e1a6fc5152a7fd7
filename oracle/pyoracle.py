"""ctypes wrapper of oracle/libcc_oracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module. It is the checker
for the HIP path, never part of it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from continuous_clustering_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "cc_oracle.cpp"), os.path.join(_HERE, "eval_oracle.cpp"), os.path.join(_HERE, "kitti_oracle.cpp"), os.path.join(_HERE, "gt_oracle.cpp"),
            os.path.join(_HERE, "..", "include", "cc_hip.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "libcc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(capi.Config), C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_record.argtypes = [C.c_void_p, C.c_int]
        L.orc_keep_published_tail.argtypes = [C.c_void_p, C.c_int64]
        L.orc_keep_published_tail.restype = None
        L.orc_set_config.argtypes = [C.c_void_p, C.POINTER(capi.Config)]
        L.orc_reset.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_robot_from_sensor.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_add_firings.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_time_firings.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_time_firings.restype = C.c_double
        L.orc_time_each_firing.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_time_each_firing.restype = C.c_double
        L.orc_time_firings_pipeline.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_time_firings_pipeline.restype = C.c_double
        L.orc_add_firings_pipeline.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_stream_state.argtypes = [C.c_void_p, C.POINTER(capi.StreamState)]
        L.orc_num_events.restype = C.c_int64
        L.orc_num_events.argtypes = [C.c_void_p]
        L.orc_drain_events.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_read_published.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(capi.ColumnView)]
        L.orc_cluster_members.restype = C.c_int64
        L.orc_cluster_members.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_published_base.restype = C.c_int64
        L.orc_published_base.argtypes = [C.c_void_p]
        L.orc_published_count.restype = C.c_int64
        L.orc_published_count.argtypes = [C.c_void_p]
        L.orc_eval_frame.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_eval_mean_std.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_eval_mean_std.restype = None
        vp, i64, u64 = C.c_void_p, C.c_int64, C.c_uint64
        L.korc_recover_laser_indices.argtypes = [i64, vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.korc_bin_transforms.argtypes = [i64, vp, vp, u64, u64, vp, vp, C.c_int32]
        L.korc_bin_transforms.restype = C.c_int32
        L.korc_undo_ego_motion.argtypes = [i64, vp, u64, u64, vp, i64, vp, vp]
        L.korc_undo_ego_motion.restype = None
        L.korc_generate_range_image.argtypes = [i64, vp, vp, C.c_int, vp]
        L.korc_generate_range_image.restype = i64
        L.korc_make_firings.argtypes = [i64, vp, vp, u64, u64, C.c_int, C.c_int, vp, vp, vp, vp]
        L.korc_make_firings.restype = None
        L.korc_interpolate.argtypes = [i64, vp, vp, u64, vp]
        L.korc_interpolate.restype = None
        L.korc_start_end_stamps.argtypes = [i64, vp, vp, vp]
        L.korc_start_end_stamps.restype = None
        L.korc_pose_from_line.argtypes = [vp, vp, vp]
        L.korc_pose_from_line.restype = None
        L.orc_generate_euclidean_labels.argtypes = [i64, vp, vp, vp, vp, C.POINTER(C.c_int32)]
        L.orc_generate_euclidean_labels.restype = None
        _lib = L
    return _lib


IDENTITY_TF = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64)


class Oracle:
    """One sensor stream processed by the CPU restatement (single-threaded reference order)."""

    def __init__(self, cfg: capi.Config, num_rows: int, robot_from_sensor=IDENTITY_TF, record: bool = True):
        self.L = lib()
        self.cfg = cfg.copy()
        self.num_rows = num_rows
        self.h = self.L.orc_create(C.byref(self.cfg), num_rows)
        if robot_from_sensor is not None:
            self.set_robot_from_sensor(robot_from_sensor)
        self.L.orc_record(self.h, 1 if record else 0)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def keep_published_tail(self, n: int):
        """Keep only (at least) the last ``n`` published column snapshots (long runs: bench.py's verify block)."""
        self.L.orc_keep_published_tail(self.h, int(n))

    def set_robot_from_sensor(self, tf12):
        tf = np.ascontiguousarray(tf12, dtype=np.float64).reshape(12)
        self.L.orc_set_robot_from_sensor(self.h, tf.ctypes.data)

    def reset(self, num_rows=None):
        self.L.orc_reset(self.h, self.num_rows if num_rows is None else num_rows)

    def set_config(self, cfg: capi.Config):
        self.cfg = cfg.copy()
        self.L.orc_set_config(self.h, C.byref(self.cfg))

    def add_firings(self, xyz, intensity, poses) -> int:
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        intensity = np.ascontiguousarray(intensity, dtype=np.uint8)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        n = xyz.shape[0]
        assert xyz.shape == (n, self.num_rows, 3) and intensity.shape == (n, self.num_rows) and poses.shape == (n, 12)
        return self.L.orc_add_firings(self.h, n, xyz.ctypes.data, intensity.ctypes.data, poses.ctypes.data)

    def time_firings(self, xyz, intensity, poses) -> float:
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        intensity = np.ascontiguousarray(intensity, dtype=np.uint8)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        return self.L.orc_time_firings(self.h, xyz.shape[0], xyz.ctypes.data, intensity.ctypes.data, poses.ctypes.data)

    def time_firings_pipeline(self, xyz, intensity, poses) -> float:
        """BASELINE.md mode B: the firings through the three-thread stage pipeline (cc_oracle.cpp: Oracle::Pipe); seconds, -1 on error."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        intensity = np.ascontiguousarray(intensity, dtype=np.uint8)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        return self.L.orc_time_firings_pipeline(self.h, xyz.shape[0], xyz.ctypes.data, intensity.ctypes.data, poses.ctypes.data)

    def add_firings_pipeline(self, xyz, intensity, poses) -> int:
        """The same pipeline with recording on (tests: equal to add_firings)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        intensity = np.ascontiguousarray(intensity, dtype=np.uint8)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        return self.L.orc_add_firings_pipeline(self.h, xyz.shape[0], xyz.ctypes.data, intensity.ctypes.data, poses.ctypes.data)

    def time_each_firing(self, xyz, intensity, poses) -> np.ndarray:
        """Per-call addFiring latency in nanoseconds (BASELINE.md mode A)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        intensity = np.ascontiguousarray(intensity, dtype=np.uint8)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        out = np.zeros(xyz.shape[0], dtype=np.float64)
        if self.L.orc_time_each_firing(self.h, xyz.shape[0], xyz.ctypes.data, intensity.ctypes.data, poses.ctypes.data, out.ctypes.data) < 0:
            raise RuntimeError(self.last_error())
        return out

    def last_error(self) -> str:
        return self.L.orc_last_error(self.h).decode()

    def state(self) -> dict:
        s = capi.StreamState()
        self.L.orc_stream_state(self.h, C.byref(s))
        return capi.state_to_dict(s)

    def drain_events(self) -> np.ndarray:
        n = self.L.orc_num_events(self.h)
        out = np.zeros(n, dtype=capi.EVENT_DTYPE)
        got = C.c_int64(0)
        self.L.orc_drain_events(self.h, out.ctypes.data, n, C.byref(got))
        return out[: got.value]

    def cluster_members(self, idx: int):
        """(global column, row) of the idx-th id-carrying cluster since reset, in the reference's cluster_points order."""
        n = self.L.orc_cluster_members(self.h, idx, 0, None, None)
        if n < 0:
            raise IndexError(idx)
        g, r = np.zeros(max(1, n), dtype=np.int64), np.zeros(max(1, n), dtype=np.int32)
        self.L.orc_cluster_members(self.h, idx, n, g.ctypes.data, r.ctypes.data)
        return g[:n], r[:n]

    def published_range(self):
        base = self.L.orc_published_base(self.h)
        cnt = self.L.orc_published_count(self.h)
        return base, base + cnt - 1

    def read_published(self, frm: int, to: int, fields=None) -> dict:
        v, arrays = capi.make_column_view(to - frm + 1, self.num_rows, fields)
        rc = self.L.orc_read_published(self.h, frm, to, C.byref(v))
        if rc != 0:
            raise ValueError(f"orc_read_published({frm},{to}) -> {rc}")
        return arrays


def eval_frame(semantic, euclid, is_ground, detection) -> np.ndarray:
    """Reference-order label compare of one frame -> [tp, fn, fp, tn, OSE, USE]."""
    semantic = np.ascontiguousarray(semantic, dtype=np.uint16)
    euclid = np.ascontiguousarray(euclid, dtype=np.uint32)
    is_ground = np.ascontiguousarray(is_ground, dtype=np.uint8)
    detection = np.ascontiguousarray(detection, dtype=np.uint32)
    out = np.zeros(6, dtype=np.float64)
    lib().orc_eval_frame(semantic.shape[0], semantic.ctypes.data, euclid.ctypes.data, is_ground.ctypes.data, detection.ctypes.data,
                         out.ctypes.data)
    return out


def mean_std(data):
    data = np.ascontiguousarray(data, dtype=np.float64)
    m, s = C.c_double(0), C.c_double(0)
    lib().orc_eval_mean_std(data.ctypes.data, data.shape[0], C.byref(m), C.byref(s))
    return m.value, s.value


# ---- KITTI replay path (oracle/kitti_oracle.cpp) -----------------------------------------------------------------------------

KITTI_ROWS, KITTI_COLS = 64, 2200


def _poses(stamps, poses):
    stamps = np.ascontiguousarray(stamps, dtype=np.uint64)
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 12)
    return stamps, poses


def kitti_recover_laser_indices(points):
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
    laser = np.zeros(pts.shape[0], dtype=np.uint8)
    rows, maxc = C.c_int32(0), C.c_int32(0)
    threw = lib().korc_recover_laser_indices(pts.shape[0], pts.ctypes.data, laser.ctypes.data, C.byref(rows), C.byref(maxc))
    return laser, rows.value, maxc.value, bool(threw)


def kitti_bin_transforms(stamps, poses, start, end, mid_pose):
    stamps, poses = _poses(stamps, poses)
    mid = np.ascontiguousarray(mid_pose, dtype=np.float64).reshape(12)
    out = np.zeros((512, 12), dtype=np.float64)
    nb = lib().korc_bin_transforms(stamps.shape[0], stamps.ctypes.data, poses.ctypes.data, int(start), int(end), mid.ctypes.data,
                                   out.ctypes.data, 512)
    return out[:nb].copy()


def kitti_undo_ego_motion(points, start, end, mid_pose, stamps, poses):
    pts = np.array(points, dtype=np.float32).reshape(-1, 4)
    stamps, poses = _poses(stamps, poses)
    mid = np.ascontiguousarray(mid_pose, dtype=np.float64).reshape(12)
    lib().korc_undo_ego_motion(pts.shape[0], pts.ctypes.data, int(start), int(end), mid.ctypes.data, stamps.shape[0], stamps.ctypes.data,
                               poses.ctypes.data)
    return pts


def kitti_generate_range_image(points, laser, shift=True):
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
    laser = np.ascontiguousarray(laser, dtype=np.uint8)
    cells = np.zeros((KITTI_ROWS, KITTI_COLS), dtype=np.int32)
    skipped = lib().korc_generate_range_image(pts.shape[0], pts.ctypes.data, laser.ctypes.data, 1 if shift else 0, cells.ctypes.data)
    return cells, skipped


def kitti_make_firings(points, cells, start, end, sequence=0, frame=0):
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    xyz = np.zeros((KITTI_COLS, KITTI_ROWS, 3), dtype=np.float32)
    inten = np.zeros((KITTI_COLS, KITTI_ROWS), dtype=np.uint8)
    unique = np.zeros((KITTI_COLS, KITTI_ROWS), dtype=np.uint64)
    stamps = np.zeros(KITTI_COLS, dtype=np.uint64)
    lib().korc_make_firings(pts.shape[0], pts.ctypes.data, cells.ctypes.data, int(start), int(end), sequence, frame, xyz.ctypes.data,
                            inten.ctypes.data, unique.ctypes.data, stamps.ctypes.data)
    return xyz, inten, unique, stamps


def kitti_interpolate(stamps, poses, stamp):
    stamps, poses = _poses(stamps, poses)
    out = np.zeros(12, dtype=np.float64)
    lib().korc_interpolate(stamps.shape[0], stamps.ctypes.data, poses.ctypes.data, int(stamp), out.ctypes.data)
    return out


def kitti_start_end_stamps(middle):
    middle = np.ascontiguousarray(middle, dtype=np.uint64)
    start, end = np.zeros_like(middle), np.zeros_like(middle)
    lib().korc_start_end_stamps(middle.shape[0], middle.ctypes.data, start.ctypes.data, end.ctypes.data)
    return start, end


def kitti_pose_from_line(row12, cam0_from_x):
    row = np.ascontiguousarray(row12, dtype=np.float64).reshape(12)
    cam = np.ascontiguousarray(cam0_from_x, dtype=np.float64).reshape(12)
    out = np.zeros(12, dtype=np.float64)
    lib().korc_pose_from_line(row.ctypes.data, cam.ctypes.data, out.ctypes.data)
    return out


def generate_euclidean_labels(points, semantic, instance):
    """generateEuclideanClusteringLabels (oracle/gt_oracle.cpp) -> (labels u16 [n], number of kept clusters)."""
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
    sem = np.ascontiguousarray(semantic, dtype=np.uint16)
    inst = np.ascontiguousarray(instance, dtype=np.uint16)
    out = np.zeros(pts.shape[0], dtype=np.uint16)
    nc = C.c_int32(0)
    lib().orc_generate_euclidean_labels(pts.shape[0], pts.ctypes.data, sem.ctypes.data, inst.ctypes.data, out.ctypes.data, C.byref(nc))
    return out, nc.value
