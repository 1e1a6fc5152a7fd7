"""continuous_clustering_amd — MI355X (gfx950) implementation of the per-column hot path of
UniBwTAS/continuous_clustering behind a C-ABI (include/cc_hip.h).

The product is ``libcc_hip.so`` (hand-written HIP kernels + C-ABI host code, built in-tree by
``continuous_clustering_amd.build``). This Python module is a thin ctypes host mirror of that ABI used by the
tests and bench.py; the C++ drop-in class lives in ``csrc/continuous_clustering.hpp``. There is NO CPU
implementation behind this module: loading fails loudly when the library is missing and engine creation
fails with CC_ERR_NO_DEVICE when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from . import capi
from .capi import Config

__all__ = ["Config", "Engine", "EngineError", "load_library", "capi", "IDENTITY_TF"]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("CC_HIP_LIB", "libcc_hip.so"))  # (CC_HIP_LIB: another build of the same library, for A/B tools)
_lib = None

IDENTITY_TF = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64)


class EngineError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"cc_hip error {code}: {msg}")
        self.code = code


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64. If this module
    loaded /opt/rocm's copy first and torch its own afterwards, the second runtime would find no GPU. So when a torch
    wheel is installed, bind to ITS runtime (without importing torch); otherwise the system runtime is used."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def load_library():
    """dlopen libcc_hip.so (never builds implicitly, never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m continuous_clustering_amd.build` "
                          f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.cc_config_default.argtypes = [C.POINTER(Config)]
    L.cc_config_default.restype = None
    L.cc_config_kitti.argtypes = [C.POINTER(Config)]
    L.cc_config_kitti.restype = None
    L.cc_version.restype = C.c_char_p
    L.cc_engine_create.argtypes = [C.POINTER(vp), i32, i32, i32, C.POINTER(Config)]
    L.cc_engine_destroy.argtypes = [vp]
    L.cc_engine_destroy.restype = None
    L.cc_engine_set_config.argtypes = [vp, C.POINTER(Config)]
    L.cc_engine_reset.argtypes = [vp, i32]
    L.cc_engine_set_robot_from_sensor.argtypes = [vp, i32, vp]
    L.cc_engine_add_firings.argtypes = [vp, i32, i64, vp, vp, vp]
    L.cc_engine_add_firings_device.argtypes = [vp, i64, vp, vp, vp]
    L.cc_engine_sync.argtypes = [vp]
    L.cc_engine_inputs_released.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.cc_engine_hip_stream.argtypes = [vp]
    L.cc_engine_hip_stream.restype = vp
    L.cc_engine_record_events.argtypes = [vp, i32]
    L.cc_engine_drain_events.argtypes = [vp, i32, vp, i64, C.POINTER(i64)]
    L.cc_engine_pending_events.argtypes = [vp, i32, C.POINTER(i64)]
    L.cc_engine_drain_links.argtypes = [vp, i32, vp, i64, C.POINTER(i64)]
    L.cc_engine_stream_state.argtypes = [vp, i32, C.POINTER(capi.StreamState)]
    L.cc_engine_read_columns.argtypes = [vp, i32, i64, i64, C.POINTER(capi.ColumnView)]
    L.cc_engine_read_column_ranges.argtypes = [vp, i32, i32, vp, vp, C.POINTER(capi.ColumnView)]
    L.cc_engine_output_planes.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp)]
    L.cc_engine_set_option.argtypes = [vp, C.c_char_p, i64]
    L.cc_engine_gather_cluster_points.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, vp]
    L.cc_engine_enable_timing.argtypes = [vp, i32]
    L.cc_engine_kernel_times.argtypes = [vp, C.POINTER(C.c_double * 7), C.POINTER(C.c_uint64)]
    L.cc_engine_totals.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 4
    L.cc_engine_batch_counters.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 2 + [C.POINTER(C.c_uint64 * 8)]
    L.cc_engine_gate_counters.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 2
    L.cc_engine_resident_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.cc_engine_view_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.cc_engine_last_error.argtypes = [vp]
    L.cc_engine_last_error.restype = C.c_char_p
    _lib = L
    return L


def _ptr(a):
    """Device/host pointer of a numpy array, a torch tensor or a raw integer address."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()  # torch tensor


class Engine:
    """``num_streams`` independent sensor streams on one GPU — the Python face of ``cc_engine``.

    Method names mirror the reference class (``ContinuousClustering::reset / setConfiguration / addFiring /
    setTransformRobotFrameFromSensorFrame / resetRequired``, continuous_clustering.hpp:200-221); firings are
    passed in batches because one kernel launch advances all streams by a batch.
    """

    def __init__(self, cfg: Config, num_rows: int, num_streams: int = 1, device: int = 0, robot_from_sensor=IDENTITY_TF):
        self.L = load_library()
        self.cfg = cfg.copy()
        self.num_rows = num_rows
        self.num_streams = num_streams
        self.h = C.c_void_p()
        rc = self.L.cc_engine_create(C.byref(self.h), device, num_streams, num_rows, C.byref(self.cfg))
        if rc != capi.CC_OK:
            self.h = None
            raise EngineError(rc, "cc_engine_create failed" + (" (no MI355X visible)" if rc == capi.CC_ERR_NO_DEVICE else ""))
        if robot_from_sensor is not None:
            self.set_robot_from_sensor(robot_from_sensor)

    def close(self):
        if getattr(self, "h", None):
            self.L.cc_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != capi.CC_OK:
            raise EngineError(rc, self.L.cc_engine_last_error(self.h).decode())

    # ---- configuration / lifecycle --------------------------------------------------------------
    def set_config(self, cfg: Config):
        self.cfg = cfg.copy()
        self._check(self.L.cc_engine_set_config(self.h, C.byref(self.cfg)))

    def reset(self, num_rows: int | None = None):
        if num_rows is not None:
            self.num_rows = num_rows
        self._check(self.L.cc_engine_reset(self.h, self.num_rows))

    def set_robot_from_sensor(self, tf12, stream: int = -1):
        tf = np.ascontiguousarray(tf12, dtype=np.float64).reshape(12)
        self._check(self.L.cc_engine_set_robot_from_sensor(self.h, stream, tf.ctypes.data))

    def record_events(self, enable: bool):
        self._check(self.L.cc_engine_record_events(self.h, 1 if enable else 0))

    # ---- data path -------------------------------------------------------------------------------
    def add_firings(self, xyz, intensity, poses, stream: int = 0) -> int:
        """Host buffers, one stream. Returns the status code (CC_OK or the reference's error class)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        intensity = np.ascontiguousarray(intensity, dtype=np.uint8)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        n = xyz.shape[0]
        if xyz.shape != (n, self.num_rows, 3) or intensity.shape != (n, self.num_rows) or poses.shape != (n, 12):
            raise ValueError("firing arrays must be [n, num_rows, 3], [n, num_rows], [n, 12]")
        return self.L.cc_engine_add_firings(self.h, stream, n, xyz.ctypes.data, intensity.ctypes.data, poses.ctypes.data)

    def add_firings_device(self, n: int, d_xyz, d_intensity, d_poses):
        """Device-resident buffers laid out [num_streams][n][...]; asynchronous — and so are the engine's reads: the three buffers must stay
        allocated and unchanged until ``inputs_released()`` reports this call (calls are numbered 1, 2, ...) or ``sync()`` has returned.
        A caller that streams batches in keeps its buffers in a ring and asks before it overwrites one (include/cc_hip.h)."""
        self._check(self.L.cc_engine_add_firings_device(self.h, n, _ptr(d_xyz), _ptr(d_intensity), _ptr(d_poses)))

    def inputs_released(self) -> tuple:
        """(last call of add_firings_device whose input buffers will not be read again, calls made so far); never waits."""
        rel, sub = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.cc_engine_inputs_released(self.h, C.byref(rel), C.byref(sub)))
        return int(rel.value), int(sub.value)

    def sync(self) -> int:
        return self.L.cc_engine_sync(self.h)

    def hip_stream(self) -> int:
        return self.L.cc_engine_hip_stream(self.h)

    def last_error(self) -> str:
        return self.L.cc_engine_last_error(self.h).decode()

    # ---- results ---------------------------------------------------------------------------------
    def state(self, stream: int = 0) -> dict:
        s = capi.StreamState()
        self._check(self.L.cc_engine_stream_state(self.h, stream, C.byref(s)))
        return capi.state_to_dict(s)

    def drain_events(self, stream: int = 0) -> np.ndarray:
        n = C.c_int64(0)
        self._check(self.L.cc_engine_pending_events(self.h, stream, C.byref(n)))
        out = np.zeros(max(1, n.value), dtype=capi.EVENT_DTYPE)
        got = C.c_int64(0)
        self._check(self.L.cc_engine_drain_events(self.h, stream, out.ctypes.data, n.value, C.byref(got)))
        return out[: got.value]

    def drain_links(self, stream: int = 0) -> np.ndarray:
        """Tree links made since the last drain: rows of (root gcol, root row, root gcol, root row)."""
        n = C.c_int64(0)
        self._check(self.L.cc_engine_drain_links(self.h, stream, None, 0, C.byref(n)))
        out = np.zeros((max(1, n.value), 4), dtype=np.int64)
        got = C.c_int64(0)
        self._check(self.L.cc_engine_drain_links(self.h, stream, out.ctypes.data, n.value, C.byref(got)))
        return out[: got.value]

    def read_columns(self, frm: int, to: int, stream: int = 0, fields=None) -> dict:
        v, arrays = capi.make_column_view(to - frm + 1, self.num_rows, fields)
        self._check(self.L.cc_engine_read_columns(self.h, stream, frm, to, C.byref(v)))
        return arrays

    def read_column_ranges(self, ranges, stream: int = 0, fields=None) -> dict:
        """Columns of up to 8 ranges [(from, to), ...] in one call; the arrays hold the ranges' columns one after the other."""
        frm = np.ascontiguousarray([r[0] for r in ranges], dtype=np.int64)
        to = np.ascontiguousarray([r[1] for r in ranges], dtype=np.int64)
        v, arrays = capi.make_column_view(int((to - frm + 1).sum()), self.num_rows, fields)
        self._check(self.L.cc_engine_read_column_ranges(self.h, stream, len(ranges), frm.ctypes.data, to.ctypes.data, C.byref(v)))
        return arrays

    def gather_cluster_points(self, cluster_events: np.ndarray, stream: int = 0):
        """Member points of finished clusters (CC_EV_CLUSTER events drained from this stream), gathered on the device:
        returns (offsets[n + 1], global_column[total], row[total]); cluster i owns [offsets[i], offsets[i + 1])."""
        ev = cluster_events[cluster_events["type"] == capi.EV_CLUSTER]
        n = len(ev)
        cid = np.ascontiguousarray(ev["c"], dtype=np.uint32)
        a = np.ascontiguousarray(ev["a"], dtype=np.int64)
        b = np.ascontiguousarray(ev["b"], dtype=np.int64)
        cnt = np.ascontiguousarray(ev["d"], dtype=np.uint32)
        offsets = np.zeros(n + 1, dtype=np.int64)
        offsets[1:] = np.cumsum(cnt.astype(np.int64))
        gcol = np.zeros(max(1, int(offsets[-1])), dtype=np.int64)
        row = np.zeros(max(1, int(offsets[-1])), dtype=np.int32)
        self._check(self.L.cc_engine_gather_cluster_points(self.h, stream, n, cid.ctypes.data, a.ctypes.data, b.ctypes.data,
                                                           cnt.ctypes.data, gcol.ctypes.data, row.ctypes.data))
        return offsets, gcol[: offsets[-1]], row[: offsets[-1]]

    def set_option(self, name: str, value: int):
        self._check(self.L.cc_engine_set_option(self.h, name.encode(), int(value)))

    def enable_timing(self, enable: bool = True):
        self._check(self.L.cc_engine_enable_timing(self.h, 1 if enable else 0))

    def kernel_times(self) -> dict:
        ms = (C.c_double * 7)()
        n = C.c_uint64(0)
        self._check(self.L.cc_engine_kernel_times(self.h, C.byref(ms), C.byref(n)))
        names = ("prep_ms", "insert_ms", "segment_ms", "scan_ms", "assoc_lds_ms", "assoc_global_ms", "publish_ms")
        d = {k: ms[i] for i, k in enumerate(names)}
        d["batches"] = n.value
        return d

    def totals(self) -> dict:
        v = [C.c_uint64(0) for _ in range(4)]
        self._check(self.L.cc_engine_totals(self.h, *[C.byref(x) for x in v]))
        return {"cells_published": v[0].value, "clusters_finished": v[1].value, "firings_consumed": v[2].value,
                "serial_columns": v[3].value}

    def batch_counters(self) -> dict:
        v = [C.c_uint64(0) for _ in range(2)]
        why = (C.c_uint64 * 8)()
        self._check(self.L.cc_engine_batch_counters(self.h, *[C.byref(x) for x in v], C.byref(why)))
        return {"batch_columns": v[0].value, "batch_bails": v[1].value, "bail_reasons": list(why)}

    def resident_counters(self) -> dict:
        """Option "resident": launches of the resident kernel, calls it has answered (as of its last exit), whether it is running now."""
        a, b, r = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        self._check(self.L.cc_engine_resident_counters(self.h, C.byref(a), C.byref(b), C.byref(r)))
        return {"launches": int(a.value), "calls": int(b.value), "running": bool(r.value)}

    def view_counters(self) -> dict:
        """read_columns calls served from the views a small call mirrored with its results / by the view kernel."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.cc_engine_view_counters(self.h, C.byref(a), C.byref(b)))
        return {"mirror": int(a.value), "kernel": int(b.value)}

    def gate_counters(self) -> dict:
        v = [C.c_uint64(0) for _ in range(2)]
        self._check(self.L.cc_engine_gate_counters(self.h, *[C.byref(x) for x in v]))
        return {"lazy_batches": v[0].value, "lazy_redone": v[1].value}

    def output_planes(self, stream: int = 0):
        g, i = C.c_void_p(), C.c_void_p()
        self._check(self.L.cc_engine_output_planes(self.h, stream, C.byref(g), C.byref(i)))
        return g.value, i.value
