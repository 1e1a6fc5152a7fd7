"""Host mirror of the evaluation path around the label compare (src/evaluation/kitti_evaluation.cpp, the frame scatter of
src/tools/kitti_demo.cpp:173-224) plus the one collective of the whole system: gathering per-frame evaluation records from the
ranks that own the sensor streams (torch.distributed: RCCL over xGMI on MI355X, gloo in the CPU tests).

The label compare itself (confusion counts, contingency table) runs on the GPU behind ``cc_eval_frame``; nothing here computes it
on the CPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import load_library

RESULT_FIELDS = ("tp", "fn", "fp", "tn", "over_segmentation_entropy", "under_segmentation_entropy")


def _lib():
    L = load_library()
    if not getattr(L, "_eval_ready", False):
        L.cc_eval_frame.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cc_eval_frame_device.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cc_eval_mean_std.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.cc_eval_mean_std.restype = None
        L.cc_eval_summarize.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.cc_eval_summarize.restype = None
        L.cc_eval_generate_euclidean_labels.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L._eval_ready = True
    return L


def generate_euclidean_labels(points, semantic, instance, device: int = 0) -> np.ndarray:
    """KittiEvaluation::generateEuclideanClusteringLabels on the GPU (kitti_evaluation.cpp:224-275): points n x 4 f32 (the .bin payload),
    semantic / instance u16 (the .label payload) -> u16 ground-truth euclidean-clustering label per point."""
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
    sem = np.ascontiguousarray(semantic, dtype=np.uint16)
    inst = np.ascontiguousarray(instance, dtype=np.uint16)
    assert sem.shape[0] == pts.shape[0] == inst.shape[0]
    out = np.zeros(pts.shape[0], dtype=np.uint16)
    rc = _lib().cc_eval_generate_euclidean_labels(device, pts.shape[0], pts.ctypes.data, sem.ctypes.data, inst.ctypes.data, out.ctypes.data)
    if rc != 0:
        from . import EngineError
        raise EngineError(rc, "cc_eval_generate_euclidean_labels failed")
    return out


def eval_frame(semantic, euclid, is_ground, detection, device: int = 0) -> np.ndarray:
    """KittiEvaluation::evaluate for one frame on the GPU -> [tp, fn, fp, tn, OSE, USE] (kitti_evaluation.hpp:38-49)."""
    semantic = np.ascontiguousarray(semantic, dtype=np.uint16)
    euclid = np.ascontiguousarray(euclid, dtype=np.uint32)
    is_ground = np.ascontiguousarray(is_ground, dtype=np.uint8)
    detection = np.ascontiguousarray(detection, dtype=np.uint32)
    out = np.zeros(6, dtype=np.float64)
    rc = _lib().cc_eval_frame(device, semantic.shape[0], semantic.ctypes.data, euclid.ctypes.data, is_ground.ctypes.data,
                              detection.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"cc_eval_frame failed with {rc}")
    return out


def eval_frame_device(n: int, d_semantic, d_euclid, d_is_ground, d_detection) -> np.ndarray:
    """Same, for arrays already in HBM (torch tensors or raw device pointers)."""
    from . import _ptr
    out = np.zeros(6, dtype=np.float64)
    rc = _lib().cc_eval_frame_device(n, _ptr(d_semantic), _ptr(d_euclid), _ptr(d_is_ground), _ptr(d_detection), out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"cc_eval_frame_device failed with {rc}")
    return out


def summarize(frames: np.ndarray) -> dict:
    """generateEvaluationResults' six (mean, sigma) pairs (kitti_evaluation.cpp:187-208) over per-frame records [k, 6]."""
    frames = np.ascontiguousarray(frames, dtype=np.float64).reshape(-1, 6)
    out = np.zeros(12, dtype=np.float64)
    _lib().cc_eval_summarize(frames.ctypes.data, frames.shape[0], out.ctypes.data)
    names = ("recall", "precision", "f1", "accuracy", "use", "ose")
    return {n: (out[2 * i], out[2 * i + 1]) for i, n in enumerate(names)}


def format_row(name: str, summary: dict) -> str:
    """One line of the reference's Markdown table (kitti_evaluation.cpp:180-208): first four metrics x100, two decimals."""
    cells = []
    for i, k in enumerate(("recall", "precision", "f1", "accuracy", "use", "ose")):
        m, s = summary[k]
        if i < 4:
            m, s = m * 100, s * 100
        cells.append(f"| {m:.2f} / {s:.2f} ")
    return f"| {name} " + "".join(cells) + "|"


class FrameScatter:
    """Maps published range-image cells back to the points of their frames, like addColumnAndEvaluateFrameIfCompleted
    (kitti_demo.cpp:173-224): globally_unique_point_index = sequence << 48 | frame << 32 | point index (kitti_demo.cpp:153-155,
    199-201); frame N is evaluated when the first cell of frame N + 1 is published (kitti_demo.cpp:208-209, 221-222)."""

    def __init__(self, sequence: int, frame_sizes, semantic, euclid, evaluate=eval_frame):
        self.sequence = sequence
        self.semantic = semantic      # list of per-frame uint16 arrays
        self.euclid = euclid          # list of per-frame uint32 arrays
        self.is_ground = [np.zeros(n, np.uint8) for n in frame_sizes]
        self.detection = [np.zeros(n, np.uint32) for n in frame_sizes]
        self.previous_frame = 0
        self.records = []             # (sequence, frame, 6 result values)
        self.evaluate = evaluate

    def add_columns(self, unique_index: np.ndarray, ground_label: np.ndarray, ids: np.ndarray, gp_ground: int = 54):
        """unique_index / ground_label / ids: [columns, rows] of the newly published columns, in column order."""
        for c in range(unique_index.shape[0]):
            u = unique_index[c]
            has = u != np.uint64(2 ** 64 - 1)
            if not has.any():
                continue
            frame = ((u[has] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
            pt = (u[has] & np.uint64(0xFFFFFFFF)).astype(np.int64)
            if (frame < self.previous_frame).any():
                raise RuntimeError("Found a point belonging to a frame that was already evaluated!")
            if (frame > self.previous_frame + 1).any():
                raise RuntimeError("Found a point whose frame is too far ahead!")
            for f in np.unique(frame):
                m = frame == f
                self.is_ground[f][pt[m]] = (ground_label[c][has][m] == gp_ground)
                self.detection[f][pt[m]] = ids[c][has][m].astype(np.uint32)
            if (frame == self.previous_frame + 1).any():
                self._evaluate_previous()

    def _evaluate_previous(self):
        f = self.previous_frame
        r = self.evaluate(self.semantic[f], self.euclid[f], self.is_ground[f], self.detection[f])
        self.records.append((self.sequence, f, *[float(v) for v in r]))
        self.previous_frame += 1

    def finish(self):
        """Evaluate the last frame that received points (the reference never flushes it; used by tests only)."""
        self._evaluate_previous()


class DeviceFrameScatter:
    """addColumnAndEvaluateFrameIfCompleted (kitti_demo.cpp:173-224) for all streams of a replay engine with the per-cell work on the GPU:
    cc_engine_scatter_info tells, per newly published column, which frames its points belong to (where frame N + 1 starts, the two error
    cases), cc_engine_scatter_apply writes is_ground / detection of the columns' points into per-frame arrays in HBM, cc_eval_frame_device
    evaluates a completed frame from them. Nothing per cell returns to the host; per step and stream the host sees two int32 per published
    column. The converter must write cc_kitti_frame::d_original_index of frame f of stream s to ``original_index_ptr(s, f)``."""

    SLOTS = 4  # frames kept per stream: a column is published within a rotation of its insertion

    def __init__(self, engine, sequence_ids, max_points: int, device: int = 0, rows: int = 64, cols: int = 2200):
        import torch
        from . import load_library
        self.e = engine
        self.L = load_library()
        i32, i64, vp = C.c_int, C.c_int64, C.c_void_p
        self.L.cc_engine_scatter_info.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp]
        self.L.cc_engine_scatter_apply.argtypes = [vp, i32, i64, i64, vp, i32, vp, vp, i64]
        self.seq = list(sequence_ids)
        S = len(self.seq)
        self.S, self.max_points, self.rows, self.cols = S, int(max_points), rows, cols
        dev = torch.device("cuda", device)
        self.d_org = torch.full((S, self.SLOTS, cols, rows), -1, dtype=torch.int32, device=dev)
        self.d_ground = torch.zeros((S, self.SLOTS, self.max_points), dtype=torch.uint8, device=dev)
        self.d_det = torch.zeros((S, self.SLOTS, self.max_points), dtype=torch.int32, device=dev)
        self.d_sem = torch.zeros((S, self.SLOTS, self.max_points), dtype=torch.int16, device=dev)
        self.d_eu = torch.zeros((S, self.SLOTS, self.max_points), dtype=torch.int32, device=dev)
        self.n_points = [[0] * self.SLOTS for _ in range(S)]
        self.previous_frame = [0] * S
        self.published_to = [-1] * S
        self.active = [True] * S
        self.records = [[] for _ in range(S)]

    def original_index_ptr(self, s: int, frame: int) -> int:
        return self.d_org[s, frame % self.SLOTS].data_ptr()

    def add_frame(self, s: int, frame: int, semantic, euclid):
        """Ground truth of frame `frame` of stream s (what kitti_demo.cpp:352-376 loads before the frame is fed)."""
        import torch
        n = int(len(semantic))
        if n > self.max_points:
            raise RuntimeError(f"frame with {n} points, capacity {self.max_points}")
        k = frame % self.SLOTS
        self.n_points[s][k] = n
        self.d_sem[s, k, :n] = torch.from_numpy(np.ascontiguousarray(semantic, dtype=np.uint16).view(np.int16)).to(self.d_sem.device)
        self.d_eu[s, k, :n] = torch.from_numpy(np.ascontiguousarray(euclid, dtype=np.uint32).view(np.int32)).to(self.d_eu.device)
        self.d_ground[s, k].zero_()
        self.d_det[s, k].zero_()
        torch.cuda.synchronize(self.d_sem.device)

    def _apply(self, s, lo, hi):
        if hi >= lo:
            rc = self.L.cc_engine_scatter_apply(self.e.h, s, lo, hi, self.d_org.data_ptr(), self.SLOTS, self.d_ground.data_ptr(),
                                                self.d_det.data_ptr(), self.max_points)
            if rc != 0:
                raise RuntimeError(f"cc_engine_scatter_apply failed with {rc}")

    def evaluate_previous(self, s):
        """evaluatePreviousFrame (kitti_demo.cpp:161-171)"""
        f = self.previous_frame[s]
        k = f % self.SLOTS
        n = self.n_points[s][k]
        if self.e.sync() != 0:
            raise RuntimeError(self.e.last_error())
        r = eval_frame_device(n, self.d_sem[s, k].data_ptr(), self.d_eu[s, k].data_ptr(), self.d_ground[s, k].data_ptr(), self.d_det[s, k].data_ptr())
        self.records[s].append((self.seq[s], f, *[float(v) for v in r]))
        self.previous_frame[s] += 1

    def publish(self):
        """Scatter everything the engine has published since the last call (all active streams, one info launch sequence + one D2H)."""
        streams, lo, hi = [], [], []
        for s in range(self.S):
            if not self.active[s]:
                continue
            h = self.e.state(s)["first_unpublished_global_column_index"] - 1
            l = max(self.published_to[s] + 1, 0)
            if h >= l:
                streams.append(s), lo.append(l), hi.append(h)
        if not streams:
            return
        a_s, a_lo, a_hi = np.array(streams, np.int32), np.array(lo, np.int64), np.array(hi, np.int64)
        total = int((a_hi - a_lo + 1).sum())
        mn, mx = np.zeros(total, np.int32), np.zeros(total, np.int32)
        rc = self.L.cc_engine_scatter_info(self.e.h, len(streams), a_s.ctypes.data, a_lo.ctypes.data, a_hi.ctypes.data, self.d_org.data_ptr(),
                                           self.SLOTS, mn.ctypes.data, mx.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"cc_engine_scatter_info failed with {rc}")
        o = 0
        for s, l, h in zip(streams, lo, hi):
            cnt = h - l + 1
            cmn, cmx = mn[o:o + cnt], mx[o:o + cnt]
            o += cnt
            start = 0
            while True:
                prev = self.previous_frame[s]
                seg_mn, seg_mx = cmn[start:], cmx[start:]
                has = seg_mx >= 0
                nxt = np.nonzero(has & (seg_mx == prev + 1))[0]
                end = int(nxt[0]) if len(nxt) else len(seg_mx) - 1   # the column in which frame prev + 1 starts closes the segment
                part = slice(0, end + 1)
                if (has[part] & (seg_mn[part] < prev)).any():
                    raise RuntimeError("Found a point belonging to a frame that was already evaluated!")   # kitti_demo.cpp:203-204
                if (seg_mx[part] > prev + 1).any():
                    raise RuntimeError("Found a point whose frame is too far ahead!")                      # kitti_demo.cpp:205-206
                self._apply(s, l + start, l + start + end)
                if not len(nxt):
                    break
                self.evaluate_previous(s)                                                                 # kitti_demo.cpp:221-222
                start += end + 1
                if start >= cnt:
                    break
            self.published_to[s] = h

    def finish(self, s):
        """"also evaluate final frame" (kitti_demo.cpp:417-419) with what has been published by now; the stream stops scattering."""
        self.evaluate_previous(s)
        self.active[s] = False


def gather_records(records, group=None, capacity: int | None = None) -> np.ndarray:
    """All-gather the per-frame records [(sequence, frame, tp, fn, fp, tn, OSE, USE)] of every rank and return them sorted by
    (sequence, frame) — the order in which the reference's single process would have produced them. Fixed-size padded buffers,
    one collective. Works on any initialised torch.distributed backend; the tensors live on the GPU for nccl
    (= RCCL) and on the CPU for gloo."""
    import torch
    import torch.distributed as dist
    local = np.asarray(records, dtype=np.float64).reshape(-1, 8)
    if not dist.is_available() or not dist.is_initialized():
        order = np.lexsort((local[:, 1], local[:, 0])) if len(local) else np.zeros(0, dtype=np.int64)
        return local[order]
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    # ONE collective: every rank sends a block of `capacity` + 1 rows, row 0 carrying its record count. The capacity is a bound every rank
    # knows without asking (the longest SemanticKITTI sequence has 4 661 frames, kitti_loader.cpp:552-562; a caller with more passes it)
    cap = int(capacity) if capacity else 8192
    if local.shape[0] > cap:
        raise ValueError(f"{local.shape[0]} records exceed the gather capacity {cap}")
    buf = torch.zeros((cap + 1, 8), dtype=torch.float64, device=dev)
    buf[0, 0] = float(local.shape[0])
    if local.shape[0]:
        buf[1: local.shape[0] + 1] = torch.from_numpy(local).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    parts = [b[1: int(b[0, 0].item()) + 1].cpu().numpy() for b in bufs]
    allr = np.concatenate(parts) if parts else np.zeros((0, 8))
    order = np.lexsort((allr[:, 1], allr[:, 0]))
    return allr[order]


def shard_streams(num_streams: int, world: int, rank: int):
    """Stream s is owned by rank s mod world: streams never interact, so ranks share nothing on the data path."""
    return [s for s in range(num_streams) if s % world == rank]
