"""Concurrent replay of KITTI-format sequences (BASELINE.json configs[4] shape): every sequence is one sensor stream of a
multi-stream engine, sequences are sharded over ranks (one process per GPU, sequence i on rank i mod world, no communication on
the data path), per-frame evaluation records are gathered with one all_gather at the end (RCCL over xGMI when the backend is nccl).

Per step every live stream advances by one frame: .bin -> cc_kitti_convert_frames (rows, un-correction, range image, pseudo-firings,
all on the GPU, written straight into the engine's input arrays) -> cc_engine_add_firings_device -> newly published columns are
scattered back to their frames (kitti_demo.cpp:173-224) -> cc_eval_frame on the GPU when a frame completes. What stays on the host is
file I/O, the per-firing pose interpolation and the bookkeeping of which cell belongs to which KITTI point.

    python -m continuous_clustering_amd.replay <root> 0 1 2 ...        # single process
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m continuous_clustering_amd.replay <root> 0 1 ...
"""
from __future__ import annotations

import os
import sys

import numpy as np

from . import Engine, capi, evaluation, kitti

NO_POINT = np.uint64(2 ** 64 - 1)


class Sequence:
    """Files and poses of one sequence folder (the setup part of KittiDemo::run, kitti_demo.cpp:238-274)."""

    def __init__(self, root: str, index: int, start_stamp: int = 1_700_000_000_000_000_000):
        self.index = index
        self.dir = os.path.join(root, "sequences", f"{index:02d}")
        times = [float(l) for l in open(os.path.join(self.dir, "times.txt")) if l.strip()]
        self.stamps = np.array([start_stamp + int(t * 1000000000) for t in times], dtype=np.uint64)
        self.start, self.end = kitti.start_end_stamps(self.stamps)
        calib = [l.split() for l in open(os.path.join(self.dir, "calib.txt"))]
        tr = np.array([float(v) for v in calib[4][1:13]])
        rows = [np.array([float(v) for v in l.split()]) for l in open(os.path.join(self.dir, "poses.txt")) if l.strip()]
        self.poses = np.stack([kitti.pose_from_line(r, tr) for r in rows[: len(times)]])
        self.n_frames = len(times)
        self.has_labels = os.path.isdir(os.path.join(self.dir, "labels"))
        self.has_gt = os.path.isdir(os.path.join(self.dir, "labels_euclidean_clustering"))

    def points(self, f: int) -> np.ndarray:
        return np.fromfile(os.path.join(self.dir, "velodyne", f"{f:06d}.bin"), dtype=np.float32).reshape(-1, 4)

    def labels(self, f: int, pts: np.ndarray):
        lab = np.fromfile(os.path.join(self.dir, "labels", f"{f:06d}.label"), dtype=np.uint16).reshape(-1, 2)
        if lab.shape[0] != pts.shape[0]:
            raise RuntimeError(f"Number of points does not match (label/bin): {lab.shape[0]} / {pts.shape[0]}")
        if self.has_gt:
            eu = np.fromfile(os.path.join(self.dir, "labels_euclidean_clustering", f"{f:06d}.label"), dtype=np.uint16)
        else:  # kitti_demo.cpp:337-346: generated online
            eu = evaluation.generate_euclidean_labels(pts, lab[:, 0], lab[:, 1])
        return lab[:, 0].copy(), eu.astype(np.uint32)


def replay(root: str, sequences, rank: int = 0, world: int = 1, device: int = 0, max_frames: int | None = None, timing: dict | None = None):
    """Replay this rank's share of `sequences` concurrently. Returns this rank's records [(sequence, frame, tp, fn, fp, tn, OSE, USE)]
    and a dict of totals. The frame scatter and the label compare run on the GPU (evaluation.DeviceFrameScatter): per step the host reads
    files, interpolates poses and looks at two integers per published column."""
    import time
    import torch
    mine = [Sequence(root, s) for i, s in enumerate(sequences) if i % world == rank]
    S = len(mine)
    if S == 0:
        return [], dict(streams=0, frames=0, cells_published=0)
    n_steps = max(q.n_frames for q in mine)
    if max_frames is not None:
        n_steps = min(n_steps, max_frames)
    dev = torch.device("cuda", device)
    cfg = capi.Config.kitti()
    engine = Engine(cfg, kitti.ROWS, S, device=device)
    max_points = 200000
    conv = kitti.KittiConverter(max_frames=S, max_points=max_points, device=device, hip_stream=engine.hip_stream())
    engine.set_option("input_on_engine_stream", 1)   # the converter writes the engine's inputs on the engine's HIP stream
    d_xyz = torch.full((S, kitti.COLS, kitti.ROWS, 3), float("nan"), dtype=torch.float32, device=dev)
    d_int = torch.zeros((S, kitti.COLS, kitti.ROWS), dtype=torch.uint8, device=dev)
    h_pose = np.zeros((S, kitti.COLS, 12), dtype=np.float64)
    scatter = evaluation.DeviceFrameScatter(engine, [q.index for q in mine], max_points, device=device, rows=kitti.ROWS, cols=kitti.COLS)
    for s, q in enumerate(mine):
        scatter.active[s] = q.has_labels
    frames_done = 0
    records = []
    t_io = t_dev = 0.0
    for f in range(n_steps):
        t0 = time.perf_counter()
        batch = []
        for s, q in enumerate(mine):
            if f < q.n_frames:
                pts = q.points(f)
                _, fposes = kitti.firing_stamps_and_poses(q.stamps, q.poses, q.start[f], q.end[f])
                h_pose[s] = fposes
                bins = kitti.bin_transforms(q.stamps, q.poses, q.start[f], q.end[f], q.poses[f])
                if scatter.active[s]:
                    sem, eu = q.labels(f, pts)
                    scatter.add_frame(s, f, sem, eu)
                frames_done += 1
            else:  # this sequence has ended: an empty rotation keeps the stream in step with the others
                pts, bins = np.zeros((0, 4), np.float32), None
                h_pose[s] = np.tile(q.poses[q.n_frames - 1], (kitti.COLS, 1))
            batch.append(dict(points=pts, stages=kitti.ALL_STAGES if bins is not None and len(bins) else kitti.ALL_STAGES & ~kitti.UNDO_EGO_MOTION,
                              start=q.start[min(f, q.n_frames - 1)], end=q.end[min(f, q.n_frames - 1)], bins=bins, d_xyz=d_xyz[s].data_ptr(),
                              d_intensity=d_int[s].data_ptr(), d_original_index=scatter.original_index_ptr(s, f)))
        d_pose = torch.from_numpy(h_pose).to(dev)
        torch.cuda.synchronize(dev)                                                     # the pose upload ran on torch's stream
        t1 = time.perf_counter()
        conv.convert(batch)                                                             # same HIP stream as the engine
        engine.add_firings_device(kitti.COLS, d_xyz.data_ptr(), d_int.data_ptr(), d_pose.data_ptr())
        if engine.sync() != 0:
            raise RuntimeError(engine.last_error())
        scatter.publish()
        for s, q in enumerate(mine):
            if scatter.active[s] and f == min(q.n_frames, n_steps) - 1:
                scatter.finish(s)        # "also evaluate final frame" (kitti_demo.cpp:417-419) with what has been published by now
                records.extend(scatter.records[s])   # columns the padding rotations flush later belong to no evaluation
        t2 = time.perf_counter()
        t_io += t1 - t0
        t_dev += t2 - t1
    tot = engine.totals()
    if timing is not None:
        timing.update(host_io_s=t_io, device_s=t_dev)
    return records, dict(streams=S, frames=frames_done, cells_published=int(tot["cells_published"]))


def main(argv=None):
    import torch
    import torch.distributed as dist
    argv = sys.argv[1:] if argv is None else argv
    root, sequences = argv[0], [int(a) for a in argv[1:]]
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
    records, totals = replay(root, sequences, rank, world, device=local)
    allr = evaluation.gather_records(records)
    if rank == 0:
        for seq in sorted(set(int(r[0]) for r in allr)):
            rs = allr[allr[:, 0] == seq][:, 2:8]
            print(evaluation.format_row(f"{seq:02d}", evaluation.summarize(rs)))
        print(f"frames evaluated: {len(allr)}; this rank: {totals}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
