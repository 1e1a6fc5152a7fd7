"""Real-data acceptance (BASELINE.json configs[0] / configs[4]): replay SemanticKITTI train sequences through the MI355X hot path exactly like the
reference's no-ROS harness (src/tools/kitti_demo.cpp: loader -> 2200 pseudo-firings per frame -> addFiring -> frame scatter -> label compare) and
compare the per-sequence table of generateEvaluationResults (kitti_evaluation.cpp:159-213) with the one the reference publishes for itself
(README.md:213-245), cell by cell, at the printed two decimals.

The dataset is not part of this repository or of the build image; everything here runs only where a root with `sequences/NN/{velodyne,labels,
poses.txt,times.txt,calib.txt}` is mounted (`$SEMANTIC_KITTI_ROOT`, `bench.py --kitti-root`, tests/test_gpu_semantickitti.py). Ground-truth
euclidean-clustering labels are read from `labels_euclidean_clustering/` when present and generated on the GPU otherwise
(cc_eval_generate_euclidean_labels; kitti_demo.cpp:337-346 does the same with PCL).

Multi-GPU: sequence i runs on rank i mod world (no data-path collective), the per-frame records meet in ONE all_gather (evaluation.gather_records).
"""
from __future__ import annotations

import json
import os

import numpy as np

METRICS = ("recall", "precision", "f1", "accuracy", "use", "ose")
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_TABLES = os.path.join(_ROOT, "tests", "golden", "semantickitti_readme_tables.json")
TRAIN_SEQUENCES = tuple(range(11))
TRAIN_FRAMES = {0: 4541, 1: 1101, 2: 4661, 3: 801, 4: 271, 5: 2761, 6: 1101, 7: 1101, 8: 4071, 9: 1591, 10: 1201}  # kitti_loader.cpp:552-562


def load_tables(path: str = DEFAULT_TABLES) -> dict:
    return json.load(open(path))["tables"]


def printed(summary: dict) -> dict:
    """The six "mu / sigma" cells of one table row as the reference prints them (std::fixed, setprecision(2); the four ground metrics x 100)."""
    out = {}
    for i, k in enumerate(METRICS):
        m, s = summary[k]
        if i < 4:
            m, s = m * 100, s * 100
        out[k] = [f"{m:.2f}", f"{s:.2f}"]
    return out


def available_sequences(root: str, wanted=TRAIN_SEQUENCES):
    ok = []
    for s in wanted:
        d = os.path.join(root, "sequences", f"{s:02d}")
        if all(os.path.exists(os.path.join(d, f)) for f in ("velodyne", "labels", "poses.txt", "times.txt", "calib.txt")):
            ok.append(s)
    return ok


def compare(records: np.ndarray, sequences, tables: dict | None = None, complete: bool = True) -> dict:
    """records: [k, 8] (sequence, frame, tp, fn, fp, tn, OSE, USE) of all ranks. Returns per-sequence rows {metric: {got, want, ok}}, the
    'all' row when every train sequence took part, and `all_ok`. With complete = False (frame cap) nothing is asserted: rows carry ok = None."""
    from . import evaluation
    tables = load_tables() if tables is None else tables
    records = np.asarray(records, dtype=np.float64).reshape(-1, 8)
    rows = {}

    def one(key, rs):
        got = printed(evaluation.summarize(rs[:, 2:8]))
        want = tables.get(key)
        rows[key] = {m: {"got": " / ".join(got[m]), "want": " / ".join(want[m]) if want else None,
                         "ok": (got[m] == want[m]) if (complete and want) else None} for m in METRICS}
        rows[key]["frames"] = int(rs.shape[0])

    for s in sequences:
        rs = records[records[:, 0] == s]
        if rs.shape[0]:
            order = np.argsort(rs[:, 1], kind="stable")
            one(str(int(s)), rs[order])
    if sorted(int(s) for s in sequences) == list(TRAIN_SEQUENCES):
        # evaluation_per_sequence[-1] receives every frame in the order the sequences are run (kitti_demo.cpp: sequence loop, frame loop)
        order = np.lexsort((records[:, 1], records[:, 0]))
        one("all", records[order])
    checked = [c["ok"] for r in rows.values() for k, c in r.items() if k != "frames" and c["ok"] is not None]
    return {"rows": rows, "cells_checked": len(checked), "cells_equal": int(sum(checked)), "all_ok": bool(checked) and all(checked) if complete else None}


def run(root: str, sequences=TRAIN_SEQUENCES, rank: int = 0, world: int = 1, device: int = 0, max_frames: int | None = None,
        tables: dict | None = None) -> dict:
    """Replay + gather + compare. Every rank calls this (the gather is a collective when torch.distributed is initialised)."""
    import time
    from . import evaluation, replay
    seqs = available_sequences(root, sequences)
    if not seqs:
        raise FileNotFoundError(f"no SemanticKITTI sequence folders under {root}/sequences")
    timing = {}
    t0 = time.perf_counter()
    recs, totals = replay.replay(root, seqs, rank=rank, world=world, device=device, max_frames=max_frames, timing=timing)
    el = time.perf_counter() - t0
    allr = evaluation.gather_records(recs)
    out = compare(allr, seqs, tables, complete=max_frames is None)
    out.update(sequences=seqs, frames=int(len(allr)), seconds=el, frames_per_s_this_rank=totals["frames"] / max(el, 1e-9),
               device_s=timing.get("device_s"), host_io_s=timing.get("host_io_s"), world=world,
               frame_cap=max_frames, source="README.md:213-245 of the reference (commit fa3c53b), compared at the printed two decimals")
    return out


def main(argv=None):
    import sys
    import torch
    import torch.distributed as dist
    argv = sys.argv[1:] if argv is None else argv
    root = argv[0] if argv else os.environ.get("SEMANTIC_KITTI_ROOT", "")
    seqs = [int(a) for a in argv[1:]] or list(TRAIN_SEQUENCES)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
    res = run(root, seqs, rank, world, device=local)
    if rank == 0:
        for key, row in res["rows"].items():
            print(key, {m: (c["got"], c["want"], c["ok"]) for m, c in row.items() if m != "frames"})
        print("all_ok:", res["all_ok"], f"({res['cells_equal']}/{res['cells_checked']} cells)")
    if world > 1:
        dist.destroy_process_group()
    return 0 if res["all_ok"] else 1


if __name__ == "__main__":
    raise SystemExit(main())
