#!/usr/bin/env python3
"""Regenerate tests/golden/g_kitti_replay.npz and tests/golden/g_gt_labels.npz.

PROVENANCE: like make_golden.py — produced by the CPU oracle (oracle/kitti_oracle.cpp, oracle/gt_oracle.cpp), restatements of
KittiLoader / kitti_demo's firing builder / generateEuclideanClusteringLabels. The reference holds no vectors for these functions and
cannot be built here (Eigen3 / PCL absent): regression vectors that pin the oracle ("parity unpinned" w.r.t. the reference).

    python tests/golden/make_kitti_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from continuous_clustering_amd import kitti  # noqa: E402  (synthetic inputs and the host pose helpers are not used for the expected values)
from oracle import pyoracle as orc  # noqa: E402


def small_cloud(seed):
    """~6 k points: 24 rows of ~260 returns, with pile-ups, a row overflow region and two NaN points."""
    pts, rows = kitti.synthetic_frame(seed, n_rows=24, cols_per_row=300, dropout=0.1, duplicate=0.3)
    pts[17, 0] = np.nan
    pts[4000, 1] = np.nan
    return pts


def main():
    rows, times = kitti.synthetic_poses(4, (9.0, 0.4, 0.02, 0.35))
    poses = np.stack([orc.kitti_pose_from_line(r, kitti.CALIB_TR) for r in rows])
    stamps = (times * 1e9).astype(np.uint64) + np.uint64(1_700_000_000_000_000_000)
    start, end = orc.kitti_start_end_stamps(stamps)
    f = 2
    pts = small_cloud(5)
    laser, found, maxc, _ = orc.kitti_recover_laser_indices(pts)
    unc = orc.kitti_undo_ego_motion(pts, start[f], end[f], poses[f], stamps, poses)
    cells, skipped = orc.kitti_generate_range_image(unc, laser, True)
    cells_plain, _ = orc.kitti_generate_range_image(unc, laser, False)
    xyz, inten, unique, fstamps = orc.kitti_make_firings(unc, cells, start[f], end[f], 3, f)
    filled = np.argwhere(cells >= 0)
    np.savez_compressed(os.path.join(HERE, "g_kitti_replay.npz"), points=pts, stamps=stamps, poses=poses, start=start, end=end, frame=f,
                        laser=laser, rows_found=found, max_columns=maxc, skipped=skipped, uncorrected=unc,
                        cell_rc=filled.astype(np.int32), cell_src=cells[cells >= 0], plain_rc=np.argwhere(cells_plain >= 0).astype(np.int32),
                        plain_src=cells_plain[cells_plain >= 0], bins=orc.kitti_bin_transforms(stamps, poses, start[f], end[f], poses[f]),
                        firing_stamps=fstamps, firing_pose_0=orc.kitti_interpolate(stamps, poses, int(fstamps[0])),
                        firing_pose_1099=orc.kitti_interpolate(stamps, poses, int(fstamps[1099])),
                        intensity_sum=np.int64(inten.astype(np.int64).sum()), unique_xor=np.bitwise_xor.reduce(unique.reshape(-1)))
    rng = np.random.default_rng(8)
    centers = rng.uniform(-15, 15, (30, 3)).astype(np.float32)
    which = rng.integers(0, 30, 3000)
    gpts = np.concatenate([centers[which] + rng.normal(0, 0.4, (3000, 3)).astype(np.float32), rng.random((3000, 1)).astype(np.float32)], axis=1)
    sem = rng.choice(np.array([10, 30, 40, 50, 0, 72, 80], dtype=np.uint16), 30)[which]
    inst = (which % 3).astype(np.uint16)
    labels, nclusters = orc.generate_euclidean_labels(gpts, sem, inst)
    np.savez_compressed(os.path.join(HERE, "g_gt_labels.npz"), points=gpts.astype(np.float32), semantic=sem, instance=inst, labels=labels,
                        clusters=nclusters)
    for n in ("g_kitti_replay.npz", "g_gt_labels.npz"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")


if __name__ == "__main__":
    main()
