"""CPU: the oracle of the KITTI replay path and of the ground-truth labels against the committed regression vectors
(tests/golden/g_kitti_replay.npz, g_gt_labels.npz; provenance in tests/golden/make_kitti_golden.py)."""
import os

import numpy as np

from oracle import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dense(rc, src):
    cells = np.full((64, 2200), -1, dtype=np.int32)
    cells[rc[:, 0], rc[:, 1]] = src
    return cells


def test_oracle_reproduces_the_kitti_vectors():
    g = np.load(os.path.join(GOLD, "g_kitti_replay.npz"))
    f = int(g["frame"])
    laser, found, maxc, _ = orc.kitti_recover_laser_indices(g["points"])
    assert np.array_equal(laser, g["laser"]) and found == int(g["rows_found"]) and maxc == int(g["max_columns"])
    unc = orc.kitti_undo_ego_motion(g["points"], g["start"][f], g["end"][f], g["poses"][f], g["stamps"], g["poses"])
    a, b = unc, g["uncorrected"]
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)])
    cells, skipped = orc.kitti_generate_range_image(unc, laser, True)
    assert skipped == int(g["skipped"]) and np.array_equal(cells, dense(g["cell_rc"], g["cell_src"]))
    plain, _ = orc.kitti_generate_range_image(unc, laser, False)
    assert np.array_equal(plain, dense(g["plain_rc"], g["plain_src"]))
    assert np.array_equal(orc.kitti_bin_transforms(g["stamps"], g["poses"], g["start"][f], g["end"][f], g["poses"][f]).view(np.uint64),
                          g["bins"].view(np.uint64))
    xyz, inten, unique, fstamps = orc.kitti_make_firings(unc, cells, g["start"][f], g["end"][f], 3, f)
    assert np.array_equal(fstamps, g["firing_stamps"]) and int(inten.astype(np.int64).sum()) == int(g["intensity_sum"])
    assert np.bitwise_xor.reduce(unique.reshape(-1)) == g["unique_xor"]
    assert np.array_equal(orc.kitti_interpolate(g["stamps"], g["poses"], int(fstamps[1099])).view(np.uint64), g["firing_pose_1099"].view(np.uint64))


def test_oracle_reproduces_the_label_vectors():
    g = np.load(os.path.join(GOLD, "g_gt_labels.npz"))
    labels, n = orc.generate_euclidean_labels(g["points"], g["semantic"], g["instance"])
    assert n == int(g["clusters"]) and np.array_equal(labels, g["labels"])
