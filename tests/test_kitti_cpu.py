"""CPU: the KITTI replay path upstream of insertion.

* the oracle restatement (oracle/kitti_oracle.cpp) against properties of the format (recovered rows == generating rows, every
  cell holds a distinct point, shift rule never loses more points than plain overwrite, identity motion is a no-op),
* the product's HOST pose arithmetic (cc_kitti_pose_interpolate, cc_kitti_bin_transforms, ... — plain C inside libcc_hip.so,
  no device work) bit-for-bit against the oracle's separate restatement of the same Eigen algorithms,
* the struct layouts of include/cc_kitti.h, and that the device part refuses to run without a GPU.
"""
import ctypes

import numpy as np
import pytest

from continuous_clustering_amd import kitti
from oracle import pyoracle as orc


def _drive(n_frames=6, motion=(8.0, 0.5, 0.02, 0.3)):
    rows, times = kitti.synthetic_poses(n_frames, motion)
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.0])
    poses = np.stack([orc.kitti_pose_from_line(r, kitti.CALIB_TR) for r in rows])
    stamps = (times * 1e9).astype(np.uint64) + np.uint64(1_700_000_000_000_000_000)
    return stamps, poses, rows


def test_struct_layouts():
    assert ctypes.sizeof(kitti.Frame) == 8 + 8 + 8 + 4 + 4 + 8 + 8 + 8 + 8 + 8 + 8
    assert ctypes.sizeof(kitti.FrameInfo) == 4 + 4 + 8 + 8


def test_oracle_recovers_the_generating_rows():
    pts, rows = kitti.synthetic_frame(seed=3)
    laser, found, maxc, threw = orc.kitti_recover_laser_indices(pts)
    assert found == 64 and not threw
    assert np.array_equal(laser, rows)
    assert maxc == max(np.bincount(rows)[:63])


def test_oracle_row_overflow_and_short_frames():
    pts, rows = kitti.synthetic_frame(seed=4)
    extra = pts[rows >= 60]
    laser, found, _, _ = orc.kitti_recover_laser_indices(np.concatenate([pts, extra]))
    assert found == 65                        # "Wrong number of rows found: 65" (kitti_loader.cpp:92-94)
    assert np.array_equal(laser[: pts.shape[0]], rows)
    assert not laser[pts.shape[0]:].any()     # everything from the break on keeps row 0 (:74-76)
    laser, found, _, _ = orc.kitti_recover_laser_indices(pts[rows < 10])
    assert found == 10
    laser, found, maxc, _ = orc.kitti_recover_laser_indices(np.zeros((0, 4), np.float32))
    assert found == 1 and maxc == 0


def test_oracle_range_image_properties():
    pts, rows = kitti.synthetic_frame(seed=5, duplicate=0.2)
    cells, skipped = orc.kitti_generate_range_image(pts, rows, shift=True)
    plain, _ = orc.kitti_generate_range_image(pts, rows, shift=False)
    assert skipped == 0
    filled = cells[cells >= 0]
    assert len(np.unique(filled)) == len(filled)                       # a point sits in at most one cell
    assert (rows[filled] == np.nonzero(cells >= 0)[0]).all()           # and in its own row
    assert (cells >= 0).sum() > (plain >= 0).sum()                     # shifting rescues collisions
    assert (plain >= 0).sum() < pts.shape[0]
    # without shifting a cell keeps the LAST point that maps to it (kitti_loader.cpp:163-167)
    az = np.arctan2(pts[:, 1], pts[:, 0]).astype(np.float32).astype(np.float64)
    col = np.minimum(((np.pi - az) / (2 * np.pi / 2200)).astype(np.int64), 2199)
    last = {}
    for i, (r, c) in enumerate(zip(rows, col)):
        last[(int(r), int(c))] = i
    for (r, c), i in list(last.items())[::97]:
        assert plain[r, c] == i


def test_oracle_identity_motion_leaves_points_alone():
    pts, _ = kitti.synthetic_frame(seed=6)
    stamps = np.array([0, 100_000_000, 200_000_000], dtype=np.uint64) + np.uint64(10**18)
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.0])
    poses = np.stack([ident] * 3)
    out = orc.kitti_undo_ego_motion(pts, stamps[1] - 50_000_000, stamps[1] + 50_000_000, ident, stamps, poses)
    assert np.array_equal(out.view(np.uint32), pts.view(np.uint32))


def test_oracle_interpolate_endpoints_and_rigidity():
    stamps, poses, _ = _drive()
    assert np.array_equal(orc.kitti_interpolate(stamps, poses, int(stamps[0]) - 5), poses[0])
    assert np.array_equal(orc.kitti_interpolate(stamps, poses, int(stamps[0])), poses[0])
    for k in range(1, len(stamps)):  # an interior node is reached with f == 1 through the quaternion round trip (kitti_loader.cpp:309-316)
        assert np.allclose(orc.kitti_interpolate(stamps, poses, int(stamps[k])), poses[k], atol=1e-12)
    assert np.array_equal(orc.kitti_interpolate(stamps, poses, int(stamps[-1]) + 10**9), poses[-1])
    mid = orc.kitti_interpolate(stamps, poses, (int(stamps[2]) + int(stamps[3])) // 2).reshape(3, 4)
    R = mid[:, :3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
    assert np.allclose(mid[:, 3], (poses[2].reshape(3, 4)[:, 3] + poses[3].reshape(3, 4)[:, 3]) / 2, atol=1e-9)


def test_host_pose_arithmetic_matches_oracle_bit_for_bit():
    stamps, poses, rows = _drive(8, motion=(11.0, -0.7, 0.03, 0.45))
    for r in rows:
        assert np.array_equal(kitti.pose_from_line(r, kitti.CALIB_TR), orc.kitti_pose_from_line(r, kitti.CALIB_TR))
    rng = np.random.default_rng(0)
    for q in rng.integers(int(stamps[0]) - 10**7, int(stamps[-1]) + 10**7, 400):
        a, b = kitti.pose_interpolate(stamps, poses, int(q)), orc.kitti_interpolate(stamps, poses, int(q))
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), q
    start, end = kitti.start_end_stamps(stamps)
    so, eo = orc.kitti_start_end_stamps(stamps)
    assert np.array_equal(start, so) and np.array_equal(end, eo)
    for f in (0, 3, 7):
        a = kitti.bin_transforms(stamps, poses, start[f], end[f], poses[f])
        b = orc.kitti_bin_transforms(stamps, poses, start[f], end[f], poses[f])
        assert a.shape == b.shape and a.shape[0] == 100
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
        fs, fp = kitti.firing_stamps_and_poses(stamps, poses, start[f], end[f])
        _, _, _, os_ = orc.kitti_make_firings(np.zeros((0, 4), np.float32), np.full((64, 2200), -1, np.int32), start[f], end[f])
        assert np.array_equal(fs, os_)
        for c in (0, 1, 1099, 2199):
            assert np.array_equal(fp[c].view(np.uint64), orc.kitti_interpolate(stamps, poses, int(fs[c])).view(np.uint64))


def test_rotation_about_other_axes_takes_the_other_quaternion_branches():
    """trace <= 0 branches of the matrix -> quaternion conversion (rotations by ~pi about x, y, z)."""
    def rot(axis, ang):
        c, s = np.cos(ang), np.sin(ang)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        R = np.eye(3)
        R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
        return R
    stamps = np.array([10**18, 10**18 + 10**8], dtype=np.uint64)
    for axis in range(3):
        poses = []
        for ang in (3.0, 3.1):
            T = np.zeros((3, 4))
            T[:, :3] = rot(axis, ang)
            T[:, 3] = (ang, -ang, 0.5)
            poses.append(T.reshape(12))
        poses = np.stack(poses)
        for q in (10**18 + 1, 10**18 + 3 * 10**7, 10**18 + 10**8 - 1):
            a, b = kitti.pose_interpolate(stamps, poses, q), orc.kitti_interpolate(stamps, poses, q)
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
            R = a.reshape(3, 4)[:, :3]
            assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)


def test_no_gpu_means_no_converter():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from continuous_clustering_amd import EngineError, capi
    with pytest.raises(EngineError) as ei:
        kitti.KittiConverter()
    assert ei.value.code == capi.CC_ERR_NO_DEVICE
