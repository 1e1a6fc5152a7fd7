"""GPU parity of the KITTI replay path (include/cc_kitti.h) against the oracle (oracle/kitti_oracle.cpp): bit-exact rows,
un-corrected points, range-image winners and firing arrays, per stage and chained, including the reference's edge cases;
then the whole chain .bin -> firings in HBM -> cc_engine_add_firings_device against oracle loader -> oracle clustering."""
import numpy as np
import pytest

from continuous_clustering_amd import capi, kitti
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu

IDENT = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.0])


def _drive(n_frames=4, motion=(8.0, 0.5, 0.02, 0.3)):
    rows, times = kitti.synthetic_poses(n_frames, motion)
    poses = np.stack([kitti.pose_from_line(r, kitti.CALIB_TR) for r in rows])
    stamps = (times * 1e9).astype(np.uint64) + np.uint64(1_700_000_000_000_000_000)
    start, end = kitti.start_end_stamps(stamps)
    return stamps, poses, start, end


def _bits(a):
    return a.view(np.uint32)


def _oracle_chain(pts, stamps, poses, start, end, f, undo=True, shift=True):
    laser, found, maxc, _ = orc.kitti_recover_laser_indices(pts)
    unc = orc.kitti_undo_ego_motion(pts, start[f], end[f], poses[f], stamps, poses) if undo else pts.copy()
    cells, skipped = orc.kitti_generate_range_image(unc, laser, shift)
    xyz, inten, unique, fstamps = orc.kitti_make_firings(unc, cells, start[f], end[f], 0, f)
    return dict(laser=laser, found=found, maxc=maxc, unc=unc, cells=cells, skipped=skipped, xyz=xyz, inten=inten, unique=unique,
                stamps=fstamps)


def _compare(res, o, n):
    assert res["rows_found"] == o["found"]
    assert res["max_columns"] == o["maxc"]
    assert res["skipped"] == o["skipped"]
    assert np.array_equal(res["laser_index"], o["laser"])
    assert np.array_equal(_bits(res["points"]), _bits(o["unc"]))
    assert np.array_equal(res["cell_source"], o["cells"])


@pytest.mark.parametrize("seed,motion,dup", [(1, (8.0, 0.0, 0.0, 0.05), 0.02), (2, (15.0, -1.0, 0.1, 0.6), 0.1), (3, (0.0, 0.0, 0.0, 0.0), 0.5)])
def test_full_chain_matches_oracle(seed, motion, dup):
    import torch
    stamps, poses, start, end = _drive(4, motion)
    conv = kitti.KittiConverter(max_frames=1)
    for f in (0, 2):
        pts, _ = kitti.synthetic_frame(seed * 10 + f, motion=motion, duplicate=dup)
        o = _oracle_chain(pts, stamps, poses, start, end, f)
        d_xyz = torch.empty((2200, 64, 3), dtype=torch.float32, device="cuda")
        d_int = torch.empty((2200, 64), dtype=torch.uint8, device="cuda")
        d_org = torch.empty((2200, 64), dtype=torch.int32, device="cuda")
        bins = kitti.bin_transforms(stamps, poses, start[f], end[f], poses[f])
        conv.convert([dict(points=pts, stages=kitti.ALL_STAGES, start=start[f], end=end[f], bins=bins, d_xyz=d_xyz.data_ptr(),
                           d_intensity=d_int.data_ptr(), d_original_index=d_org.data_ptr())])
        res = conv.result(0, pts.shape[0])
        _compare(res, o, pts.shape[0])
        torch.cuda.synchronize()
        gx = d_xyz.cpu().numpy()
        assert np.array_equal(np.isnan(gx), np.isnan(o["xyz"]))
        assert np.array_equal(_bits(gx)[~np.isnan(gx)], _bits(o["xyz"])[~np.isnan(o["xyz"])])
        assert np.array_equal(d_int.cpu().numpy(), o["inten"])
        org = d_org.cpu().numpy()
        assert np.array_equal(org, o["cells"].T)
        # globally_unique_point_index (kitti_demo.cpp:152-156) from the device's original index
        unique = (np.uint64(0) << np.uint64(48)) | (np.uint64(f) << np.uint64(32)) | org.astype(np.int64).astype(np.uint64)
        assert np.array_equal(unique, o["unique"])
        # the firing stamps of the host helper
        fs, _ = kitti.firing_stamps_and_poses(stamps, poses, start[f], end[f])
        assert np.array_equal(fs, o["stamps"])


def test_stages_run_separately_like_the_loader_methods():
    """recoverLaserIndices / undoEgoMotionCorrection / generateRangeImage as three calls with host round trips, rows given by the caller
    in arbitrary (non-contiguous) order, and the no-shift variant."""
    stamps, poses, start, end = _drive(3, (10.0, 0.3, 0.0, 0.4))
    pts, rows = kitti.synthetic_frame(21, duplicate=0.15)
    conv = kitti.KittiConverter(max_frames=1)
    conv.convert([dict(points=pts, stages=kitti.RECOVER_ROWS)])
    r1 = conv.result(0, pts.shape[0], cells=False)
    laser, found, maxc, _ = orc.kitti_recover_laser_indices(pts)
    assert np.array_equal(r1["laser_index"], laser) and r1["rows_found"] == found and r1["max_columns"] == maxc
    assert np.array_equal(_bits(r1["points"]), _bits(pts))
    bins = kitti.bin_transforms(stamps, poses, start[1], end[1], poses[1])
    conv.convert([dict(points=pts, stages=kitti.UNDO_EGO_MOTION, start=start[1], end=end[1], bins=bins)])
    r2 = conv.result(0, pts.shape[0], cells=False)
    unc = orc.kitti_undo_ego_motion(pts, start[1], end[1], poses[1], stamps, poses)
    assert np.array_equal(_bits(r2["points"]), _bits(unc))
    rng = np.random.default_rng(5)
    perm = rng.permutation(pts.shape[0])          # rows no longer contiguous: the stable sort has to do real work
    for shift in (True, False):
        st = kitti.RANGE_IMAGE | (kitti.SHIFT_OCCUPIED if shift else 0)
        conv.convert([dict(points=unc[perm], laser_index=laser[perm], stages=st)])
        r3 = conv.result(0, pts.shape[0])
        cells, skipped = orc.kitti_generate_range_image(unc[perm], laser[perm], shift)
        assert np.array_equal(r3["cell_source"], cells) and r3["skipped"] == skipped


def test_edge_cases():
    conv = kitti.KittiConverter(max_frames=1)
    pts, rows = kitti.synthetic_frame(31)
    # (a) empty cloud
    conv.convert([dict(points=np.zeros((0, 4), np.float32), stages=kitti.RECOVER_ROWS | kitti.RANGE_IMAGE | kitti.SHIFT_OCCUPIED)])
    r = conv.result(0, 0)
    assert r["rows_found"] == 1 and r["max_columns"] == 0 and (r["cell_source"] == -1).all()
    # (b) more than 64 rows: the 65th row and everything after it fall back to row 0 (kitti_loader.cpp:74-76) and collide there
    extra = pts[rows >= 58]
    big = np.concatenate([pts, extra])
    conv = kitti.KittiConverter(max_frames=1, max_points=big.shape[0])
    conv.convert([dict(points=big, stages=kitti.RECOVER_ROWS | kitti.RANGE_IMAGE | kitti.SHIFT_OCCUPIED)])
    r = conv.result(0, big.shape[0])
    laser, found, maxc, _ = orc.kitti_recover_laser_indices(big)
    assert found == 65 and r["rows_found"] == 65 and r["break_index"] == pts.shape[0] and r["max_columns"] == maxc
    assert np.array_equal(r["laser_index"], laser)
    cells, _ = orc.kitti_generate_range_image(big, laser, True)
    assert np.array_equal(r["cell_source"], cells)
    # (c) fewer rows, one single point
    for sub in (pts[rows < 7], pts[:1]):
        conv.convert([dict(points=sub, stages=kitti.RECOVER_ROWS | kitti.RANGE_IMAGE | kitti.SHIFT_OCCUPIED)])
        r = conv.result(0, sub.shape[0])
        laser, found, maxc, _ = orc.kitti_recover_laser_indices(sub)
        assert r["rows_found"] == found and r["max_columns"] == maxc and np.array_equal(r["laser_index"], laser)
        assert np.array_equal(r["cell_source"], orc.kitti_generate_range_image(sub, laser, True)[0])
    # (d) points on the axes: azimuth exactly 0, pi, -pi (column 2200 -> 2199, :124-125), +-pi/2, NaN coordinates, zeros
    special = np.array([[5, 0, 0, .5], [-5, 0, 0, .5], [-5, -0.0, 0, .5], [0, 5, 0, .5], [0, -5, 0, .5], [0, 0, 1, .5], [-0.0, 0, 1, .5],
                        [np.nan, 1, 0, .5], [1, np.nan, 0, .5], [-7, 1e-30, 0, .2], [-7, -1e-30, 0, .2], [3, 3, 1, 1.0], [3, 3, 1, 0.999],
                        [3, -3, 1, 1.5], [2, -3, 1, -0.1], [2, -2.5, 1, 1e12], [2, -2.2, 1, np.nan]], dtype=np.float32)
    import torch
    d_int = torch.zeros((2200, 64), dtype=torch.uint8, device="cuda")
    conv.convert([dict(points=special, stages=kitti.RANGE_IMAGE | kitti.SHIFT_OCCUPIED | kitti.FIRINGS, d_intensity=d_int.data_ptr())])
    r = conv.result(0, special.shape[0])
    cells, skipped = orc.kitti_generate_range_image(special, np.zeros(len(special), np.uint8), True)
    assert skipped == 2 and r["skipped"] == 2
    assert np.array_equal(r["cell_source"], cells)
    assert cells[0, 2199] >= 0 and cells[0, 0] >= 0
    _, inten, _, _ = orc.kitti_make_firings(special, cells, 0, 10**8)
    torch.cuda.synchronize()
    assert np.array_equal(d_int.cpu().numpy(), inten)
    # (e) a rotation that lasts a whole number of milliseconds with a point at azimuth -pi: last bin, not one past it
    stamps, poses, start, end = _drive(3)
    s0, e0 = int(start[1]), int(start[1]) + 100_000_000
    bins = kitti.bin_transforms(stamps, poses, s0, e0, poses[1])
    assert bins.shape[0] == 100
    conv.convert([dict(points=special, stages=kitti.UNDO_EGO_MOTION, start=s0, end=e0, bins=bins)])
    r = conv.result(0, special.shape[0], cells=False)
    unc = orc.kitti_undo_ego_motion(special, s0, e0, poses[1], stamps, poses)
    a, b = r["points"], unc
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(_bits(a)[~np.isnan(a)], _bits(b)[~np.isnan(b)])


def test_batch_of_frames_in_one_call():
    stamps, poses, start, end = _drive(5, (12.0, 0.0, 0.0, 0.2))
    frames, want = [], []
    for f in range(5):
        pts, _ = kitti.synthetic_frame(40 + f, dropout=0.05 + 0.1 * f, duplicate=0.05)
        frames.append(dict(points=pts, stages=kitti.ALL_STAGES & ~kitti.FIRINGS, start=start[f], end=end[f],
                           bins=kitti.bin_transforms(stamps, poses, start[f], end[f], poses[f])))
        want.append(_oracle_chain(pts, stamps, poses, start, end, f))
    conv = kitti.KittiConverter(max_frames=5)
    conv.convert(frames)
    for f in range(5):
        _compare(conv.result(f, frames[f]["points"].shape[0]), want[f], frames[f]["points"].shape[0])


def test_replay_into_the_engine_matches_oracle_end_to_end():
    """.bin -> (GPU) firings in HBM -> cc_engine_add_firings_device, against oracle loader -> oracle clustering: the published
    ground labels and canonical cluster ids of three replayed frames."""
    import torch
    import util
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    NF = 3
    motion = (9.0, 0.2, 0.0, 0.25)
    stamps, poses, start, end = _drive(NF, motion)
    cfg = capi.Config.kitti()
    e = Engine(cfg, 64, 1)
    o = Oracle(cfg, 64)
    conv = kitti.KittiConverter(max_frames=1, hip_stream=e.hip_stream())
    e.set_option("input_on_engine_stream", 1)
    d_xyz = torch.empty((1, 2200, 64, 3), dtype=torch.float32, device="cuda")
    d_int = torch.empty((1, 2200, 64), dtype=torch.uint8, device="cuda")
    for f in range(NF):
        pts, _ = kitti.synthetic_frame(70 + f, motion=motion)
        oc = _oracle_chain(pts, stamps, poses, start, end, f)
        fposes = np.stack([orc.kitti_interpolate(stamps, poses, int(s)) for s in oc["stamps"]])
        assert o.add_firings(oc["xyz"], oc["inten"], fposes) == 0, o.last_error()
        _, gposes = kitti.firing_stamps_and_poses(stamps, poses, start[f], end[f])
        d_pose = torch.from_numpy(gposes.reshape(1, 2200, 12)).cuda()
        torch.cuda.synchronize()
        conv.convert([dict(points=pts, stages=kitti.ALL_STAGES, start=start[f], end=end[f],
                           bins=kitti.bin_transforms(stamps, poses, start[f], end[f], poses[f]), d_xyz=d_xyz.data_ptr(),
                           d_intensity=d_int.data_ptr())])
        lo = max(0, e.state(0)["first_unpublished_global_column_index"])
        e.add_firings_device(2200, d_xyz.data_ptr(), d_int.data_ptr(), d_pose.data_ptr())   # same HIP stream: no host sync between
        assert e.sync() == 0, e.last_error()
        so, se = o.state(), e.state(0)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], k
        hi = se["first_unpublished_global_column_index"] - 1   # what this call published is still in the ring
        lo = max(lo, o.published_range()[0])
        if hi >= lo:
            util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi), lo)
    assert se["clusters_finished"] > 5 and se["cells_published"] > 64 * 2 * 2200


def _random_cloud(rng):
    """An unorganised cloud in .bin order with random row count (also > 64), row lengths (also > 2200: the reference throws), azimuth
    jitter, duplicates, gaps and a few NaN / zero points."""
    n_rows = int(rng.choice([1, 3, 17, 63, 64, 64, 64, 65, 70]))
    rows = []
    for r in range(n_rows):
        m = int(rng.choice([0, 5, 300, 1500, 2083, 2199, 2200, 2300], p=[.05, .05, .1, .2, .4, .1, .05, .05]))
        az = np.sort(rng.uniform(0, 2 * np.pi, m))
        if m and rng.random() < 0.5:
            az = np.round(az / (2 * np.pi / 2200) * rng.choice([1, 2])) * (2 * np.pi / 2200) / rng.choice([1, 2])  # pile-ups in cells
        az = np.where(az > np.pi, az - 2 * np.pi, az)           # file order 0 -> pi -> -pi -> 0
        rad = rng.uniform(2, 60, m)
        incl = np.deg2rad(2.0 - 0.42 * r)
        p = np.stack([rad * np.cos(incl) * np.cos(az), rad * np.cos(incl) * np.sin(az), rad * np.sin(incl), rng.random(m)], axis=1)
        rows.append(p)
    pts = np.concatenate(rows).astype(np.float32) if rows else np.zeros((0, 4), np.float32)
    if len(pts) > 10:
        k = rng.integers(0, len(pts), 3)
        pts[k[0], 0] = np.nan
        pts[k[1], :2] = 0
        pts[k[2], 1] = -0.0
    return pts


def test_randomised_frames_all_stages():
    rng = np.random.default_rng(2024)
    stamps, poses, start, end = _drive(4, (13.0, -0.4, 0.05, 0.5))
    clouds = [_random_cloud(rng) for _ in range(14)]
    conv = kitti.KittiConverter(max_frames=len(clouds), max_points=max(len(c) for c in clouds) + 1)
    frames = []
    for i, pts in enumerate(clouds):
        f = i % 4
        frames.append(dict(points=pts, stages=kitti.ALL_STAGES & ~kitti.FIRINGS, start=start[f], end=end[f],
                           bins=kitti.bin_transforms(stamps, poses, start[f], end[f], poses[f])))
    conv.convert(frames)
    for i, pts in enumerate(clouds):
        f = i % 4
        o = _oracle_chain(pts, stamps, poses, start, end, f)
        res = conv.result(i, pts.shape[0])
        assert res["rows_found"] == o["found"] and res["max_columns"] == o["maxc"] and res["skipped"] == o["skipped"], i
        assert np.array_equal(res["laser_index"], o["laser"]), i
        a, b = res["points"], o["unc"]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(_bits(a)[~np.isnan(a)], _bits(b)[~np.isnan(b)]), i
        assert np.array_equal(res["cell_source"], o["cells"]), i
