"""GPU: the HIP path reproduces the committed golden fixtures (no oracle involved at run time)."""
import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import capi
from test_oracle_golden import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", cases.GOLDEN_CASES)
def test_engine_matches_golden(name):
    from continuous_clustering_amd import Engine, IDENTITY_TF
    z, cfg, tf = load_golden(name)
    R = int(z["num_rows"])
    e = Engine(cfg, R, 1, 0, IDENTITY_TF if tf is None else tf)
    gold_ev = z["events"]
    out_fields = [k[4:] for k in z.files if k.startswith("out_")]
    first = int(z["first_column"])
    n = z["xyz"].shape[0]
    pos = 0
    chunk = cfg.num_columns
    for f in range(0, n, chunk):
        assert e.add_firings(z["xyz"][f:f + chunk], z["intensity"][f:f + chunk], z["poses"][f:f + chunk]) == 0
        ev = e.drain_events()
        ref = gold_ev[pos:pos + len(ev)]
        assert len(ref) == len(ev)
        for fld in ("type", "a", "b", "c", "d", "column"):
            assert np.array_equal(ev[fld], ref[fld]), fld
        pos += len(ev)
        pub = ev[(ev["type"] == capi.EV_PUBLISH_COLUMNS) & (ev["b"] >= ev["a"])]
        if len(pub):
            lo, hi = int(pub["a"].min()), int(pub["b"].max())
            cols = e.read_columns(lo, hi, fields=out_fields)
            for fld in out_fields:
                g = z["out_" + fld][lo - first:hi - first + 1]
                a = cols[fld]
                if a.dtype.kind == "f":
                    util.assert_float_equal(fld, a, g)
                else:
                    assert np.array_equal(a.astype(np.int64), g.astype(np.int64)), fld
    assert pos == len(gold_ev)
    st = e.state()
    assert [st[k] for k in util.STATE_FIELDS] == list(z["state"])


def test_kitti_replay_and_labels_match_committed_vectors():
    """The HIP KITTI converter and the ground-truth label kernels against tests/golden/g_kitti_replay.npz / g_gt_labels.npz."""
    import os
    from continuous_clustering_amd import evaluation, kitti
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "g_kitti_replay.npz"))
    f = int(g["frame"])
    bins = kitti.bin_transforms(g["stamps"], g["poses"], g["start"][f], g["end"][f], g["poses"][f])
    assert np.array_equal(bins.view(np.uint64), g["bins"].view(np.uint64))
    conv = kitti.KittiConverter(max_frames=1)
    pts = g["points"]
    for shift, rc, src in ((True, g["cell_rc"], g["cell_src"]), (False, g["plain_rc"], g["plain_src"])):
        stages = kitti.RECOVER_ROWS | kitti.UNDO_EGO_MOTION | kitti.RANGE_IMAGE | (kitti.SHIFT_OCCUPIED if shift else 0)
        conv.convert([dict(points=pts, stages=stages, start=g["start"][f], end=g["end"][f], bins=bins)])
        r = conv.result(0, pts.shape[0])
        assert np.array_equal(r["laser_index"], g["laser"]) and r["rows_found"] == int(g["rows_found"]) and r["skipped"] == int(g["skipped"])
        a, b = r["points"], g["uncorrected"]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)])
        cells = np.full((64, 2200), -1, dtype=np.int32)
        cells[rc[:, 0], rc[:, 1]] = src
        assert np.array_equal(r["cell_source"], cells)
    fs, fp = kitti.firing_stamps_and_poses(g["stamps"], g["poses"], g["start"][f], g["end"][f])
    assert np.array_equal(fs, g["firing_stamps"]) and np.array_equal(fp[1099].view(np.uint64), g["firing_pose_1099"].view(np.uint64))
    gl = np.load(os.path.join(gold, "g_gt_labels.npz"))
    assert np.array_equal(evaluation.generate_euclidean_labels(gl["points"], gl["semantic"], gl["instance"]), gl["labels"])
