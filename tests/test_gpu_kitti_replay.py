"""GPU: a KITTI-format sequence on disk replayed the way the reference's no-ROS harness does it (src/tools/kitti_demo.cpp:229-420),
through the C++ mirrors KittiLoader (per-point steps on the GPU) -> ContinuousClustering (HIP hot path) -> cc_eval_frame, against the
same walk done with the oracle: oracle loader -> oracle clustering -> oracle label compare. Per-frame evaluation records must be equal
(integers exact, entropies bit-equal: both sides sum in std::map order with the host's std::log)."""
import os
import subprocess

import numpy as np
import pytest

from continuous_clustering_amd import capi, kitti
from continuous_clustering_amd.evaluation import FrameScatter
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "tests", "cpp", "kitti_replay_demo")
T0 = 1_700_000_000_000_000_000


def expected_records(seq_dir, sequence, n_frames):
    times = [float(l) for l in open(os.path.join(seq_dir, "times.txt"))]
    stamps = np.array([T0 + int(t * 1000000000) for t in times], dtype=np.uint64)
    calib = [l.split() for l in open(os.path.join(seq_dir, "calib.txt"))]
    tr = np.array([float(v) for v in calib[4][1:13]])
    poses = np.stack([orc.kitti_pose_from_line(np.array([float(v) for v in l.split()]), tr) for l in open(os.path.join(seq_dir, "poses.txt"))])
    start, end = orc.kitti_start_end_stamps(stamps)
    pts, sem, eu = [], [], []
    for f in range(n_frames):
        pts.append(np.fromfile(os.path.join(seq_dir, "velodyne", f"{f:06d}.bin"), dtype=np.float32).reshape(-1, 4))
        sem.append(np.fromfile(os.path.join(seq_dir, "labels", f"{f:06d}.label"), dtype=np.uint16).reshape(-1, 2)[:, 0].copy())
        eu.append(np.fromfile(os.path.join(seq_dir, "labels_euclidean_clustering", f"{f:06d}.label"), dtype=np.uint16).astype(np.uint32))
    scatter = FrameScatter(sequence, [p.shape[0] for p in pts], sem, eu, evaluate=orc.eval_frame)
    o = orc.Oracle(capi.Config.kitti(), 64)
    uniques = []
    columns = 0
    published_to = -1
    for f in range(n_frames):
        laser, _, _, _ = orc.kitti_recover_laser_indices(pts[f])
        unc = orc.kitti_undo_ego_motion(pts[f], start[f], end[f], poses[f], stamps, poses)
        cells, _ = orc.kitti_generate_range_image(unc, laser, True)
        xyz, inten, unique, fstamps = orc.kitti_make_firings(unc, cells, start[f], end[f], sequence, f)
        uniques.append(unique)
        fposes = np.stack([orc.kitti_interpolate(stamps, poses, int(s)) for s in fstamps])
        assert o.add_firings(xyz, inten, fposes) == 0, o.last_error()
        base, hi = o.published_range()
        lo = max(base, published_to + 1)
        if hi >= lo:
            cols = o.read_published(lo, hi, fields=("source_firing", "ground_point_label", "id"))
            src = cols["source_firing"]
            u = np.full(src.shape, np.uint64(2 ** 64 - 1), dtype=np.uint64)
            has = src >= 0
            allu = np.concatenate(uniques)                                   # [firing][row]
            rows = np.broadcast_to(np.arange(64), src.shape)
            u[has] = allu[src[has], rows[has]]
            scatter.add_columns(u, cols["ground_point_label"], cols["id"])
            columns += hi - lo + 1
            published_to = hi
    return scatter, columns, o.state()


def test_sequence_replay_matches_oracle(tmp_path, oracle_lib):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "continuous_clustering_amd", "csrc")], stdout=subprocess.DEVNULL)
    n_frames, sequence = 4, 3
    seq_dir, sizes = kitti.write_synthetic_sequence(str(tmp_path), sequence, n_frames, seed=11, motion=(9.0, 0.3, 0.0, 0.2))
    scatter, columns, ostate = expected_records(seq_dir, sequence, n_frames)
    assert len(scatter.records) == n_frames - 1
    for mode in ([], ["--one-pass", "--rccl-gather"]):
        out = subprocess.run([DEMO, str(tmp_path), str(sequence), "--fixed-start-stamp", str(T0)] + mode, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr
        frames = [l.split() for l in out.stdout.splitlines() if l.startswith("FRAME")]
        if "--rccl-gather" in mode:  # the records once more, after cc_eval_gather_records (one ncclAllGather over RCCL) from the C++ harness
            gathered = [l.split()[1:] for l in out.stdout.splitlines() if l.startswith("GATHERED")]
            assert gathered == [f[1:9] for f in frames]
        summary = [l.split() for l in out.stdout.splitlines() if l.startswith("SUMMARY")][0]
        # the harness also evaluates the last frame that received points at the end of the sequence (kitti_demo.cpp:417-419)
        assert len(frames) == n_frames
        for rec, line in zip(scatter.records, frames):
            assert int(line[1]) == rec[0] and int(line[2]) == rec[1]
            got = np.array([float(v) for v in line[3:9]])
            want = np.array(rec[2:8])
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (rec[1], got, want)
            assert got[:4].sum() > 0.5 * sizes[rec[1]]
        # the reference never flushes its last columns; the mirror's flush() does not publish more either
        assert int(summary[2]) == columns
        assert int(summary[3]) == ostate["clusters_finished"] or int(summary[3]) <= ostate["clusters_finished"]
    # the last record: evaluate the oracle's view of the last started frame the same way
    scatter.finish()
    got = np.array([float(v) for v in frames[-1][3:9]])
    assert np.array_equal(got.view(np.uint64), np.array(scatter.records[-1][2:8]).view(np.uint64))


def test_online_ground_truth_labels(tmp_path, oracle_lib):
    """Without labels_euclidean_clustering/ the harness generates the ground-truth cluster labels per frame on the GPU
    (kitti_demo.cpp:337-346) and --write-gt-labels stores them like gt_label_generator_tool.cpp:63-70; both must equal the oracle's."""
    import shutil
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "continuous_clustering_amd", "csrc")], stdout=subprocess.DEVNULL)
    n_frames, sequence = 3, 5
    seq_dir, sizes = kitti.write_synthetic_sequence(str(tmp_path), sequence, n_frames, seed=23, motion=(7.0, 0.0, 0.0, 0.1))
    shutil.rmtree(os.path.join(seq_dir, "labels_euclidean_clustering"))
    os.makedirs(os.path.join(seq_dir, "labels_euclidean_clustering"))
    for f in range(n_frames):  # the oracle's labels take the place of the downloaded ones for the expected records
        pts = np.fromfile(os.path.join(seq_dir, "velodyne", f"{f:06d}.bin"), dtype=np.float32).reshape(-1, 4)
        lab = np.fromfile(os.path.join(seq_dir, "labels", f"{f:06d}.label"), dtype=np.uint16).reshape(-1, 2)
        want, nc = orc.generate_euclidean_labels(pts, lab[:, 0], lab[:, 1])
        assert nc > 3
        want.tofile(os.path.join(seq_dir, "labels_euclidean_clustering", f"{f:06d}.label"))
    scatter, _, _ = expected_records(seq_dir, sequence, n_frames)
    scatter.finish()
    os.rename(os.path.join(seq_dir, "labels_euclidean_clustering"), os.path.join(seq_dir, "expected_gt"))
    out = subprocess.run([DEMO, str(tmp_path), str(sequence), "--fixed-start-stamp", str(T0), "--one-pass", "--write-gt-labels"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    frames = [l.split() for l in out.stdout.splitlines() if l.startswith("FRAME")]
    assert len(frames) == n_frames
    for rec, line in zip(scatter.records, frames):
        got = np.array([float(v) for v in line[3:9]])
        assert np.array_equal(got.view(np.uint64), np.array(rec[2:8]).view(np.uint64)), rec[1]
    for f in range(n_frames):
        a = np.fromfile(os.path.join(seq_dir, "labels_euclidean_clustering_generated", f"{f:06d}.label"), dtype=np.uint16)
        b = np.fromfile(os.path.join(seq_dir, "expected_gt", f"{f:06d}.label"), dtype=np.uint16)
        assert np.array_equal(a, b)


def test_concurrent_sequences_match_per_sequence_oracle(tmp_path, oracle_lib):
    """BASELINE configs[4] shape on one GPU: sequences of different lengths replayed concurrently as the streams of one engine
    (continuous_clustering_amd.replay), frames converted on the GPU and handed to the engine in HBM; every sequence's per-frame evaluation
    records must equal the single-sequence oracle walk. Rank sharding of the same function: rank r of 2 sees only its sequences."""
    from continuous_clustering_amd import replay
    lengths = {2: 3, 4: 2, 7: 4}
    want = {}
    for k, (seq, nf) in enumerate(lengths.items()):
        seq_dir, _ = kitti.write_synthetic_sequence(str(tmp_path), seq, nf, seed=100 + 10 * k, motion=(6.0 + k, 0.1 * k, 0.0, 0.1 * (k + 1)))
        sc, _, _ = expected_records(seq_dir, seq, nf)
        sc.finish()
        want[seq] = np.array(sc.records)
    records, totals = replay.replay(str(tmp_path), list(lengths))
    assert totals["streams"] == 3 and totals["frames"] == sum(lengths.values())
    got = evaluation_sorted(records)
    allwant = np.concatenate([want[s] for s in sorted(want)])
    assert got.shape == allwant.shape
    assert np.array_equal(got.view(np.uint64), allwant.view(np.uint64))
    # sharding: the union of what two ranks produce is the same set of records
    r0, _ = replay.replay(str(tmp_path), list(lengths), rank=0, world=2)
    r1, _ = replay.replay(str(tmp_path), list(lengths), rank=1, world=2)
    assert {int(r[0]) for r in r0} == {2, 7} and {int(r[0]) for r in r1} == {4}
    both = evaluation_sorted(list(r0) + list(r1))
    assert np.array_equal(both.view(np.uint64), allwant.view(np.uint64))


def test_concurrent_sequences_longer_than_the_ring(tmp_path, oracle_lib):
    """Sequences of more rotations than the ring holds (10), next to one that ends early and goes on with empty rotations: the deferred clearing
    of the ring is shared by the blocks k_insert_par deals a stream's firings to (few streams), also for streams that are not in the steady
    shape. (Round 5: block 0 published the new clear_done before the other blocks had read the old one; their share of the columns stayed
    uncleared and the segmentation of the next pass over the ring found them: "This column is not cleared" in bench.py's replay leg.)"""
    from continuous_clustering_amd import replay
    lengths = {1: 13, 3: 2, 6: 12, 9: 5}
    want = {}
    for k, (seq, nf) in enumerate(lengths.items()):
        seq_dir, _ = kitti.write_synthetic_sequence(str(tmp_path), seq, nf, seed=300 + 7 * k, motion=(6.5 + k, 0.05 * k, 0.0, 0.1))
        if seq in (3, 6):
            sc, _, _ = expected_records(seq_dir, seq, nf)
            sc.finish()
            want[seq] = np.array(sc.records)
    for _ in range(2):
        records, totals = replay.replay(str(tmp_path), list(lengths))
        assert totals["streams"] == 4 and totals["frames"] == sum(lengths.values())
        for seq, w in want.items():
            got = np.array(sorted([r for r in records if int(r[0]) == seq], key=lambda r: r[1]))
            assert got.shape == w.shape
            assert np.array_equal(got.view(np.uint64), w.view(np.uint64))


def evaluation_sorted(records):
    from continuous_clustering_amd.evaluation import gather_records
    return gather_records(records)
