"""GPU: parity across the 10-rotation ring (cc.cpp:17 `ring_buffer_max_columns = num_columns * 10`).

Streams of >= 12 rotations re-use every local column at least once (cc.cpp:178), so these cases compare with the oracle what the
shorter parity cases never reach: insertion into a cell whose previous tenant was cleared (cc.cpp:188-206 after :1094-1145), the
"not cleared" check of segmentation (:320-345), the one-rotation retention of published columns (:1077-1091), the deferred clearing of
the engine (`StreamState::clear_done / clear_allowed`), the `G_f - ring_cols >= clear_done` guard of k_insert_par, and the re-use of
tree planes / root cells after a wrap — through every entry point: the host path in chunks of {1, 97, columns}, the pipelined device
path with events off, parallel insertion on / off, one- and two-wave association, a 128-row sensor with per-laser azimuth offsets.
One case deliberately overruns the ring (no finished-cluster check for > 10 rotations) and expects the reference's
"This column is not cleared" error class (CC_ERR_RING_OVERRUN) from both sides at the same firing."""
import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,chunks", [
    ("w_s64_240x13", [240]),
    ("w_s64_240x13", [97]),
    ("w_s64_240x13", [1]),
    ("w_s64_360x12_turn", [360]),
    ("w_s64_360x12_turn", [700, 7, 333]),
    ("w_s64_ring_wall_240x12", [97]),
    ("w_s128_offsets_340x12", [340]),
    ("w_s128_offsets_340x12", [97]),
    ("w_s32_256x12", [256]),
])
def test_ring_wrap_host_path(name, chunks, oracle_lib):
    stream, cfg, tf = cases.build_case(name)
    s = util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=tf)
    cols = stream.sensor.num_columns
    assert s["published_columns"] > 10 * cols, "the ring was not wrapped"
    assert s["engine_state"]["ring_buffer_start_global_column_index"] > 9 * cols
    assert s["clusters"] > 30


@pytest.mark.parametrize("name,option,value", [
    ("w_s64_240x13", "parallel_insert", 0),
    ("w_s64_240x13", "assoc_waves", 1),
    ("w_s64_240x13", "assoc_waves", 3),
    ("w_s64_ring_wall_240x12", "assoc_waves", 1),
    ("w_s64_ring_wall_240x12", "assoc_waves", 3),
    ("w_s128_offsets_340x12", "assoc_waves", 3),
    ("w_s64_ring_wall_240x12", "lds_tree_limit", 3),
    ("w_s128_offsets_340x12", "assoc_waves", 1),
])
def test_ring_wrap_kernel_variants(name, option, value, oracle_lib):
    stream, cfg, tf = cases.build_case(name)
    s = util.run_and_compare(stream, cfg, chunks=[stream.sensor.num_columns, 61], robot_tf=tf,
                             engine_setup=lambda e: e.set_option(option, value))
    assert s["published_columns"] > 10 * stream.sensor.num_columns


@pytest.mark.parametrize("nth", [2, 7])
def test_ring_wrap_every_nth_column(nth, oracle_lib):
    stream, cfg, tf = cases.build_case("w_s64_240x13")
    cfg = cfg.copy()
    cfg.cluster_point_trees_every_nth_column = nth
    s = util.run_and_compare(stream, cfg, chunks=[240, 33], robot_tf=tf)
    assert s["published_columns"] > 10 * 240


@pytest.mark.parametrize("pipeline,sub_batch,F", [(1, 0, 240), (2, 0, 240), (1, 100, 480), (0, 0, 240), (1, 0, 24)])
def test_ring_wrap_pipelined_device_path(pipeline, sub_batch, F, oracle_lib):
    """Events off: batches overlap on the engine's chains of HIP streams while a stream's ring start moves and clearing runs behind it;
    after 13 rotations every stream must be in the oracle's state and hold the oracle's columns in the retained part of the ring."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    cols, rot = 240, 13
    sen = synth.SensorModel(num_rows=64, num_columns=cols)
    cfg = capi.Config.kitti()
    cfg.num_columns = cols
    S = 6
    motions = [synth.Motion.static(), synth.Motion.translate(), synth.Motion.turn()]
    scenes = [synth.SceneModel(), synth.SceneModel(n_objects=25, wall_radius=14.0, wall_gaps_deg=(), object_range=(3.0, 11.0))]
    streams = [synth.make_stream(cols * rot, seed=700 + s, sensor=sen, motion=motions[s % 3], scene=scenes[s % 2]) for s in range(S)]
    NB = cols * rot // F
    e = Engine(cfg, 64, S)
    e.record_events(False)
    e.set_option("pipeline", pipeline)
    e.set_option("sub_batch", sub_batch)
    # batch-major device buffers that stay alive until the engine has consumed them (the calls are asynchronous)
    xyz = torch.from_numpy(np.stack([st.xyz[:NB * F].reshape(NB, F, 64, 3) for st in streams], axis=1)).cuda()
    inten = torch.from_numpy(np.stack([st.intensity[:NB * F].reshape(NB, F, 64) for st in streams], axis=1)).cuda()
    poses = torch.from_numpy(np.stack([st.poses[:NB * F].reshape(NB, F, 12) for st in streams], axis=1)).cuda()
    torch.cuda.synchronize()
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0, e.last_error()
    for s in range(S):
        o = Oracle(cfg, 64)
        assert o.add_firings(streams[s].xyz[:NB * F], streams[s].intensity[:NB * F], streams[s].poses[:NB * F]) == 0
        so, se = o.state(), e.state(s)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (s, k, so[k], se[k])
        assert se["ring_buffer_start_global_column_index"] > 9 * cols
        hi = se["first_unpublished_global_column_index"] - 1
        lo = se["ring_buffer_start_global_column_index"]
        util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s), lo, mirror=False)


@pytest.mark.parametrize("sync_every_call", [0, 1])
def test_ring_wrap_with_streams_that_leave_the_steady_shape(sync_every_call, oracle_lib):
    """Few streams (k_insert_par deals a stream's firings to several blocks, which share the deferred clearing of the ring) of which some
    keep leaving the steady one-column-per-firing shape (whole calls of empty firings, repeated firings): 24 rotations through the
    10-rotation ring. A stream that is not steady still has its share of the clearing done by every block (round 5: block 0 published
    clear_done too early and "This column is not cleared" came up one pass over the ring later)."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    cols, rot, F = 240, 24, 240
    sen = synth.SensorModel(num_rows=64, num_columns=cols)
    cfg = capi.Config.kitti()
    cfg.num_columns = cols
    S = 7
    motions = [synth.Motion.static(), synth.Motion.translate(), synth.Motion.turn()]
    streams = [synth.make_stream(cols * rot, seed=900 + s, sensor=sen, motion=motions[s % 3], scene=synth.SceneModel()) for s in range(S)]
    NB = rot
    X = [st.xyz[:NB * F].copy() for st in streams]
    rng = np.random.default_rng(5)
    for s in (1, 3, 4, 6):
        for b in rng.choice(np.arange(2, NB), size=7, replace=False):
            if (s + b) % 2:
                X[s][b * F:(b + 1) * F] = np.nan                                         # a call of empty firings
            else:
                X[s][b * F + 100:b * F + 140] = X[s][b * F + 99:b * F + 100]             # one firing 41 times: the column does not advance
    e = Engine(cfg, 64, S)
    e.record_events(False)
    xyz = torch.from_numpy(np.stack([x.reshape(NB, F, 64, 3) for x in X], axis=1)).cuda()
    inten = torch.from_numpy(np.stack([st.intensity[:NB * F].reshape(NB, F, 64) for st in streams], axis=1)).cuda()
    poses = torch.from_numpy(np.stack([st.poses[:NB * F].reshape(NB, F, 12) for st in streams], axis=1)).cuda()
    torch.cuda.synchronize()
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
        if sync_every_call:
            assert e.sync() == 0, e.last_error()
    assert e.sync() == 0, e.last_error()
    for s in range(S):
        o = Oracle(cfg, 64)
        assert o.add_firings(X[s], streams[s].intensity[:NB * F], streams[s].poses[:NB * F]) == 0
        so, se = o.state(), e.state(s)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (s, k, so[k], se[k])
        hi = se["first_unpublished_global_column_index"] - 1
        lo = se["ring_buffer_start_global_column_index"]
        assert lo > 10 * cols
        util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s), lo, mirror=False)


@pytest.mark.parametrize("chunks", [[240], [1], [977]])
def test_deliberate_ring_overrun(chunks, oracle_lib):
    """cluster_point_trees_every_nth_column larger than the ring: nothing is ever published or cleared (cc.cpp:841-842), the eleventh
    rotation runs into its own tail and segmentation must refuse the column exactly where the reference throws (cc.cpp:320-345)."""
    stream, cfg, tf = cases.build_case("w_s64_240x13")
    cfg = cfg.copy()
    cfg.cluster_point_trees_every_nth_column = 100000
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    o = Oracle(cfg, 64)
    orc = o.add_firings(stream.xyz, stream.intensity, stream.poses)
    assert orc == capi.CC_ERR_RING_OVERRUN, o.last_error()
    import re
    stale, col, ring = (int(v) for v in re.search(r": (-?\d+), (-?\d+), (\d+)", o.last_error()).groups())
    assert ring == 2400 and col >= 2400 and stale == col - ring
    e = Engine(cfg, 64)
    rc, f, i = 0, 0, 0
    while f < stream.n_firings and rc == 0:
        m = min(chunks[i % len(chunks)], stream.n_firings - f)
        rc = e.add_firings(stream.xyz[f:f + m], stream.intensity[f:f + m], stream.poses[f:f + m])
        f += m
        i += 1
    assert rc == capi.CC_ERR_RING_OVERRUN, (rc, e.last_error())
    se = e.state()
    # the reference's exception text carries (stale global column found in the cell, column being segmented, ring size)
    assert (se["error_a"], se["error_b"]) == (stale, col), (se, o.last_error())
    assert "not cleared" in e.last_error() and f"{stale}, {col}, {ring}" in e.last_error()


def test_reshape_engine_between_resets(oracle_lib):
    """reset(num_rows) with a different row count (cc.cpp:11-27 re-sizes the range image): 32 -> 64 -> 128 rows on one engine, fed by
    single-firing calls (pinned small-call staging is per row count) and by whole rotations."""
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    cfg = capi.Config.kitti()
    cfg.num_columns = 360
    e = Engine(cfg, 32)
    for rows, n, chunk in ((32, 500, 1), (64, 500, 1), (128, 500, 1), (64, 800, 360), (32, 420, 7)):
        sen = synth.SensorModel(num_rows=rows, num_columns=360)
        st = synth.make_stream(n, seed=rows + n, sensor=sen, motion=synth.Motion.translate())
        e.reset(rows)
        e.set_robot_from_sensor(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64))
        o = Oracle(cfg, rows)
        assert o.add_firings(st.xyz, st.intensity, st.poses) == 0
        eo = o.drain_events()
        got = []
        for f in range(0, n, chunk):
            assert e.add_firings(st.xyz[f:f + chunk], st.intensity[f:f + chunk], st.poses[f:f + chunk]) == 0, e.last_error()
            got.append(e.drain_events())
        ee = np.concatenate(got)
        assert len(ee) == len(eo)
        for fld in ("type", "a", "b", "c", "d", "column"):
            assert np.array_equal(ee[fld], eo[fld]), (rows, fld)
        so, se = o.state(), e.state()
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (rows, k)
        hi = se["first_unpublished_global_column_index"] - 1
        lo = se["ring_buffer_start_global_column_index"]
        util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi), lo)
