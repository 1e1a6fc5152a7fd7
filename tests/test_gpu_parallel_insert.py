"""GPU: the block-parallel insertion kernel (k_insert_par) against the oracle on KITTI-shaped streams that leave its fast
shape in the middle of a call — duplicated firings (two firings in one column), empty firings, a backwards firing, skipped columns,
a rotation wrap inside the call — so that every hand-over to the serial kernel (and back at the next call) is exercised; with the
emission limit inside the parallel part (continuation passes), through the pipelined device entry, and with the option off."""
import numpy as np
import pytest

import util
from continuous_clustering_amd import capi, synth

pytestmark = pytest.mark.gpu


def perturbed_stream(seed, n=2200 * 3 + 300):
    st = synth.make_stream(n, seed=seed, motion=synth.Motion.translate())
    rng = np.random.default_rng(seed)
    order = np.arange(n)
    # duplicates (same firing twice in a row -> second one lands in occupied cells)
    dup = np.sort(rng.choice(np.arange(300, n - 300), 6, replace=False))
    order = np.insert(order, dup, order[dup])
    # a backwards pair
    k = int(rng.integers(1500, n - 1500))
    order[k], order[k + 1] = order[k + 1], order[k]
    # skipped columns
    k2 = int(rng.integers(800, n - 800))
    order = np.delete(order, np.arange(k2, k2 + 37))
    xyz, inten, poses = st.xyz[order].copy(), st.intensity[order].copy(), st.poses[order].copy()
    # empty firings
    for e in rng.choice(np.arange(200, len(order) - 200), 5, replace=False):
        xyz[e] = np.nan
    # a firing whose returns straddle two columns (not the single-column shape)
    e = int(rng.integers(1000, len(order) - 1000))
    xyz[e, ::2] = xyz[e + 1, ::2]
    return synth.Stream(xyz=xyz, intensity=inten, poses=poses, sensor=st.sensor)


@pytest.mark.parametrize("seed,chunks", [(1, [2200]), (2, [997, 64, 1300]), (3, [150, 2200, 63, 700])])
def test_hand_over_between_parallel_and_serial_insertion(seed, chunks, oracle_lib):
    cfg = capi.Config.kitti()
    s = util.run_and_compare(perturbed_stream(seed), cfg, chunks=chunks)
    assert s["published_columns"] > 4000 and s["clusters"] > 50


def test_emission_limit_inside_the_parallel_part(oracle_lib):
    cfg = capi.Config.kitti()
    stream = synth.make_stream(2200 * 3, seed=9, motion=synth.Motion.translate())
    s = util.run_and_compare(stream, cfg, chunks=[2200], engine_setup=lambda e: e.set_option("limit_columns", 700))
    assert s["published_columns"] > 4000


def test_option_off_gives_the_same_result(oracle_lib):
    cfg = capi.Config.kitti()
    stream = perturbed_stream(5)
    a = util.run_and_compare(stream, cfg, chunks=[1100], engine_setup=lambda e: e.set_option("parallel_insert", 0))
    b = util.run_and_compare(stream, cfg, chunks=[1100])
    assert a["events"] == b["events"] and a["published_columns"] == b["published_columns"]


def test_counters_show_the_parallel_kernel_took_the_steady_part(oracle_lib):
    import ctypes as C
    from continuous_clustering_amd import Engine, load_library
    cfg = capi.Config.kitti()
    stream = synth.make_stream(2200 * 3, seed=11, motion=synth.Motion.translate())
    e = Engine(cfg, 64, 1)
    for b in range(3):
        assert e.add_firings(stream.xyz[b * 2200:(b + 1) * 2200], stream.intensity[b * 2200:(b + 1) * 2200], stream.poses[b * 2200:(b + 1) * 2200]) == 0
    L = load_library()
    L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(16, dtype=np.uint64)
    L.cc_engine_debug_counters(e.h, 0, out.ctypes.data)
    assert out[7] == 2 and out[6] == 2 * 2200      # not entered in the first call (ring not started), everything afterwards


def test_device_call_longer_than_the_parallel_kernel_takes(oracle_lib):
    """One cc_engine_add_firings_device call of 5000 firings per stream: k_insert_par considers at most IP_MAXF = 4608 of them, the serial
    kernel the rest; two streams, events off (pipelined mode). State and the columns still in the ring must equal the oracle's."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    cfg = capi.Config.kitti()
    N1, N2 = 2200, 5000
    streams = [synth.make_stream(N1 + N2, seed=31 + s, motion=synth.Motion.translate()) for s in range(2)]
    e = Engine(cfg, 64, 2)
    e.record_events(False)
    e.set_option("limit_columns", 8000)   # one pass per call: a continuation pass would clear what the first pass published (cc_hip.h)
    for lo, hi in ((0, N1), (N1, N1 + N2)):
        xyz = torch.from_numpy(np.stack([st.xyz[lo:hi] for st in streams])).cuda()
        inten = torch.from_numpy(np.stack([st.intensity[lo:hi] for st in streams])).cuda()
        poses = torch.from_numpy(np.stack([st.poses[lo:hi] for st in streams])).cuda()
        torch.cuda.synchronize()
        e.add_firings_device(hi - lo, xyz.data_ptr(), inten.data_ptr(), poses.data_ptr())
        assert e.sync() == 0, e.last_error()
    for s in range(2):
        o = Oracle(cfg, 64)
        assert o.add_firings(streams[s].xyz, streams[s].intensity, streams[s].poses) == 0
        so, se = o.state(), e.state(s)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (s, k)
        hi = se["first_unpublished_global_column_index"] - 1
        util.compare_columns(o.read_published(hi - 3000, hi), e.read_columns(hi - 3000, hi, stream=s), hi - 3000, mirror=False)


def _fused_batches(e):
    import ctypes as C
    from continuous_clustering_amd import load_library
    L = load_library()
    L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(16, dtype=np.uint64)
    L.cc_engine_debug_counters(e.h, 0, out.ctypes.data)
    return int(out[4])


@pytest.mark.parametrize("chunks", [[2200], [700, 64, 1500], [100, 2100]])
def test_fused_segmentation_is_taken_and_changes_nothing(chunks, oracle_lib):
    """Round 4: k_insert_par also does the per-cell part of the segmentation of the columns it fills and closes a batch it took completely as
    FUSED (no k_table / k_seg_pre for the stream). On a steady stream every batch but the first must be closed that way (debug counter 4), on a
    stream that keeps leaving the fast shape some are and some are not — and with the option off none: the oracle's results in all three."""
    cfg = capi.Config.kitti()
    stream = synth.make_stream(2200 * 3 + 300, seed=21, motion=synth.Motion.translate())
    box = {}
    util.run_and_compare(stream, cfg, chunks=chunks, engine_setup=lambda e: box.__setitem__("e", e))
    calls = 0
    f = i = 0
    while f < stream.n_firings:
        f += min(chunks[i % len(chunks)], stream.n_firings - f)
        i += 1
        calls += 1
    big = sum(1 for k in range(calls) if min(chunks[k % len(chunks)], 10 ** 9) >= 64)
    assert _fused_batches(box["e"]) >= big - 2, (_fused_batches(box["e"]), big)   # (all calls of >= 64 firings but the first one or two)
    util.run_and_compare(stream, cfg, chunks=chunks, engine_setup=lambda e: (e.set_option("fuse_front", 0), box.__setitem__("off", e)))
    assert _fused_batches(box["off"]) == 0
    rough = perturbed_stream(7)
    util.run_and_compare(rough, cfg, chunks=[1100], engine_setup=lambda e: box.__setitem__("r", e))
    assert 0 < _fused_batches(box["r"]) < 7


def test_fused_segmentation_with_columns_no_firing_fills(oracle_lib):
    """A sensor that skips every third column (firings 1.5 columns apart): the columns in between are segmented as columns without returns by
    the wavefront of the firing in front of them (stage_gap), still inside the fused kernel."""
    cfg = capi.Config.kitti()
    cfg.num_columns = 720
    sensor = synth.SensorModel(num_rows=64, num_columns=480)   # 480 firings per rotation into 720 columns
    stream = synth.make_stream(480 * 4 + 100, seed=33, sensor=sensor, motion=synth.Motion.translate())
    box = {}
    s = util.run_and_compare(stream, cfg, chunks=[480, 200], engine_setup=lambda e: box.__setitem__("e", e))
    assert s["published_columns"] > 1500 and _fused_batches(box["e"]) >= 5
