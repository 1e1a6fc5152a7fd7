"""GPU: the HIP path, called through the C-ABI (libcc_hip.so), against the CPU oracle on the same seeded inputs.
Bit-exact bar: events (reference callback order), ground / debug labels, ignore flags, geometry bit patterns, tree
roots, raw cluster ids (and therefore the canonical partition)."""
import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import capi, synth

pytestmark = pytest.mark.gpu

CHUNKS = {
    "s64_static": [720, 1, 33, 500],
    "s64_translate": [300],
    "s64_full_2200": [2200],
    "s128_full_1700": [1700, 3],
    "s64_forced_finish_ring": [97],
}


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_engine_matches_oracle(name, oracle_lib):
    stream, cfg, tf = cases.build_case(name)
    summary = util.run_and_compare(stream, cfg, chunks=CHUNKS.get(name, [stream.sensor.num_columns]), robot_tf=tf)
    assert summary["published_columns"] > stream.sensor.num_columns
    assert summary["clusters"] >= 3
    es = summary["engine_state"]
    # (error_a, error_b) double as (clusters exceeding one rotation, serially replayed columns) when there is no error
    assert es["error_a"] == summary["oracle_state"]["error_a"]
    if name in ("s64_forced_finish_ring", "s64_no_early_stop"):
        assert es["error_b"] > 0, "the exact serial association path was expected to run"


def test_single_firing_calls(oracle_lib):
    """addFiring one firing at a time (the reference's calling pattern)."""
    stream, cfg, tf = cases.build_case("g_s64_translate")
    util.run_and_compare(stream, cfg, chunks=[1], robot_tf=tf)


def test_reset_and_reuse(oracle_lib):
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    stream, cfg, tf = cases.build_case("g_s64_translate")
    e = Engine(cfg, 64)
    o = Oracle(cfg, 64)
    for rep in range(2):
        assert e.add_firings(stream.xyz[:500], stream.intensity[:500], stream.poses[:500]) == 0
        assert o.add_firings(stream.xyz[:500], stream.intensity[:500], stream.poses[:500]) == 0
        ee, eo = e.drain_events(), o.drain_events()
        assert len(ee) == len(eo) and all(np.array_equal(ee[f], eo[f]) for f in ("type", "a", "b", "c", "d", "column"))
        e.reset()
        o.reset()
        # reset() drops the robot transform (cc.cpp:39): the next segmented column must fail until it is set again
        assert e.add_firings(stream.xyz[:50], stream.intensity[:50], stream.poses[:50]) == capi.CC_ERR_NO_ROBOT_TRANSFORM
        assert o.add_firings(stream.xyz[:50], stream.intensity[:50], stream.poses[:50]) == capi.CC_ERR_NO_ROBOT_TRANSFORM
        e.reset()
        o.reset()
        e.set_robot_from_sensor(capi_identity())
        o.set_robot_from_sensor(capi_identity())


def capi_identity():
    return np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64)


def test_bad_start_sets_reset_required(oracle_lib):
    from continuous_clustering_amd import Engine
    stream, cfg, tf = cases.build_case("g_s64_translate")
    bad = stream.xyz[:1].copy()
    bad[0, :32, 1] = np.abs(bad[0, :32, 1]) + 0.1
    bad[0, 32:, 1] = -np.abs(bad[0, 32:, 1]) - 0.1
    bad[0, :, 0] = -np.abs(bad[0, :, 0]) - 1.0
    e = Engine(cfg, 64)
    assert e.add_firings(bad, stream.intensity[:1], stream.poses[:1]) == 0
    assert e.state()["reset_required"] == 1  # cc.cpp:252-261
    cfg2 = cfg.copy()
    cfg2.num_columns = 400
    e2 = Engine(cfg, 64)
    e2.set_config(cfg2)
    assert e2.state()["reset_required"] == 1  # cc.cpp:73-74


@pytest.mark.parametrize("name,limit", [("s64_no_supplement_no_incl_ignore", 4), ("s32_small_sensor", 2), ("s64_translate", 1)])
def test_lds_tree_pool_overflow_continues_in_global_memory(name, limit, oracle_lib):
    """When a stream has more unfinished point trees than fit the LDS pool, k_assoc_lds hands the stream to the
    global-memory kernel mid-batch; results must not change. The pool size is lowered to force the hand-over."""
    stream, cfg, tf = cases.build_case(name)
    util.run_and_compare(stream, cfg, chunks=[stream.sensor.num_columns // 2], robot_tf=tf,
                         engine_setup=lambda e: e.set_option("lds_tree_limit", limit))


@pytest.mark.parametrize("name", ["s64_translate", "s64_forced_finish_ring", "s64_no_early_stop", "s128_full_1700"])
def test_single_wave_association_kernel(name, oracle_lib):
    """k_assoc_lds (one wavefront per stream; the engine uses it for cluster_point_trees_every_nth_column != 1) stays exact:
    option assoc_waves = 1 selects it for every configuration."""
    stream, cfg, tf = cases.build_case(name)
    summary = util.run_and_compare(stream, cfg, chunks=CHUNKS.get(name, [stream.sensor.num_columns]), robot_tf=tf,
                                   engine_setup=lambda e: (e.set_option("assoc_waves", 1), e.set_option("assoc_batch", 0)))
    assert summary["clusters"] >= 3


@pytest.mark.parametrize("waves", [0, 3, 4])
def test_cooperating_wave_kernels_roll_back_speculation(waves, oracle_lib):
    """Small calls make the resolving wavefront of k_assoc3 run ahead of freshly finished trees on every launch; the serially
    replayed columns must show up (error_b doubles as their count) and nothing may change. assoc_waves: 3 / 4 = k_assoc3
    without / with its links wavefront, 0 = the default choice."""
    stream, cfg, tf = cases.build_case("s64_forced_finish_ring")
    summary = util.run_and_compare(stream, cfg, chunks=[37, 5, 211], robot_tf=tf,
                                   engine_setup=lambda e: (e.set_option("assoc_waves", waves), e.set_option("assoc_batch", 0)))
    assert summary["engine_state"]["error_b"] > 0


def test_segmentation_look_back_beyond_the_lds_ring(oracle_lib):
    """k_seg_scan keeps the azimuth-plane distance of 16 rows per column in LDS; the downward fix-up (cc.cpp:513-535) of this case walks over
    up to 62 ground cells, so most of its reads come from the staging plane."""
    stream, cfg, tf = cases.build_case("s64_deep_lookback")
    oracle, rc = util.run_oracle(stream, cfg, tf)
    dbg = np.asarray(oracle.read_published(0, 400)["debug_ground_point_label"])
    runs = [max(len(r) for r in "".join("x" if v == 32 else " " for v in col).split()) if (col == 32).any() else 0 for col in dbg]  # 32 = dark red
    assert max(runs) > 40 and sum(r > 16 for r in runs) > 100
    util.run_and_compare(stream, cfg, chunks=[360, 97], robot_tf=tf)


@pytest.mark.parametrize("waves", [1, 3, 4])
@pytest.mark.parametrize("chunks", [[360, 97, 82, 231], [360, 97, 82, 100, 131]])
def test_serial_kernels_column_unresolved_in_the_middle_of_a_group(waves, chunks, oracle_lib):
    """The front wave of k_assoc3 stops at a column it cannot resolve (a point attaches to a tree that finished before the launch:
    the long ground runs of this case do that when a launch starts at the right column); the back wave waits for whole groups of columns.
    Rounds 1 and 2 let it wait for ever when that column was not the last of its group (spin limit -> error -772) — found in round 3 when
    k_assocb's hand-over made the serial kernel start at such columns."""
    stream, cfg, tf = cases.build_case("s64_deep_lookback")
    for batch in (0, 1):
        util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=tf,
                             engine_setup=lambda e: (e.set_option("assoc_waves", waves), e.set_option("assoc_batch", batch)))


@pytest.mark.parametrize("name,waves", [("s64_translate", 3), ("s64_no_early_stop", 3), ("s128_full_1700", 3), ("j_s64_jitter_wide", 3),
                                        ("s128_offsets", 1), ("s64_dropouts", 4), ("s128_full_1700", 4)])
def test_association_kernel_selection(name, waves, oracle_lib):
    """Every serial association kernel the option can select reproduces the oracle — alone (assoc_batch = 0) and behind the batch-parallel
    kernel, which hands them whatever it cannot take."""
    stream, cfg, tf = cases.build_case(name)
    for batch in (0, 1):
        util.run_and_compare(stream, cfg, chunks=CHUNKS.get(name, [stream.sensor.num_columns, 97]), robot_tf=tf,
                             engine_setup=lambda e: (e.set_option("assoc_waves", waves), e.set_option("assoc_batch", batch)))


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_serial_association_only(name, oracle_lib):
    """assoc_batch = 0: the serial kernels (k_assoc3 by default) alone, as in rounds 1 and 2 — the exact fallback of the batch-parallel
    kernel has to stay exact on everything."""
    stream, cfg, tf = cases.build_case(name)
    box = {}
    summary = util.run_and_compare(stream, cfg, chunks=CHUNKS.get(name, [stream.sensor.num_columns]), robot_tf=tf,
                                   engine_setup=lambda e: (e.set_option("assoc_batch", 0), box.setdefault("e", e)))
    assert box["e"].batch_counters()["batch_columns"] == 0
    assert summary["clusters"] >= 3


@pytest.mark.parametrize("name", ["s64_static", "s64_full_2200", "s64_dropouts", "s128_full_1700", "s32_small_sensor", "j_s64_jitter_wide",
                                  "s64_robot_tf_tilted", "s64_fog_and_ego"])
def test_batch_parallel_association_takes_ordinary_streams(name, oracle_lib):
    """k_assocb (groups of 64 columns at once: pointer jumping over the parent chains, one alive word per tree, a lane-parallel cluster
    timeline) associates every column of streams without exceptions — also in mirror mode, where it re-takes the visit counts of columns
    whose scan looked past the first unpublished column — with the oracle's events, roots, ids and per-tree values."""
    stream, cfg, tf = cases.build_case(name)
    box = {}
    summary = util.run_and_compare(stream, cfg, chunks=[stream.sensor.num_columns, 97, 1, 200], robot_tf=tf,
                                   engine_setup=lambda e: box.setdefault("e", e))
    bc = box["e"].batch_counters()
    assert bc["batch_bails"] == 0 and summary["engine_state"]["error_b"] == 0, bc
    assert bc["batch_columns"] >= summary["published_columns"]


@pytest.mark.parametrize("name", cases.EXCEPTION_CASES)
@pytest.mark.parametrize("batch", [1, 0])
def test_streams_made_to_provoke_refused_attaches(name, batch, oracle_lib):
    """Gaps in slanted / near surfaces, jittered firing azimuths, and a hand-made case in which the reference refuses attaches because the
    candidate's tree was finished a column earlier (cc.cpp:658; possible for steep lasers, where the 3-D angle between two returns is smaller
    than their azimuth difference): equal to the oracle with the batch-parallel kernel in front and without it."""
    stream, cfg, tf = cases.build_case(name)
    box = {}
    for chunks in ([stream.sensor.num_columns], [61, 97, 1, 200]):
        summary = util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=tf,
                                       engine_setup=lambda e: (e.set_option("assoc_batch", batch), box.__setitem__("e", e)))
        if name == "x_s64_refused_attach":
            assert summary["engine_state"]["error_b"] > 0, "the exact serial replay of the refused attaches was expected"
            if batch:
                why = box["e"].batch_counters()["bail_reasons"]
                # a tree met after its cluster finished (round 4's kernel classified two of these stops as reason 6, a candidate behind the first
                # unpublished column; the pipelined kernel of round 5 stops at the same groups and names the late point — tools/find_reach_bail.py looks for
                # inputs that still give reason 6)
                assert why[4] + why[5] > 0, why


@pytest.mark.parametrize("name,reason", [("c_s64_sparse_clutter", 0), ("c_s64_near_clutter", 3), ("c_s64_mixed_clutter", 0), ("c_s128_sparse_clutter", 0)])
def test_vegetation_stays_exact_on_and_off_the_fast_path(name, reason, oracle_lib):
    """Natural data at the edge of what the batch-parallel association takes (cc.cpp:654-659, 688-690, 913-924 are what the serial kernels replay).
    Sparse leaves: dozens of trees unfinished side by side and dozens of new ones per group of columns — since a group takes at most half of the
    free tree lanes (cc_assocb.h: group_header) the fast path keeps all of it (no stop; round 5's first kernel stopped once per rotation with
    reason 1). A shell of near returns all around grows a tree that may span a rotation (reason 3: those groups go to the serial kernel, the fast
    path resumes behind them). With fewer tree lanes (lds_tree_limit 40) the sparse scenes do run out of lanes (reason 1) and the hand-over to
    the serial kernel is exercised on the same data. Equal to the oracle every time."""
    stream, cfg, tf = cases.build_case(name)
    box = {}
    for chunks in ([stream.sensor.num_columns], [97, 1, 200]):
        summary = util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=tf, engine_setup=lambda e: box.__setitem__("e", e))
        bc = box["e"].batch_counters()
        if reason:
            assert bc["batch_bails"] > 0 and bc["bail_reasons"][reason] > 0, bc
        else:
            assert bc["batch_bails"] == 0 and bc["batch_columns"] >= summary["published_columns"], bc
        assert bc["batch_columns"] > 0 and summary["clusters"] > 50
        if not reason:
            summary = util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=tf,
                                           engine_setup=lambda e: (e.set_option("lds_tree_limit", 40), box.__setitem__("e", e)))
            bc = box["e"].batch_counters()
            assert bc["bail_reasons"][1] > 0 and bc["batch_columns"] > 0, bc


def test_small_calls_ask_the_host_for_the_serial_kernel(oracle_lib):
    """Calls of a few firings are ONE launch (k_small_all); when the batch-parallel association in it has to stop (here: the refused attaches of
    x_s64_refused_attach, cc.cpp:654-659) the kernel says so through pinned memory and the host launches the serial fall-back kernel behind it.
    One- to three-firing calls over the whole stream: equal to the oracle, the request was made, and with small_all = 0 (three launches per call)
    the same results without it."""
    stream, cfg, tf = cases.build_case("x_s64_refused_attach")
    box = {}
    for small_all in (1, 0):
        summary = util.run_and_compare(stream, cfg, chunks=[1, 3, 2, 17, 40], robot_tf=tf,
                                       engine_setup=lambda e: (e.set_option("small_all", small_all), box.__setitem__("e", e)))
        why = box["e"].batch_counters()["bail_reasons"]
        assert summary["engine_state"]["error_b"] > 0, "the exact serial replay of the refused attaches was expected"
        assert (why[7] > 0) == bool(small_all), why


@pytest.mark.parametrize("first_call", [132, 236, 496])
def test_attach_to_a_tree_finished_in_an_earlier_launch(first_call, oracle_lib):
    """cc.cpp:654-659 across launches: in x_s64_refused_attach the second post of a pair joins (by 3-D distance) the first post's tree one column
    after that tree's cluster finished — while the unbroken wall keeps the first unpublished column far behind, so the finished tree's cells are
    still inside the scan's reach. A call boundary right behind the second post's firing puts the finish into one launch of the batch-parallel
    kernel and the refused attach into the first group of the next: the slot ring starts out with the tree marked dead, the point's parent chain
    ends there, and the kernel must stop for reason 4 (AB_BAIL_DEAD, cc_assocb.h) and leave the group to the exact serial kernel
    (tools/find_dead_bail.py found the call sizes). Everything equal to the oracle."""
    stream, cfg, tf = cases.build_case("x_s64_refused_attach")
    box = {}
    summary = util.run_and_compare(stream, cfg, chunks=[first_call, 700], robot_tf=tf, engine_setup=lambda e: box.__setitem__("e", e))
    why = box["e"].batch_counters()["bail_reasons"]
    assert why[4] > 0, why
    assert summary["engine_state"]["error_b"] > 0  # (columns replayed by the serial kernel)


@pytest.mark.parametrize("name,reason", [("s64_forced_finish_ring", 3), ("s64_no_early_stop", 2), ("s64_min_steps_3", 2)])
@pytest.mark.parametrize("rounds", [0, 1, 2, 4])
def test_batch_parallel_association_hands_exceptions_to_the_serial_kernel(name, reason, rounds, oracle_lib):
    """Groups that could differ from the sequential semantics (3: a tree / cluster that may reach the one-rotation limits cc.cpp:657, 913-924;
    2: more link candidates than k_scan records) are left to k_assoc3: with assoc_rounds > 1 only that group, then the batch-parallel kernel
    continues."""
    stream, cfg, tf = cases.build_case(name)
    box = {}
    summary = util.run_and_compare(stream, cfg, chunks=[stream.sensor.num_columns, 97], robot_tf=tf,
                                   engine_setup=lambda e: (e.set_option("assoc_rounds", rounds), box.setdefault("e", e)))
    bc = box["e"].batch_counters()
    assert bc["batch_bails"] > 0 and bc["bail_reasons"][reason] > 0, bc
    assert summary["engine_state"]["error_b"] > 0


@pytest.mark.parametrize("name,option,value", [
    ("s64_translate", "scan_packed", 1), ("s64_dropouts", "scan_packed", 1), ("j_s64_jitter_wide", "scan_packed", 1),
    ("s128_offsets", "scan_packed", 0), ("s128_full_1700", "scan_packed", 0), ("s96_offsets", "scan_packed", 0), ("s32_small_sensor", "scan_packed", 1),
    ("s128_offsets", "parallel_insert", 2), ("s128_full_1700", "parallel_insert", 0), ("j_s128_offsets_jitter", "parallel_insert", 2),
    ("j_s64_jitter", "parallel_insert", 0), ("s64_full_2200", "parallel_insert", 2),
])
def test_kernel_variants_give_the_same_result(name, option, value, oracle_lib):
    """The window scan has two kernels (rows as lanes in lock step — since round 4 also with two rows per lane — / active points packed into the
    lanes) and the insertion three
    (block-parallel single-column, block-parallel multi-column with the collision rule checked, serial): every selection must reproduce the
    oracle. The defaults are covered by test_engine_matches_oracle; here the other choice of each option."""
    stream, cfg, tf = cases.build_case(name)
    util.run_and_compare(stream, cfg, chunks=[stream.sensor.num_columns, 211], robot_tf=tf, engine_setup=lambda e: e.set_option(option, value))


def test_multi_column_insertion_takes_the_steady_part(oracle_lib):
    """VLS-128-shaped stream (every firing spans ~60 columns): after the ring has started, k_insert_multi takes whole calls (debug
    counter 6 = firings taken by the block-parallel kernels)."""
    import ctypes as C
    from continuous_clustering_amd import Engine, load_library
    stream, cfg, tf = cases.build_case("s128_full_1700")
    e = Engine(cfg, 128, 1)
    n = 0
    for b in range(2):
        sl = slice(b * 1700, (b + 1) * 1700)
        assert e.add_firings(stream.xyz[sl], stream.intensity[sl], stream.poses[sl]) == 0
    L = load_library()
    L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(16, dtype=np.uint64)
    L.cc_engine_debug_counters(e.h, 0, out.ctypes.data)
    assert out[6] >= 1700, out[6]  # the whole second call (the first one starts the ring in the serial kernel)


@pytest.mark.parametrize("rows", [8, 16, 24, 40, 48, 50, 80, 96, 100, 112])
def test_row_counts_between_the_usual_ones(rows, oracle_lib):
    """Sensors with row counts other than 32 / 64 / 128: k_seg_scan loads 16 rows x 64 columns at a time when the row count is a multiple of 16 and
    falls back to every lane reading its own rows otherwise (8 at a time, or one by one when the count is not a multiple of 8); above 64 rows every
    kernel works on two rows per lane with the upper lanes partly idle. Events and published columns equal the oracle's, chunked calls."""
    sen = synth.SensorModel(num_rows=rows, num_columns=360, incl_top_deg=6.0, incl_bottom_deg=-26.0)
    stream = synth.make_stream(360 * 2 + 90, seed=3000 + rows, sensor=sen, motion=synth.Motion.translate(8.0))
    cfg = cases._kitti(360)
    util.run_and_compare(stream, cfg, chunks=[360, 97, 23], robot_tf=None)
