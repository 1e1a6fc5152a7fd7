"""GPU: the resident single-stream kernel (engine option "resident", cc_k_publish.h: k_resident) — the reference's synchronous calling pattern
(is_single_threaded: addFiring returns when the firing's columns are through, thread_pool.hpp:58-64, cc.cpp:88-93) without a kernel dispatch
per call. Every chunking of the parity cases through the resident path must equal the oracle event for event and column for column (the
columns are read beside the idling kernel, on another stream); the kernel must start and stop cleanly around reset / set_config /
set_option / larger calls / destroy, leave by itself when no call comes (watchdog), and hand calls that need the serial fall-backs to the host."""
import time

import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import Engine, capi, synth

pytestmark = pytest.mark.gpu


def _resident(holder):
    def setup(e):
        e.set_option("resident", 1)
        holder.append(e)
    return setup


@pytest.mark.parametrize("name,chunks", [("s64_translate", [1]), ("s64_turn", [1, 2, 3, 5, 8]), ("s64_fog_and_ego", [7, 63, 1, 1, 1]),
                                         ("s64_forced_finish_ring", [1]), ("j_s64_jitter", [1, 1, 4]), ("s64_robot_tf_tilted", [2]),
                                         ("c_s64_mixed_clutter", [1]), ("x_s64_refused_attach", [1, 3]), ("s64_counterclockwise", [5, 1])])
def test_every_chunking_through_the_resident_kernel_equals_the_oracle(name, chunks, oracle_lib):
    stream, cfg, tf = cases.build_case(name)
    h = []
    s = util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=tf, engine_setup=_resident(h))
    c = h[0].resident_counters()
    assert s["events"] > 0 and c["launches"] >= 1
    n_calls = -(-stream.n_firings * len(chunks) // sum(chunks))  # about: calls made
    h[0].set_option("resident", 0)  # stops the kernel: its call count is final now
    c = h[0].resident_counters()
    assert not c["running"] and c["calls"] >= 0.9 * n_calls - 2, (c, n_calls)
    # few launches: the kernel stays (it leaves for calls that need the serial fall-backs, which these cases provoke now and then)
    assert c["launches"] <= max(3, n_calls // 4), (c, n_calls)
    h[0].close()


def test_watchdog_and_restart(oracle_lib):
    cfg = capi.Config.kitti()
    cfg.num_columns = 720
    sensor = synth.SensorModel(num_rows=64, num_columns=720)
    stream = synth.make_stream(720 + 200, seed=77, sensor=sensor, motion=synth.Motion.translate())
    o, rc = util.run_oracle(stream, cfg)
    assert rc == 0
    e = Engine(cfg, 64, 1)
    e.set_option("resident", 1)
    e.set_option("resident_idle_ms", 5)
    f = 0
    for k in range(920):
        assert e.add_firings(stream.xyz[f:f + 1], stream.intensity[f:f + 1], stream.poses[f:f + 1]) == 0, e.last_error()
        f += 1
        if k in (100, 500):
            assert e.resident_counters()["running"]
            time.sleep(0.05)  # ten watchdog periods: the kernel has left
            assert not e.resident_counters()["running"]
    c = e.resident_counters()
    assert c["launches"] >= 3, c
    so, se = o.state(), e.state()
    for k in util.STATE_FIELDS:
        assert so[k] == se[k], (k, so[k], se[k])
    hi = se["first_unpublished_global_column_index"] - 1
    lo = max(0, se["ring_buffer_start_global_column_index"])
    util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi), lo)
    e.close()


def test_clean_stop_around_everything_else(oracle_lib):
    """reset, set_config, set_robot_from_sensor, a call too large for the kernel, device-side calls, destroy with the kernel running."""
    import torch
    cfg = capi.Config.kitti()
    cfg.num_columns = 720
    sensor = synth.SensorModel(num_rows=64, num_columns=720)
    a = synth.make_stream(720 * 2, seed=78, sensor=sensor, motion=synth.Motion.translate())
    e = Engine(cfg, 64, 1)
    e.set_option("resident", 1)
    for f in range(300):
        assert e.add_firings(a.xyz[f:f + 1], a.intensity[f:f + 1], a.poses[f:f + 1]) == 0
    assert e.resident_counters()["running"]
    e.reset()
    assert not e.resident_counters()["running"]
    from continuous_clustering_amd import IDENTITY_TF
    e.set_robot_from_sensor(IDENTITY_TF)
    o, rc = util.run_oracle(a, cfg)
    f = 0
    i = 0
    sizes = [1, 1, 1, 200, 1, 1, 64, 1, 2, 1]  # 200 and 64 do not fit the resident kernel: it stops, the general path runs, it starts again
    while f < a.n_firings:
        m = min(sizes[i % len(sizes)], a.n_firings - f)
        assert e.add_firings(a.xyz[f:f + m], a.intensity[f:f + m], a.poses[f:f + m]) == 0, e.last_error()
        f += m
        i += 1
        if i == 20:
            e.set_config(cfg)  # same configuration: stops the kernel, no reset required
    so, se = o.state(), e.state()
    for k in util.STATE_FIELDS:
        assert so[k] == se[k], (k, so[k], se[k])
    hi = se["first_unpublished_global_column_index"] - 1
    lo = max(0, se["ring_buffer_start_global_column_index"])
    util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi), lo)
    assert e.resident_counters()["launches"] >= 3
    assert e.add_firings(a.xyz[:1], a.intensity[:1], a.poses[:1]) in (0, capi.CC_OK)  # (whatever it does to the stream: the kernel runs again)
    e.close()  # destroy with the kernel on the stream: must return


def test_latency_of_a_one_firing_call_is_reported(oracle_lib):
    """Not a threshold test (boxes differ, and the two paths are within 2 - 3 % of each other): prints p50 / p99 of one-firing calls with and
    without the resident kernel; fails only if the resident path is grossly slower than the launch-per-call path."""
    cfg = capi.Config.kitti()
    sensor = synth.SensorModel.s64()
    st = synth.make_stream(2200 * 2, seed=79, sensor=sensor, motion=synth.Motion.translate())
    res = {}
    for mode in (0, 1):
        e = Engine(cfg, 64, 1)
        e.set_option("resident", mode)
        lat = []
        for f in range(st.n_firings):
            t0 = time.perf_counter()
            assert e.add_firings(st.xyz[f:f + 1], st.intensity[f:f + 1], st.poses[f:f + 1]) == 0
            lat.append(time.perf_counter() - t0)
        e.close()
        v = np.array(lat[400:]) * 1e6
        res[mode] = (float(np.percentile(v, 50)), float(np.percentile(v, 99)))
    print("one-firing call, engine only, us p50 / p99: launch per call", res[0], "resident", res[1])
    assert res[1][0] <= res[0][0] * 1.3, res


def test_mirrored_views_equal_the_view_kernel(oracle_lib):
    """A small call mirrors the host views of the columns its events name together with its results (HostMirror::view); cc_engine_read_columns then
    serves them without a kernel. Every field must equal what k_view produces for the same column at the same moment: two engines, one with the
    option off, fed the same firings one call at a time (calls of 1, 2, 3 and 7 firings), every column an event names read from both."""
    stream, cfg, tf = cases.build_case("s64_turn")
    from continuous_clustering_amd import IDENTITY_TF
    a = Engine(cfg, 64, 1, 0, IDENTITY_TF if tf is None else tf)
    b = Engine(cfg, 64, 1, 0, IDENTITY_TF if tf is None else tf)
    b.set_option("mirror_views", 0)
    fields = [k for k in capi.COLUMN_FIELDS if k != "number_of_child_points"]  # (the child counts need the view kernel's extra pass)
    f, i, sizes, checked = 0, 0, [1, 1, 2, 1, 3, 1, 7], 0
    while f < stream.n_firings:
        m = min(sizes[i % len(sizes)], stream.n_firings - f)
        for e in (a, b):
            assert e.add_firings(stream.xyz[f:f + m], stream.intensity[f:f + m], stream.poses[f:f + m]) == 0, e.last_error()
        f += m
        i += 1
        ea, eb = a.drain_events(), b.drain_events()
        assert np.array_equal(ea, eb)
        for ev in ea:
            if ev["type"] == capi.EV_CLUSTER or ev["b"] < ev["a"]:
                continue
            lo, hi = int(ev["a"]), int(ev["b"])
            for c0 in range(lo, hi + 1, 8):
                c1 = min(hi, c0 + 7)
                ca, cb = a.read_columns(c0, c1, fields=fields), b.read_columns(c0, c1, fields=fields)
                for k in ca:
                    x, y = ca[k], cb[k]
                    if x.dtype.kind == "f":
                        assert np.array_equal(x.view(np.uint32 if x.dtype == np.float32 else np.uint64), y.view(np.uint32 if y.dtype == np.float32 else np.uint64)), (k, c0)
                    else:
                        assert np.array_equal(x, y), (k, c0, c1)
                checked += c1 - c0 + 1
    assert checked > 2 * stream.n_firings // 2
    va, vb = a.view_counters(), b.view_counters()
    assert vb["mirror"] == 0 and va["mirror"] > 0.5 * (va["mirror"] + va["kernel"]), (va, vb)  # (calls of 7 firings name more than 8 columns)
    a.close()
    b.close()


def test_read_column_ranges_equals_single_reads(oracle_lib):
    """cc_engine_read_column_ranges (one launch — or the mirrored views — for up to 8 ranges) returns, range after range, what cc_engine_read_columns
    returns for each range: after a large call (view kernel) and after a one-firing call (mirror), with and without the child counts."""
    stream, cfg, tf = cases.build_case("s64_translate")
    from continuous_clustering_amd import IDENTITY_TF
    e = Engine(cfg, 64, 1, 0, IDENTITY_TF if tf is None else tf)
    n = stream.n_firings
    assert e.add_firings(stream.xyz[:n - 40], stream.intensity[:n - 40], stream.poses[:n - 40]) == 0
    st = e.state()
    hi = st["first_unfinished_global_column_index"] - 1
    lo = max(st["ring_buffer_start_global_column_index"], 0)
    ranges = [(lo + 3, lo + 3), (lo + 10, lo + 17), (hi - 5, hi), (lo + 40, lo + 41)]

    def check(fields):
        got = e.read_column_ranges(ranges, fields=fields)
        off = 0
        for a, b in ranges:
            one = e.read_columns(a, b, fields=fields)
            for k in one:
                x, y = one[k], got[k][off:off + b - a + 1]
                assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (k, a, b)
            off += b - a + 1
    check(None)
    check([k for k in capi.COLUMN_FIELDS if k != "number_of_child_points"])
    for f in range(n - 40, n):
        assert e.add_firings(stream.xyz[f:f + 1], stream.intensity[f:f + 1], stream.poses[f:f + 1]) == 0
        ev = e.drain_events()
        ranges = [(int(x["a"]), int(x["b"])) for x in ev if x["type"] != capi.EV_CLUSTER and x["b"] >= x["a"]][:8]
        if ranges:
            check([k for k in capi.COLUMN_FIELDS if k != "number_of_child_points"])
    assert e.view_counters()["mirror"] > 0
    e.close()
