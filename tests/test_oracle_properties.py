"""CPU: size-independent properties of the oracle that do not depend on its own association code:
 * the published partition equals the connected components of the static neighbour graph (SURVEY.md 8a "derived fact"),
   recomputed here by an independent numpy/Python re-simulation of the window scan + a plain union-find;
 * clusters of <= 5 points keep id 0, ids are dense and increasing;
 * ground / ignore masks obey their definitions;
 * feeding the same stream in different call chunkings changes nothing."""
import math

import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import capi


def static_partition(cols: dict, cfg, num_rows: int, num_columns: int):
    """Union every (point, candidate) pair the ordered window scan of cc.cpp:698-771 examines and accepts, assuming
    every first match roots the point (no refusals). Operates on published column data only."""
    x, y, z = cols["x"], cols["y"], cols["z"]
    incl, ign, dist = cols["inclination_angle"], cols["is_ignored"], cols["distance"]
    ncols = x.shape[0]
    width = np.float32(np.float32(2 * math.pi) / np.float32(num_columns))
    maxd = np.float32(cfg.max_distance)
    maxd2 = np.float32(maxd * maxd)
    parent = np.arange(ncols * num_rows)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    def union(a, b):
        a, b = find(a), find(b)
        if a != b:
            parent[max(a, b)] = min(a, b)

    for c in range(ncols):
        for r in range(num_rows):
            if ign[c, r]:
                continue
            mad = np.float32(math.asin(np.float32(maxd / dist[c, r])))
            needed = min(int(math.ceil(np.float32(mad / width))), cfg.max_steps_in_row)
            rooted = False
            oc = c
            for sb in range(needed + 1):
                for d in (-1, 1):
                    if d == 1 and sb == 0:
                        continue
                    sv = 1 if (d == 1 or sb == 0) else 0
                    orow = r + d if (d == 1 or sb == 0) else r
                    while 0 <= orow < num_rows and sv <= cfg.max_steps_in_column:
                        if abs(np.float32(incl[oc, orow] - incl[c, r])) > mad:
                            break
                        if not ign[oc, orow]:
                            dx = np.float32(x[c, r] - x[oc, orow])
                            dy = np.float32(y[c, r] - y[oc, orow])
                            dz = np.float32(z[c, r] - z[oc, orow])
                            if np.float32(np.float32(dx * dx + dy * dy) + dz * dz) < maxd2:
                                union(c * num_rows + r, oc * num_rows + orow)
                                rooted = True
                        if rooted and cfg.stop_after_association_enabled and sv >= cfg.stop_after_association_min_steps:
                            break
                        orow += d
                        sv += 1
                if rooted and cfg.stop_after_association_enabled and sb >= cfg.stop_after_association_min_steps:
                    break
                if oc == 0:
                    break
                oc -= 1
    return np.array([find(i) for i in range(ncols * num_rows)]).reshape(ncols, num_rows)


@pytest.mark.parametrize("name", ["g_s64_translate", "g_s128_offsets"])
def test_partition_equals_static_components(name, oracle_lib):
    stream, cfg, tf = cases.build_case(name)
    o, rc = util.run_oracle(stream, cfg, tf)
    assert rc == 0
    frm, to = o.published_range()
    cols = o.read_published(frm, to)
    R = stream.sensor.num_rows
    comp = static_partition(cols, cfg, R, cfg.num_columns)
    ids = cols["id"]
    obstacle = cols["is_ignored"] == 0
    # ignore the stream tail: clusters still growing when the stream ended are not published completely
    tail = ids.shape[0] - 120
    sizes = np.bincount(comp[obstacle].ravel(), minlength=comp.size)
    checked = 0
    for c in np.unique(comp[:tail][obstacle[:tail]]):
        members = (comp == c) & obstacle
        if members[tail:].any():
            continue
        lab = np.unique(ids[members])
        assert lab.size == 1, f"component {c} carries cluster ids {lab}"
        if sizes[c] > 5:
            assert lab[0] != 0
            # and no other component shares this id
            assert ((ids == lab[0]) & ~members).sum() == 0
        else:
            assert lab[0] == 0  # cc.cpp:936
        checked += 1
    assert checked > 20


def test_masks_obey_definitions(oracle_lib):
    stream, cfg, tf = cases.build_case("s64_fog_and_ego")
    o, rc = util.run_oracle(stream, cfg, tf)
    frm, to = o.published_range()
    a = o.read_published(frm, to)
    nan = np.isnan(a["distance"])
    assert (a["is_ignored"][nan] == 1).all() and (a["ground_point_label"][nan] == capi.GP_UNKNOWN).all()
    not_obstacle = a["ground_point_label"] != capi.GP_OBSTACLE
    assert (a["is_ignored"][not_obstacle] == 1).all()
    assert (a["id"][a["is_ignored"] == 1] == 0).all()
    assert set(np.unique(a["ground_point_label"])) <= {capi.GP_UNKNOWN, capi.GP_GROUND, capi.GP_OBSTACLE, capi.GP_EGO_VEHICLE, capi.GP_FOG}
    assert (a["ground_point_label"] == capi.GP_FOG).any() and (a["ground_point_label"] == capi.GP_EGO_VEHICLE).any()
    # every cell of a published column carries that column's index (cc.cpp:348)
    assert (a["global_column_index"] == np.arange(frm, to + 1)[:, None]).all()
    ids = np.unique(a["id"])
    assert ids[0] == 0 and (np.diff(ids) >= 1).all()


def test_events_are_ordered_like_the_reference_callbacks(oracle_lib):
    stream, cfg, tf = cases.build_case("g_s64_translate")
    o, rc = util.run_oracle(stream, cfg, tf)
    ev = o.drain_events()
    # per column: ground callback first, then clusters, then exactly one publish callback (every_nth_column = 1)
    cols = ev["column"]
    assert (np.diff(cols) >= 0).all()
    for c in np.unique(cols)[:200]:
        t = ev["type"][cols == c]
        assert t[0] == capi.EV_GROUND_COLUMN and t[-1] == capi.EV_PUBLISH_COLUMNS
        assert (t[1:-1] == capi.EV_CLUSTER).all()
    pub = ev[ev["type"] == capi.EV_PUBLISH_COLUMNS]
    assert (pub["a"][1:] == pub["b"][:-1] + 1).all()  # published ranges tile the column axis without gaps


def test_forced_finish_path_is_exercised(oracle_lib):
    stream, cfg, tf = cases.build_case("s64_forced_finish_ring")
    o, rc = util.run_oracle(stream, cfg, tf)
    assert rc == 0 and o.state()["error_a"] >= 2  # "Found a cluster exceeding one rotation" fired (cc.cpp:913-919)


def test_error_paths(oracle_lib):
    from oracle.pyoracle import Oracle
    stream, cfg, tf = cases.build_case("g_s64_translate")
    o = Oracle(cfg, 64, robot_from_sensor=None)
    assert o.add_firings(stream.xyz[:50], stream.intensity[:50], stream.poses[:50]) == capi.CC_ERR_NO_ROBOT_TRANSFORM
    assert "Transform robot frame" in o.last_error()
    # first firing straddling the negative x axis -> reset_required (cc.cpp:252-261)
    bad = stream.xyz[:1].copy()
    bad[0, :32, 1] = np.abs(bad[0, :32, 1]) + 0.1
    bad[0, 32:, 1] = -np.abs(bad[0, 32:, 1]) - 0.1
    bad[0, :, 0] = -np.abs(bad[0, :, 0]) - 1.0
    o2 = Oracle(cfg, 64)
    assert o2.add_firings(bad, stream.intensity[:1], stream.poses[:1]) == 0
    assert o2.state()["reset_required"] == 1


def test_stage_pipeline_of_the_cpu_baseline_gives_the_single_threaded_results(oracle_lib):
    """BASELINE.md mode B (bench.py: cpu_baseline.mode_b) runs the oracle's stages on three threads connected by bounded queues. It is a timing
    harness — but it must do the same work: events and published columns equal to the depth-first single-threaded run."""
    import numpy as np
    from continuous_clustering_amd import capi, synth
    from oracle.pyoracle import Oracle
    cfg = capi.Config.kitti()
    cfg.num_columns = 360
    sensor = synth.SensorModel(num_rows=64, num_columns=360)
    st = synth.make_stream(360 * 4 + 50, seed=9, sensor=sensor, motion=synth.Motion.translate())
    a, b = Oracle(cfg, 64), Oracle(cfg, 64)
    assert a.add_firings(st.xyz, st.intensity, st.poses) == 0
    assert b.add_firings_pipeline(st.xyz, st.intensity, st.poses) == 0, b.last_error()
    ea, eb = a.drain_events(), b.drain_events()
    # the ground-view events of the pipeline come from another thread than the cluster-view ones: compare per type, in order
    for t in (1, 2, 3):
        assert np.array_equal(ea[ea["type"] == t], eb[eb["type"] == t]), t
    assert a.state() == b.state()
    fa, ta = a.published_range()
    pa, pb = a.read_published(fa, ta), b.read_published(fa, ta)
    for k in pa:
        x, y = pa[k], pb[k]
        if x.dtype.kind == "f":
            assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(x[~np.isnan(x)], y[~np.isnan(y)]), k
        else:
            assert np.array_equal(x, y), k
    c = Oracle(cfg, 64, record=False)
    assert c.time_firings_pipeline(st.xyz, st.intensity, st.poses) > 0
    assert c.state()["cells_published"] == a.state()["cells_published"]
