"""Shared helpers of the parity tests: run a stream through the CPU oracle and through the HIP engine and
compare everything the reference would have reported (events + published columns)."""
from __future__ import annotations

import numpy as np

from continuous_clustering_amd import capi

FLOAT_FIELDS = ("x", "y", "z", "distance", "inclination_angle", "continuous_azimuth_angle", "finished_at_continuous_azimuth_angle")
EXACT_FIELDS = ("ground_point_label", "debug_ground_point_label", "is_ignored", "global_column_index", "source_firing",
                "tree_root_global_column", "tree_root_row", "tree_num_points", "cluster_width", "number_of_child_points",
                "number_of_visited_neighbors", "belongs_to_finished_cluster", "tree_parent_global_column", "tree_parent_row")
STATE_FIELDS = ("reset_required", "ring_buffer_start_global_column_index", "ring_buffer_end_global_column_index",
                "first_unfinished_global_column_index", "first_unpublished_global_column_index", "cluster_counter",
                "firings_consumed", "cells_published", "clusters_finished", "n_unfinished_trees")


def run_oracle(stream, cfg, robot_tf=None):
    from oracle.pyoracle import Oracle, IDENTITY_TF
    o = Oracle(cfg, stream.sensor.num_rows, IDENTITY_TF if robot_tf is None else robot_tf)
    rc = o.add_firings(stream.xyz, stream.intensity, stream.poses)
    return o, rc


def canonical_ids(ids: np.ndarray) -> np.ndarray:
    """Relabel clusters by order of first appearance in column-major order (0 stays 0)."""
    flat = ids.reshape(-1)
    _, first = np.unique(flat, return_index=True)
    order = flat[np.sort(first)]
    order = order[order != 0]
    lut = {int(v): i + 1 for i, v in enumerate(order)}
    lut[0] = 0
    return np.vectorize(lut.get, otypes=[np.uint64])(flat).reshape(ids.shape)


def assert_float_equal(name, a, b):
    an, bn = np.isnan(a), np.isnan(b)
    assert np.array_equal(an, bn), f"{name}: NaN pattern differs at {np.argwhere(an != bn)[:5]}"
    itype = np.uint32 if a.dtype == np.float32 else np.uint64
    av, bv = a[~an].view(itype), b[~bn].view(itype)
    bad = np.nonzero(av != bv)[0]
    assert bad.size == 0, f"{name}: {bad.size} values differ bitwise, first {a[~an][bad[:3]]} vs {b[~bn][bad[:3]]}"


# produced only with the engine option "mirror_fields" (on while events are recorded, off in the pipelined throughput mode)
MIRROR_ONLY_FIELDS = ("number_of_visited_neighbors", "finished_at_continuous_azimuth_angle", "tree_num_points", "cluster_width")


def compare_columns(ao: dict, ae: dict, c0: int, check_raw_ids=True, mirror=True):
    for f in EXACT_FIELDS:
        if not mirror and f in MIRROR_ONLY_FIELDS:
            continue
        bad = np.argwhere(ao[f] != ae[f])
        assert bad.size == 0, f"{f}: {len(bad)} cells differ, first (col {c0 + bad[0][0]}, row {bad[0][1]}): " \
                              f"oracle {ao[f][tuple(bad[0])]} engine {ae[f][tuple(bad[0])]}"
    for f in FLOAT_FIELDS:
        if not mirror and f in MIRROR_ONLY_FIELDS:
            continue
        assert_float_equal(f, ao[f], ae[f])
    # the bar of BASELINE.json: identical canonical partition; stronger: the raw reference numbering
    assert np.array_equal(canonical_ids(ao["id"]), canonical_ids(ae["id"])), "canonical cluster labels differ"
    if check_raw_ids:
        bad = np.argwhere(ao["id"] != ae["id"])
        assert bad.size == 0, f"raw cluster ids differ at (col {c0 + bad[0][0]}, row {bad[0][1]})"


def run_and_compare(stream, cfg, chunks=None, robot_tf=None, expect_rc=0, check_raw_ids=True, engine_setup=None):
    """Feed `stream` to the oracle (all at once) and to a 1-stream engine (in `chunks` firings per call); after every
    engine call compare the events it produced and the columns it published with the oracle's record."""
    from continuous_clustering_amd import Engine, IDENTITY_TF
    oracle, orc = run_oracle(stream, cfg, robot_tf)
    assert orc == expect_rc, f"oracle rc {orc} ({oracle.last_error()}), expected {expect_rc}"
    eo = oracle.drain_events()
    engine = Engine(cfg, stream.sensor.num_rows, 1, 0, IDENTITY_TF if robot_tf is None else robot_tf)
    if engine_setup is not None:
        engine_setup(engine)
    n = stream.n_firings
    chunks = chunks or [n]
    f = i = 0
    ev_pos = 0
    rc = 0
    n_cols = 0
    while f < n and rc == 0:
        m = min(chunks[i % len(chunks)], n - f)
        rc = engine.add_firings(stream.xyz[f:f + m], stream.intensity[f:f + m], stream.poses[f:f + m])
        f += m
        i += 1
        if rc != 0:
            break
        ee = engine.drain_events()
        ref = eo[ev_pos:ev_pos + len(ee)]
        assert len(ref) == len(ee), f"engine produced more events ({ev_pos + len(ee)}) than the oracle ({len(eo)})"
        for fld in ("type", "a", "b", "c", "d", "column"):
            bad = np.nonzero(ref[fld] != ee[fld])[0]
            assert bad.size == 0, f"event {ev_pos + bad[0]} field {fld}: oracle {ref[bad[0]]} engine {ee[bad[0]]}"
        ev_pos += len(ee)
        pub = ee[ee["type"] == capi.EV_PUBLISH_COLUMNS]
        pub = pub[pub["b"] >= pub["a"]]
        if len(pub):
            lo, hi = int(pub["a"].min()), int(pub["b"].max())
            step = 2048
            for c0 in range(lo, hi + 1, step):
                c1 = min(hi, c0 + step - 1)
                compare_columns(oracle.read_published(c0, c1), engine.read_columns(c0, c1), c0, check_raw_ids)
                n_cols += c1 - c0 + 1
    assert rc == expect_rc, f"engine rc {rc} ({engine.last_error()}), expected {expect_rc}"
    if rc == 0:
        assert ev_pos == len(eo), f"engine produced {ev_pos} events, oracle {len(eo)}"
        so, se = oracle.state(), engine.state()
        for k in STATE_FIELDS:
            assert so[k] == se[k], f"state.{k}: oracle {so[k]} engine {se[k]}"
    return {"events": int(ev_pos), "clusters": int((eo["type"] == capi.EV_CLUSTER).sum()), "published_columns": n_cols,
            "oracle_state": oracle.state(), "engine_state": engine.state() if rc == 0 else None}
