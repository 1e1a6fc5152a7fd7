"""GPU: cluster hand-off — member points of finished clusters gathered on the device (cc_engine_gather_cluster_points, the point
collection of cc.cpp:985-1033) against the published cluster ids of the same engine and the oracle's events."""
import numpy as np
import pytest

import cases
from continuous_clustering_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,chunk", [("s64_translate", 240), ("s64_ring_with_objects", 97), ("s128_offsets", 300)])
def test_gathered_points_are_the_cluster(name, chunk, oracle_lib):
    from continuous_clustering_amd import Engine, IDENTITY_TF
    from oracle.pyoracle import Oracle
    stream, cfg, tf = cases.build_case(name)
    R = stream.sensor.num_rows
    e = Engine(cfg, R, 1, 0, IDENTITY_TF if tf is None else tf)
    o = Oracle(cfg, R, IDENTITY_TF if tf is None else tf)
    assert o.add_firings(stream.xyz, stream.intensity, stream.poses) == 0
    oracle_clusters = {int(ev["c"]): int(ev["d"]) for ev in o.drain_events() if ev["type"] == capi.EV_CLUSTER}
    gathered = {}
    n = stream.n_firings
    for f in range(0, n, chunk):
        m = min(chunk, n - f)
        assert e.add_firings(stream.xyz[f:f + m], stream.intensity[f:f + m], stream.poses[f:f + m]) == 0
        ev = e.drain_events()
        cl = ev[ev["type"] == capi.EV_CLUSTER]
        if len(cl) == 0:
            continue
        offsets, gcol, row = e.gather_cluster_points(cl)
        for i, c in enumerate(cl):
            g, r = gcol[offsets[i]:offsets[i + 1]], row[offsets[i]:offsets[i + 1]]
            assert len(g) == c["d"] == oracle_clusters[int(c["c"])]
            assert (g >= c["a"]).all() and (g <= c["b"]).all() and (r >= 0).all() and (r < R).all()
            key = g * R + r
            assert (np.diff(key) > 0).all(), "points must come sorted by (column, row), without duplicates"
            gathered[int(c["c"])] = key
    assert len(gathered) == len(oracle_clusters) and len(gathered) >= 3
    # every published cell that carries a cluster id must be a gathered member of exactly that cluster, and vice versa
    st = e.state()
    hi = st["first_unpublished_global_column_index"] - 1
    lo = max(st["ring_buffer_start_global_column_index"], 0)
    cols = e.read_columns(lo, hi, fields=["id"])
    ids = cols["id"].reshape(hi - lo + 1, R)
    for cid, key in gathered.items():
        inside = key[(key // R >= lo) & (key // R <= hi)]
        if len(inside):
            assert (ids[inside // R - lo, inside % R] == cid).all()
        if len(key) > 5:  # clusters of at most 5 points keep id 0 (cc.cpp:936)
            published = np.argwhere(ids == cid)
            assert len(published) <= len(key)
            assert np.isin((published[:, 0] + lo) * R + published[:, 1], key).all()
