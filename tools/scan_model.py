#!/usr/bin/env python3
"""Lane utilisation of the packed window scan (k_scan2), modelled from the ORACLE's per-point visit counts (Point::number_of_visited_neighbors,
cc.cpp:725) — CPU only. Why round 6 took the long scans out of k_scan2 (cc_k_scan.h: SCAN_CAP, k_scan2_long):

  one pass     a tile of 4 columns, its active points 64 at a time, every pass as long as its slowest point (round 2 - 5)
  split at K   k_scan2 spends at most K visits per point; the rest of a longer scan runs in k_scan2_long, where a lane that is done takes the
               next point (modelled as greedy list scheduling over the stream's long points)

usage: python tools/scan_model.py [rotations]    prints wave-iterations per column for the bench's street scene and its vegetation scene
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from continuous_clustering_amd import capi, synth  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402  (test infrastructure: this tool is not part of the product)

rot = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = capi.Config.kitti()
for name, scene, seed in (("street (bench headline)", None, 1234), ("vegetation (bench cluttered)", synth.SceneModel.cluttered(0.1), 4321)):
    kw = {"scene": scene} if scene is not None else {}
    st = synth.make_stream(2200 * rot, seed=seed, sensor=synth.SensorModel.s64(), **kw)
    o = Oracle(cfg, 64)
    assert o.add_firings(st.xyz, st.intensity, st.poses) == 0
    hi = o.state()["first_unpublished_global_column_index"] - 1
    cols = o.read_published(2200, hi)
    vis = cols["number_of_visited_neighbors"].astype(np.int64)
    act = ~cols["is_ignored"].astype(bool)
    n = vis.shape[0] // 4 * 4
    v, a = np.where(act, vis, 0)[:n], act[:n]
    va = v[a]
    print(f"{name}: {n} columns, active cells {a.mean():.3f}, visits per active point mean {va.mean():.1f} p50 {np.percentile(va, 50):.0f} "
          f"p90 {np.percentile(va, 90):.0f} p99 {np.percentile(va, 99):.0f} max {va.max()}")
    one = 0
    for t in range(0, n, 4):
        lst = v[t:t + 4].reshape(-1)[a[t:t + 4].reshape(-1)]
        for b in range(0, len(lst), 64):
            one += lst[b:b + 64].max()
    print(f"   one pass:   {one / n:6.2f} wave-iterations per column, {va.sum() / (64.0 * one):.3f} of the lanes busy (ideal {va.sum() / 64.0 / n:.2f} per column)")
    for K in (4, 6, 8, 16):
        it1 = 0
        for t in range(0, n, 4):
            lst = v[t:t + 4].reshape(-1)[a[t:t + 4].reshape(-1)]
            for b in range(0, len(lst), 64):
                it1 += min(lst[b:b + 64].max(), K)
        longp = (v > K) & a
        lanes = np.zeros(64, dtype=np.int64)
        for x in np.sort((v[longp] - K))[::-1]:
            lanes[lanes.argmin()] += x
        print(f"   split at {K:2d}: {it1 / n:6.2f} + {lanes.max() / n:5.2f} (long scans) per column; {100.0 * longp.sum() / a.sum():.2f} % of the points are long, "
              f"{100.0 * longp.any(axis=1).mean():.1f} % of the columns wait for k_scan2_epi")
