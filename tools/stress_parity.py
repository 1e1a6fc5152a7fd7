"""Randomised parity sweep (GPU box): many small scenes with lots of short-lived trees, odd chunkings, both sensors, both
association kernels. Engine vs oracle through tests/util.run_and_compare. Usage: python tools/stress_parity.py [n_cases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import util, cases
from continuous_clustering_amd import synth
from continuous_clustering_amd.synth import Motion, SceneModel
from oracle import pyoracle
pyoracle.build()

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for i in range(n_cases):
    rng = np.random.default_rng(seed0 + i)
    s128 = (i % 4) == 3
    cols = int(rng.choice([240, 360, 480, 720]))
    sensor = cases._s128(cols) if s128 else cases._s64(cols)
    scene = SceneModel(n_objects=int(rng.integers(0, 200)), object_range=(2.0, float(rng.uniform(8, 30))),
                       object_radius=(0.05, float(rng.uniform(0.15, 1.0))), wall_radius=float(rng.choice([0.0, 10.0, 25.0, 45.0])),
                       wall_gaps_deg=() if rng.random() < 0.3 else ((20.0, 32.0), (140.0, 155.0)),
                       range_noise=float(rng.choice([0.01, 0.05, 0.2])), dropout=float(rng.choice([0.02, 0.1, 0.3])))
    motion = [Motion.static(), Motion.translate(float(rng.uniform(1, 20))), Motion.turn(8.0, float(rng.uniform(-1, 1)))][int(rng.integers(0, 3))]
    n = cols * int(rng.integers(2, 4)) + int(rng.integers(0, cols))
    stream = synth.make_stream(n, seed=seed0 + 17 * i, sensor=sensor, scene=scene, motion=motion)
    over = {}
    if rng.random() < 0.3:
        over["max_distance"] = float(rng.choice([0.4, 0.7, 1.2]))
    if rng.random() < 0.2:
        over["stop_after_association_enabled"] = 0
    cfg = (cases._vls if s128 else cases._kitti)(cols, **over)
    chunks = [int(c) for c in rng.choice([1, 3, 17, 64, 97, 250, cols, 2 * cols], size=4)]
    waves = [0, 4, 3, 1][i % 4]  # default (k_assoc3 + links wavefront), pinned four / three waves, k_assoc_lds
    batch = 0 if i % 7 == 6 else 1  # the batch-parallel kernel in front (default) or not
    rounds = [0, 2, 1, 3][i % 4]
    box = {}
    try:
        summ = util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=None,
                                    engine_setup=lambda e: (e.set_option("assoc_waves", waves), e.set_option("assoc_batch", batch),
                                                            e.set_option("assoc_rounds", rounds), box.setdefault("e", e)))
        es = summ["engine_state"]
        bc = box["e"].batch_counters()
        print(f"case {i:3d} ok: rows {sensor.num_rows} cols {cols} firings {n} objects {scene.n_objects} chunks {chunks} waves {waves} "
              f"clusters {summ['clusters']} serial columns {es['error_b']} batch {batch} rounds {rounds} columns {bc['batch_columns']} "
              f"bails {bc['batch_bails']} {bc['bail_reasons'][1:7]}", flush=True)
    except AssertionError as ex:
        bad += 1
        print(f"case {i:3d} FAILED (seed {seed0 + i}): {str(ex)[:300]}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
