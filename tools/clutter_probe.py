"""Vegetation-like scenes (synth.SceneModel.cluttered) through engine and oracle: parity, how often and why k_assocb hands groups to the serial
kernel, the share of columns replayed serially. usage: [CLUTTER_KIND=mixed|sparse|near] python tools/clutter_probe.py [density ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import util
from continuous_clustering_amd import capi, synth
from oracle import pyoracle
pyoracle.build()
dens = [float(a) for a in sys.argv[1:]] or [0.15, 0.3, 0.45, 0.7]
cols = int(os.environ.get("CLUTTER_COLS", "2200"))
for d in dens:
    cfg = capi.Config.kitti(); cfg.num_columns = cols
    sensor = synth.SensorModel(num_rows=64, num_columns=cols)
    stream = synth.make_stream(cols * 2 + 300, seed=77, sensor=sensor, scene=(lambda dd: synth.SceneModel(clutter=tuple((float(a), float(b), float(c), float(e), dd) for a, b, c, e in [t.split(',') for t in os.environ['CLUTTER_SPEC'].split(';')]))) (d) if os.environ.get('CLUTTER_SPEC') else {'mixed': synth.SceneModel.cluttered, 'sparse': synth.SceneModel.sparse_clutter, 'near': synth.SceneModel.near_clutter}[os.environ.get('CLUTTER_KIND', 'mixed')](d), motion=synth.Motion.translate())
    for chunks in ([cols], [97, 1, 200]):
        box = {}
        t0 = time.time()
        try:
            summ = util.run_and_compare(stream, cfg, chunks=chunks, engine_setup=lambda e: box.__setitem__("e", e))
            bc = box["e"].batch_counters(); es = summ["engine_state"]
            print(f"density {d:4.2f} chunks {str(chunks):14s} ok  published {summ['published_columns']:6d} clusters {summ['clusters']:5d} batch columns {bc['batch_columns']:6d} "
                  f"bails {bc['batch_bails']:4d} {bc['bail_reasons'][1:7]} serial columns {es['error_b']:5d}  {time.time() - t0:.1f}s", flush=True)
        except AssertionError as ex:
            bc = box["e"].batch_counters() if "e" in box else {}
            print(f"density {d:4.2f} chunks {chunks} MISMATCH {bc} {str(ex)[:300]}", flush=True)
