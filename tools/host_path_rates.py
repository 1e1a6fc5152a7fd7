import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from continuous_clustering_amd import Engine, capi, synth
cfg = capi.Config.kitti()
st = synth.make_stream(2200*6, seed=5, motion=synth.Motion.translate())
e = Engine(cfg, 64)
e.add_firings(st.xyz[:2200], st.intensity[:2200], st.poses[:2200]); e.drain_events()
for chunk in (2200, 550, 64, 8, 1):
    t0=time.perf_counter(); n=0
    f=2200
    while f+chunk <= 2200*(3 if chunk>=64 else 2) and (chunk>=64 or n<1500):
        e.add_firings(st.xyz[f:f+chunk], st.intensity[f:f+chunk], st.poses[f:f+chunk]); f+=chunk; n+=1
    dt=time.perf_counter()-t0
    print(f"host path single stream chunk={chunk}: {n*chunk*64/dt/1e6:.2f} Mcells/s, {dt/n*1e6:.0f} us per call ({dt/(n*chunk)*1e6:.1f} us per firing)")
    e.drain_events()
e.record_events(False)
for chunk in (2200,):
    t0=time.perf_counter(); n=0; f=2200*3
    while f+chunk <= 2200*6:
        e.add_firings(st.xyz[f:f+chunk], st.intensity[f:f+chunk], st.poses[f:f+chunk]); f+=chunk; n+=1
    dt=time.perf_counter()-t0
    print(f"host path single stream chunk={chunk} (events off): {n*chunk*64/dt/1e6:.2f} Mcells/s, {dt/n*1e3:.2f} ms per call")
