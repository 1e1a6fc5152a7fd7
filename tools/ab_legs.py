#!/usr/bin/env python3
"""Same-box A/B of several builds of libcc_hip over several legs of bench.py (GPU box, through gpurun).

usage: python tools/ab_legs.py [--legs headline,few32,steady32,cluttered,s128] [--reps 3] [--steps 40] [--env "A=1 B=2"] lib1.so lib2.so ...
Every (lib, rep) is one subprocess (the library is chosen at import time); the repetitions alternate over the builds, so that box drift hits all
of them alike. One line per (lib, rep) with the legs' Mpoints/s; at the end min / median / max per lib and leg.
Legs: headline = 256 x S64, --steps timed steps (default 40); few32 = 32 streams in the driver's 20-step shape; steady32 = 32 streams, 60 steps;
cluttered = 256 vegetation-like streams, 12 steps; s128 = 256 x S128, 10 steps; few64 / steady64 likewise.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, json
sys.path.insert(0, %(root)r)
import continuous_clustering_amd as c
c.LIB_PATH = os.path.join(os.path.dirname(c.LIB_PATH), %(lib)r)
import torch
import bench
from continuous_clustering_amd import capi, synth
legs = %(legs)r
steps = %(steps)d
dev = torch.device("cuda", 0)
ctx = bench.Ctx(torch, None, False, 1, 0, dev, 0, False)
out = {}
s64, cfg64 = synth.SensorModel.s64(), capi.Config.kitti()
F = 2200
def thr(sensor, cfg, seeds, F, k, warm, inputs=None, options=None):
    r, e, own = bench.run_throughput(ctx, sensor, cfg, seeds, F, k, warm, 0, inputs=inputs, options=options)
    e.close()
    return r
if "headline" in legs:
    r = thr(s64, cfg64, [1234 + j for j in range(256)], F, steps, 3)
    out["headline"] = r["value"]; out["headline_kernels"] = {k: round(v, 3) for k, v in r["kernel_ms_per_step"].items()}
    torch.cuda.empty_cache()
for n in (32, 64):
    if "few%%d" %% n in legs:
        out["few%%d" %% n] = thr(s64, cfg64, [1234 + j for j in range(n)], F, 20, 3)["value"]
    if "steady%%d" %% n in legs:
        out["steady%%d" %% n] = thr(s64, cfg64, [1234 + j for j in range(n)], F, 60, 3)["value"]
for n in (32, 64, 128):
    if "cluttered%%d" %% n in legs:
        inputs = bench.gen_inputs(torch, dev, s64, [4321 + j for j in range(n)], F, 33, scene=synth.SceneModel.cluttered(0.1))
        out["cluttered%%d" %% n] = thr(s64, cfg64, list(range(n)), F, 30, 3, inputs=inputs)["value"]
        del inputs; torch.cuda.empty_cache()
if "cluttered" in legs:
    inputs = bench.gen_inputs(torch, dev, s64, [4321 + j for j in range(256)], F, 15, scene=synth.SceneModel.cluttered(0.1))
    r = thr(s64, cfg64, list(range(256)), F, 12, 3, inputs=inputs)
    out["cluttered"] = r["value"]; out["cluttered_kernels"] = {k: round(v, 3) for k, v in r["kernel_ms_per_step"].items()}
    del inputs; torch.cuda.empty_cache()
if "s128" in legs:
    out["s128"] = thr(synth.SensorModel.s128(), capi.Config.vls128(), [1234 + j for j in range(256)], 1700, 10, 3)["value"]
print("ABLEGS " + json.dumps(out))
'''


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", default="headline")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--env", default="", help='environment for every child, e.g. "CC_ENABLE_ENV_OPTS=1 CC_INSERT_WIDE=256"')
    ap.add_argument("libs", nargs="+", help="library file names under continuous_clustering_amd/; 'name.so@A=1,B=2' adds per-variant environment")
    a = ap.parse_args()
    legs = a.legs.split(",")
    res = {}
    for rep in range(a.reps):
        for spec in a.libs:
            lib, _, envs = spec.partition("@")
            env = dict(os.environ)
            for kv in (a.env.split() + (envs.split(",") if envs else [])):
                k, _, v = kv.partition("=")
                env[k] = v
            if envs or a.env:
                env["CC_ENABLE_ENV_OPTS"] = "1"
            code = CHILD % {"root": ROOT, "lib": lib, "legs": legs, "steps": a.steps}
            try:
                p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
            except subprocess.TimeoutExpired:
                print(spec, "rep", rep, "TIMEOUT", flush=True)
                continue
            line = [l for l in p.stdout.splitlines() if l.startswith("ABLEGS ")]
            if not line:
                print(spec, "rep", rep, "FAILED", p.stderr[-400:].replace("\n", " | "), flush=True)
                continue
            d = json.loads(line[-1][7:])
            print(spec, "rep", rep, {k: (round(v) if isinstance(v, float) else v) for k, v in d.items()}, flush=True)
            for k, v in d.items():
                if isinstance(v, float):
                    res.setdefault(spec, {}).setdefault(k, []).append(v)
    print("---- min / median / max")
    for spec, legsd in res.items():
        print(spec, {k: (round(min(v)), round(statistics.median(v)), round(max(v))) for k, v in legsd.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
