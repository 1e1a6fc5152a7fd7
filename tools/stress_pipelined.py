"""Pipelined device path under perturbed inputs: S streams with duplicated / empty / backwards / skipped / multi-column firings, fed
through cc_engine_add_firings_device (events off: three chains, parallel insertion, publish off-chain), against one oracle per stream.
usage: python tools/stress_pipelined.py [rounds] [streams] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import util
from continuous_clustering_amd import Engine, capi
from oracle.pyoracle import Oracle
from test_gpu_parallel_insert import perturbed_stream

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
cfg = capi.Config.kitti()
bad = 0
for r in range(rounds):
    streams = [perturbed_stream(seed0 + 100 * r + s) for s in range(S)]
    n = min(st.n_firings for st in streams)
    F = int(np.random.default_rng(seed0 + r).choice([700, 1100, 2200]))
    NB = n // F
    e = Engine(cfg, 64, S); e.record_events(False)
    for kv in filter(None, os.environ.get("CC_STRESS_OPTS", "").split(",")):  # e.g. CC_STRESS_OPTS=lazy_gate=0
        e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    alive = []  # the calls are asynchronous: a batch's buffers stay untouched until the engine has been synchronised (include/cc_hip.h)
    for b in range(NB):
        if os.environ.get("CC_STRESS_FREE_INPUTS") != "1":
            alive.append((xyz, inten, poses) if b else None)
        xyz = torch.from_numpy(np.stack([st.xyz[b * F:(b + 1) * F] for st in streams])).cuda()
        inten = torch.from_numpy(np.stack([st.intensity[b * F:(b + 1) * F] for st in streams])).cuda()
        poses = torch.from_numpy(np.stack([st.poses[b * F:(b + 1) * F] for st in streams])).cuda()
        torch.cuda.synchronize()
        e.add_firings_device(F, xyz.data_ptr(), inten.data_ptr(), poses.data_ptr())
    assert e.sync() == 0, e.last_error()
    for s in range(S):
        o = Oracle(cfg, 64)
        assert o.add_firings(streams[s].xyz[:NB * F], streams[s].intensity[:NB * F], streams[s].poses[:NB * F]) == 0
        so, se = o.state(), e.state(s)
        ok = all(so[k] == se[k] for k in util.STATE_FIELDS)
        if ok:
            hi = se["first_unpublished_global_column_index"] - 1
            lo = max(se["ring_buffer_start_global_column_index"], hi - 1200)
            try:
                util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s), lo, mirror=False)  # (events off: no mirror-only fields)
            except AssertionError as ex:
                ok = False; print("  columns:", str(ex)[:200])
        if not ok:
            bad += 1; print(f"round {r} stream {s}: MISMATCH", {k: (so[k], se[k]) for k in util.STATE_FIELDS if so[k] != se[k]})
    print(f"round {r}: {S} streams x {NB} calls of {F} firings ok" if bad == 0 else f"round {r}: failures so far {bad}", e.batch_counters(), e.totals())
    e.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
