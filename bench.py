#!/usr/bin/env python3
"""bench.py — throughput of the continuous-clustering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the throughput configuration of the metric "Mpoints/s clustered (64-beam
stream)"): 256 concurrent synthetic S64 sensor streams per GPU (64 rows x 2200 columns per rotation, KITTI
parameters of src/tools/kitti_demo.cpp:279-294, seeded scenes of SURVEY.md 8d, sensor translating at 10 m/s).
One *step* = one pass of the hot path (insertion -> ground segmentation -> association / union-find ->
finished-cluster check -> publish) over one batch = one rotation (2200 firings) of every stream. Inputs are
generated directly in HBM before the timed region. value = published range-image cells per second (NaN cells
included) over all GPUs, in Mpoints/s. Streams never interact, so ranks share nothing on the data path
(weak scaling: 256 streams per GPU); the only collective is the gather of per-rank result counts at the end
(RCCL when launched under torch.distributed.run, also at world size 1: `rccl_world` in the JSON line).

The JSON line also carries
  roofline          dominant kernel vs the 8 TB/s HBM roof: algorithmic bytes per launch (19.5 B per cell at 64 rows,
                    SURVEY 8d) / its average duration measured with HIP events on the engine's stream
  verified_streams  after the timed region (outside it) the whole input of a few of the 256 streams is replayed through the
                    CPU oracle; stream state and the last 1500 published columns must be bit-equal or the bench fails
  s128              the same measurement for BASELINE.json configs[3] (128 rows x 1700 columns, VLS-128 firing shape with
                    per-laser azimuth offsets, library defaults), fewer steps; the headline stays S64
  cpu_baseline      the CPU oracle (oracle/, a restatement of the reference's single-threaded path) timed on this box's
                    host cores: BASELINE.md mode C = N independent single-threaded instances, one PROCESS each, pinned to
                    distinct physical cores, oracle constructed inside the pinned worker, >= 20 rotations per instance,
                    N swept over {1, 16, 64, all physical cores}; mode A = per-call addFiring latency p50/p99 on one core
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

STATE_FIELDS = ("reset_required", "ring_buffer_start_global_column_index", "ring_buffer_end_global_column_index",
                "first_unfinished_global_column_index", "first_unpublished_global_column_index", "cluster_counter",
                "firings_consumed", "cells_published", "clusters_finished", "n_unfinished_trees")


MIRROR_ONLY_FIELDS = ("number_of_visited_neighbors", "finished_at_continuous_azimuth_angle", "tree_num_points", "cluster_width")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)  # (a timed region fills and drains a three-deep pipeline once: 40 steps read 5 % low)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3, help="repeats of the headline leg inside one run: the line carries the median and the spread")
    ap.add_argument("--streams", type=int, default=256, help="sensor streams per GPU")
    ap.add_argument("--firings", type=int, default=2200, help="firings per stream per step (2200 = one rotation)")
    ap.add_argument("--sensor", default="s64", choices=["s64", "s128"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-s128", action="store_true")
    ap.add_argument("--verify-streams", type=int, default=3)
    ap.add_argument("--cpu-procs", type=int, default=0, help="largest instance count of the CPU sweep (0 = all physical cores)")
    ap.add_argument("--cpu-rotations", type=int, default=20)
    ap.add_argument("--total-streams", type=int, default=256,
                    help="strong-split leg (north_star: '256 concurrent streams' over the node): this many streams in total, dealt stream s -> rank s mod N")
    ap.add_argument("--no-strong-split", action="store_true")
    ap.add_argument("--no-few-streams", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true")
    ap.add_argument("--no-cluttered", action="store_true")
    ap.add_argument("--kitti-root", default=os.environ.get("SEMANTIC_KITTI_ROOT", ""),
                    help="SemanticKITTI dataset root (the directory that holds sequences/): adds the real-data acceptance leg (README.md:213-245)")
    ap.add_argument("--kitti-sequences", default="0,1,2,3,4,5,6,7,8,9,10")
    ap.add_argument("--kitti-max-frames", type=int, default=0, help="0 = whole sequences")
    ap.add_argument("--stub-engine", action="store_true",
                    help="TEST ONLY (tests/test_bench_launcher.py): tests/bench_stub.py instead of the HIP engine, gloo on CPU; exercises the launcher and the "
                         "aggregation over ranks, the line says data = 'stub' and is not a measurement")
    return ap.parse_args()


def pin_to_gpu_numa_node(torch, dev_index):
    """N ranks share one host: every step has a host thread wait for its GPU's insertion chain (cc_engine.hip: the gate) and launch a dozen
    kernels, so the rank's threads belong on the cores of the NUMA node its GPU hangs off (sysfs: numa_node / local_cpulist of the PCI device).
    Returns what was done (for the bench line), None where sysfs does not say."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(base + "/numa_node").read().strip())
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        cpus &= os.sched_getaffinity(0)
        if node < 0 or not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "cpus": len(cpus)}
    except Exception:  # noqa: BLE001
        return None


def maybe_spawn(args):
    """`python bench.py --gpus N` without a launcher around it starts the N ranks itself: re-exec under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1). Under a launcher (RANK set) the world is what the launcher says."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def gen_inputs(torch, dev, sensor, seeds, n_firings, n_batches, scene=None):
    """[batch][stream][firing][row][3] etc., generated in HBM. Stream k of the engine is the seeded scene seeds[k]."""
    from continuous_clustering_amd import synth
    R = sensor.num_rows
    n_streams = len(seeds)
    xyz = torch.empty((n_batches, n_streams, n_firings, R, 3), dtype=torch.float32, device=dev)
    inten = torch.empty((n_batches, n_streams, n_firings, R), dtype=torch.uint8, device=dev)
    poses = torch.empty((n_batches, n_streams, n_firings, 12), dtype=torch.float64, device=dev)
    for s, seed in enumerate(seeds):
        st = synth.make_stream(n_firings * n_batches, seed=seed, sensor=sensor, scene=scene, motion=synth.Motion.translate(10.0),
                               start_column=40 if sensor.azimuth_offsets_deg else 0, xp=torch, device=dev, chunk=n_firings)
        xyz[:, s] = st.xyz.view(n_batches, n_firings, R, 3)
        inten[:, s] = st.intensity.view(n_batches, n_firings, R)
        poses[:, s] = st.poses.view(n_batches, n_firings, 12)
    return xyz, inten, poses


# ---------------------------------------------------------------------------------------------------------------------------
# verification against the oracle (outside the timed region)
# ---------------------------------------------------------------------------------------------------------------------------
def _bits_equal(a, b):
    if a.dtype.kind == "f":
        an, bn = np.isnan(a), np.isnan(b)
        if not np.array_equal(an, bn):
            return False
        it = np.uint32 if a.dtype == np.float32 else np.uint64
        return np.array_equal(a[~an].view(it), b[~bn].view(it))
    return np.array_equal(a, b)


def verify_against_oracle(eng, cfg, R, xyz, inten, poses, which, tail_cols=1500):
    """Replay the complete input of the streams `which` through the CPU oracle and require the engine's stream state and the last
    `tail_cols` published columns (every field of the column view: geometry bit patterns, labels, ignore flags, tree roots, raw
    cluster ids) to be equal. Raises SystemExit on any difference: a fast run with different results is not a result."""
    from oracle.pyoracle import Oracle
    nb, _, F = xyz.shape[0], xyz.shape[1], xyz.shape[2]
    t0 = time.perf_counter()
    for s in which:
        hx = xyz[:, s].reshape(nb * F, R, 3).cpu().numpy()
        hi = inten[:, s].reshape(nb * F, R).cpu().numpy()
        hp = poses[:, s].reshape(nb * F, 12).cpu().numpy()
        o = Oracle(cfg, R, record=True)
        o.keep_published_tail(tail_cols + 256)
        rc = o.add_firings(hx, hi, hp)
        if rc != 0:
            raise SystemExit(f"verify: oracle rc {rc} on stream {s}: {o.last_error()}")
        so, se = o.state(), eng.state(s)
        for k in STATE_FIELDS:
            if so[k] != se[k]:
                raise SystemExit(f"verify: stream {s} state.{k}: oracle {so[k]} engine {se[k]}")
        hi_col = se["first_unpublished_global_column_index"] - 1
        lo_col = max(hi_col - tail_cols + 1, se["ring_buffer_start_global_column_index"], o.published_range()[0])
        ao, ae = o.read_published(lo_col, hi_col), eng.read_columns(lo_col, hi_col, stream=s)
        for f in ao:
            if f in MIRROR_ONLY_FIELDS:
                continue  # (only produced with the engine option "mirror_fields": off in the throughput mode measured here)
            if not _bits_equal(ao[f], ae[f]):
                raise SystemExit(f"verify: stream {s} columns [{lo_col}, {hi_col}] field {f} differs from the oracle")
    return {"streams": [int(s) for s in which], "columns_compared_per_stream": int(tail_cols), "rotations_replayed": int(nb * F // cfg.num_columns),
            "seconds": round(time.perf_counter() - t0, 2)}


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline: BASELINE.md 3 modes A and C on the host cores of this box
# ---------------------------------------------------------------------------------------------------------------------------
def physical_cores():
    """One logical CPU per physical core (first SMT sibling), restricted to what this process may run on."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, cores = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            cores.append(c)
    return cores


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_worker(idx, core, cfg, R, hx, hi, hp, barrier, conn, latency):
    try:
        os.sched_setaffinity(0, {core})
        from oracle.pyoracle import Oracle
        # private, first-touched-here copies of the inputs; the oracle's ring is allocated (and touched by reset) inside this pinned process
        x, i, p = np.array(hx, copy=True), np.array(hi, copy=True), np.array(hp, copy=True)
        o = Oracle(cfg, R, record=False)
        buf_a, buf_b = np.ones(1 << 24, dtype=np.float64), np.ones(1 << 24, dtype=np.float64)  # 128 MB each, touched here: beyond every cache
        barrier.wait(timeout=600)
        sec = o.time_firings(x, i, p)
        cells = o.state()["cells_published"]
        lat = None
        if latency:
            o2 = Oracle(cfg, R, record=False)
            ns = o2.time_each_firing(x, i, p)[2 * cfg.num_columns:]  # steady state: skip the first two rotations
            lat = (float(np.percentile(ns, 50)), float(np.percentile(ns, 99)), float(ns.mean()))
        barrier.wait(timeout=600)
        t0 = time.perf_counter()
        for _ in range(3):
            np.copyto(buf_b, buf_a)
        bw = 3 * 2 * buf_a.nbytes / (time.perf_counter() - t0) / 1e9  # read + write
        conn.send((idx, cells, sec, lat, bw))
    except Exception as ex:  # noqa: BLE001
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass
        conn.send((idx, 0, -1.0, None, 0.0, repr(ex)))
    finally:
        conn.close()


def cpu_mode_c(cfg, R, hx, hi, hp, cores):
    """len(cores) independent single-threaded oracle instances, one process each, pinned; all start their timed addFiring loop at one
    barrier; aggregate rate = cells of all instances / slowest instance."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")  # children never touch HIP; they inherit the host arrays without a copy
    n = len(cores)
    barrier = ctx.Barrier(n)
    procs, conns = [], []
    for k, core in enumerate(cores):
        a, b = ctx.Pipe(duplex=False)
        pr = ctx.Process(target=_cpu_worker, args=(k, core, cfg, R, hx[k % len(hx)], hi[k % len(hi)], hp[k % len(hp)], barrier, b, k == 0 and n == 1))
        pr.start()
        b.close()
        procs.append(pr)
        conns.append(a)
    res = [c.recv() for c in conns]
    for pr in procs:
        pr.join()
    bad = [r for r in res if r[2] <= 0]
    if bad:
        raise RuntimeError(f"cpu baseline worker failed: {bad[0]}")
    cells = sum(r[1] for r in res)
    slowest = max(r[2] for r in res)
    return {"value": cells / slowest / 1e6, "cells": cells, "cpu_seconds": sum(r[2] for r in res), "slowest_s": slowest,
            "fastest_s": min(r[2] for r in res), "latency_ns": res[0][3], "membw_GBs": sum(r[4] for r in res)}


def _cpu_mode_b_worker(cores, cfg, R, hx, hi, hp, conn):
    try:
        os.sched_setaffinity(0, set(cores))
        from oracle.pyoracle import Oracle
        x, i, p = np.array(hx, copy=True), np.array(hi, copy=True), np.array(hp, copy=True)
        o = Oracle(cfg, R, record=False)
        sec = o.time_firings_pipeline(x, i, p)
        conn.send((o.state()["cells_published"], sec, o.last_error() if sec < 0 else ""))
    except Exception as ex:  # noqa: BLE001
        conn.send((0, -1.0, repr(ex)))
    finally:
        conn.close()


def cpu_mode_b(cfg, R, hx, hi, hp, cores):
    """BASELINE.md mode B: ONE instance, its stages on threads connected by bounded queues with back-pressure (oracle/cc_oracle.cpp: Oracle::Pipe —
    the three-thread, race-free part of the reference's five-stage pipeline, cc.cpp:49-63), pinned to three physical cores."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    a, b = ctx.Pipe(duplex=False)
    pr = ctx.Process(target=_cpu_mode_b_worker, args=(cores[:3], cfg, R, hx, hi, hp, b))
    pr.start()
    b.close()
    cells, sec, err = a.recv()
    pr.join()
    if sec <= 0:
        return {"value": None, "error": err}
    return {"value": cells / sec / 1e6, "unit": "Mpoints/s", "threads": 3, "cores": len(cores[:3]), "seconds": sec,
            "note": "one oracle instance, stages handed from thread to thread: insertion | segmentation | association + tree combination + publishing "
                    "(bounded queues, producer back-pressure). The reference's own pipeline has five stages on seven threads that share the range image "
                    "and the tree lists without locks; the oracle keeps the single-threaded data structures and runs the three stages that are free "
                    "of data races on them. Same results as single-threaded (tests/test_oracle_properties.py)"}


def cpu_baseline_report(cfg, sensor, xyz, inten, poses, S, F, args, sweep_sizes=(1, 16, 64)):
    R = sensor.num_rows
    cores = physical_cores()
    nmax = min(args.cpu_procs or len(cores), len(cores))
    nb = min(xyz.shape[0], args.cpu_rotations)
    n_inputs = min(S, nmax, 16)  # distinct streams replayed (instances beyond that re-use them: same work per instance)
    hx = [xyz[:nb, k].reshape(nb * F, R, 3).cpu().numpy() for k in range(n_inputs)]
    hi = [inten[:nb, k].reshape(nb * F, R).cpu().numpy() for k in range(n_inputs)]
    hp = [poses[:nb, k].reshape(nb * F, 12).cpu().numpy() for k in range(n_inputs)]
    sweep, best, best_n, single, lat = {}, None, 1, None, None
    for n in sorted({*sweep_sizes, nmax}):
        if n > nmax:
            continue
        r = cpu_mode_c(cfg, R, hx, hi, hp, cores[:n])
        sweep[str(n)] = {"mpoints_per_s": round(r["value"], 2), "per_instance": round(r["value"] / n, 2),
                         "slowest_s": round(r["slowest_s"], 3), "fastest_s": round(r["fastest_s"], 3),
                         "concurrent_copy_GBs": round(r["membw_GBs"], 1)}
        if n == 1:
            single, lat = r["value"], r["latency_ns"]
        if best is None or r["value"] > best["value"]:
            best, best_n = r, n
    out = {
        "value": best["value"], "unit": "Mpoints/s", "cores": best_n, "kind": "port",
        "sample": f"BASELINE.md mode C: {best_n} independent single-threaded oracle instances (one process each, pinned to distinct "
                  f"physical cores, ring allocated inside the pinned worker), each replaying {nb} rotations ({nb * F} firings) of one of the "
                  f"bench's own S{R} streams once; {best['cells']} published cells, {best['cpu_seconds']:.1f} CPU-seconds in the timed "
                  f"addFiring loops; rate = all cells / slowest instance",
        "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "physical_cores": len(cores), "rotations_per_instance": nb,
        "single_core_value": single, "sweep": sweep,
    }
    try:
        out["mode_b"] = cpu_mode_b(cfg, R, hx[0], hi[0], hp[0], cores)
        if out["mode_b"].get("value") and single:
            out["mode_b"]["vs_single_thread"] = out["mode_b"]["value"] / single
    except Exception as ex:  # noqa: BLE001
        out["mode_b"] = {"value": None, "error": repr(ex)}
    if lat:
        out["mode_a_latency_us_per_column"] = {"p50": lat[0] / 1e3, "p99": lat[1] / 1e3, "mean": lat[2] / 1e3,
                                                "note": "one instance, one core: steady_clock around every addFiring call (1 firing = 1 column), "
                                                        "first two rotations skipped"}
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# GPU throughput leg
# ---------------------------------------------------------------------------------------------------------------------------
def engine_options(eng):
    for env, opt in (("CC_SUB_BATCH", "sub_batch"), ("CC_TABLE_EARLY", "table_on_insert_chain"), ("CC_PIPELINE", "pipeline"),
                     ("CC_PUBLISH_OFF_CHAIN", "publish_off_chain"), ("CC_PARALLEL_INSERT", "parallel_insert"), ("CC_SCAN_PACKED", "scan_packed"), ("CC_SCAN_SPLIT", "scan_split"), ("CC_SCAN_CAP", "scan_cap"), ("CC_EGO_OFF_CHAIN", "ego_off_chain"),
                     ("CC_SKIP_FALLBACKS", "skip_idle_fallbacks"), ("CC_ASSOC_ROUNDS", "assoc_rounds"), ("CC_ASSOC_BATCH", "assoc_batch"),
                     ("CC_EGO_EARLY", "ego_on_insert_chain"), ("CC_ASSOC_COOLDOWN", "assoc_cooldown"), ("CC_SWEEP_BLOCKS", "assoc_sweep_blocks"), ("CC_INSERT_WIDE", "insert_wide_max_streams"), ("CC_INSERT_SPLIT", "insert_split_blocks"), ("CC_DEBUG_NO_ASSOC_FALLBACK", "debug_no_assoc_fallback"), ("CC_FUSE_FRONT", "fuse_front"), ("CC_DEFER_TAIL", "defer_tail_max_streams"), ("CC_LAZY_GATE", "lazy_gate"), ("CC_LAZY_GATE_FROM", "lazy_gate_from"), ("CC_INSERT_LDS_PAD", "insert_lds_pad"), ("CC_INSERT_NARROW", "insert_narrow_blocks")):
        if os.environ.get(env) not in (None, ""):
            eng.set_option(opt, int(os.environ[env]))


class Ctx:
    """What every leg needs to know about the process: torch, the process group (RCCL; gloo with --stub-engine), rank / world, the device."""

    def __init__(self, torch, dist, use_dist, world, rank, dev, local_rank, stub):
        self.torch, self.dist, self.use_dist, self.world, self.rank, self.dev, self.local_rank, self.stub = \
            torch, dist, use_dist, world, rank, dev, local_rank, stub

    def device_sync(self):
        if self.dev.type == "cuda":
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.use_dist:
            self.dist.barrier()

    def engine(self, cfg, R, S):
        if self.stub:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import bench_stub
            return bench_stub.StubEngine(cfg, R, S, rank=self.rank)
        from continuous_clustering_amd import Engine
        return Engine(cfg, R, S, device=self.local_rank)


def run_throughput(ctx, sensor, cfg, seeds, F, steps, warmup, n_verify, inputs=None, options=None):
    """One throughput leg: this rank's engine holds len(seeds) streams; `steps` timed passes of the hot path over one batch (F firings of
    every stream) each, bracketed by barrier + device sync on both sides. Returns the whole-job figures (cells of all ranks / slowest rank)."""
    torch, dist = ctx.torch, ctx.dist
    R, S = sensor.num_rows, len(seeds)
    n_batches = warmup + steps
    xyz, inten, poses = inputs if inputs is not None else gen_inputs(torch, ctx.dev, sensor, seeds, F, n_batches)
    ctx.device_sync()
    eng = ctx.engine(cfg, R, S)
    eng.record_events(False)
    engine_options(eng)
    for name, val in (options or {}).items():
        eng.set_option(name, val)

    for b in range(warmup):
        eng.add_firings_device(F, xyz[b], inten[b], poses[b])
    rc = eng.sync()
    if rc != 0:
        raise SystemExit(f"engine error {rc}: {eng.last_error()}")
    before = eng.totals()
    # per-kernel HIP events inside the timed region: on every 4th batch (three event records in a row on the association stream cost 15 - 25 us of
    # that chain per batch — 3.5 - 5 % of a step at 32 - 64 streams, 1 % at 256, profiles/r06_ab_timing.txt); the engine scales the sampled sums to
    # all batches. CC_BENCH_TIMING_EVERY=1 brackets every batch as the earlier rounds did, CC_BENCH_TIMING=0 none (tools/ab_legs.py experiments).
    eng.set_option("timing_every", int(os.environ.get("CC_BENCH_TIMING_EVERY", "4")))
    eng.enable_timing(os.environ.get("CC_BENCH_TIMING", "1") != "0")

    ctx.device_sync()
    ctx.barrier()
    t0 = time.perf_counter()
    for b in range(warmup, n_batches):
        eng.add_firings_device(F, xyz[b], inten[b], poses[b])
    rc = eng.sync()
    ctx.device_sync()
    own_elapsed = time.perf_counter() - t0
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    if rc != 0:
        raise SystemExit(f"engine error {rc}: {eng.last_error()}")
    after = eng.totals()
    ktimes = eng.kernel_times()
    eng.enable_timing(False)

    cells = after["cells_published"] - before["cells_published"]
    clusters = after["clusters_finished"] - before["clusters_finished"]
    per_rank = [{"rank": 0, "streams": S, "cells": int(cells), "seconds": own_elapsed, "value": cells / own_elapsed / 1e6}]
    # the one exchange step of the path: gather per-rank result counts (RCCL over xGMI), max of the elapsed times
    if ctx.use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([cells, clusters, after["serial_columns"], S, int(own_elapsed * 1e9)], dtype=torch.int64, device=ctx.dev)
        gathered = [torch.zeros_like(cnt) for _ in range(ctx.world)]
        dist.all_gather(gathered, cnt)
        cells = int(sum(int(g[0]) for g in gathered))
        clusters = int(sum(int(g[1]) for g in gathered))
        per_rank = [{"rank": r, "streams": int(g[3]), "cells": int(g[0]), "seconds": int(g[4]) / 1e9,
                     "value": int(g[0]) / max(int(g[4]) / 1e9, 1e-12) / 1e6} for r, g in enumerate(gathered)]

    verified = None
    if n_verify > 0:
        which = sorted({(k * (S - 1)) // max(1, n_verify - 1) for k in range(n_verify)}) if n_verify > 1 else [0]
        verified = verify_against_oracle(eng, cfg, R, xyz, inten, poses, which)

    alg_bytes_per_cell = 18.0 + 96.0 / R  # 13 B read + 5 B written per cell + 96 B pose per column
    batches = max(1, ktimes["batches"])  # kernel launches of each kind: the engine may cut a step into pipelined sub-batches
    launches_per_step = batches / max(1, steps)
    per_kernel = {k: v / max(1, steps) for k, v in ktimes.items() if k.endswith("_ms")}  # ms per step
    # dominant single kernel (segment_ms is the sum of k_table + k_seg_pre + k_seg_scan, profiles/ lists them separately;
    # prep_ms covers k_insert_par — preparation fused with the block-parallel insertion — plus k_prep of what it left over;
    # insert_ms is the serial kernel k_insert2 behind it; above 64 rows the block-parallel kernel is k_insert_multi)
    # assoc_lds_ms is k_assocb alone (the batch-parallel association); assoc_global_ms the serial kernels launched behind it (k_assoc3, k_associate:
    # in steady state they find nothing to do)
    KERNEL_OF = {"prep_ms": "k_insert_par" if R <= 64 else "k_insert_multi", "insert_ms": "k_insert2", "scan_ms": "k_scan2" if (R > 64 or S > 192 or os.environ.get("CC_SCAN_PACKED") == "1") and os.environ.get("CC_SCAN_PACKED") != "0" else "k_scan", "assoc_lds_ms": "k_assocb",
                 "assoc_global_ms": "k_assoc3", "publish_ms": "k_publish"}
    # Which kernel is "dominant": the one with the longest average launch in the committed rocprofv3 summary of this round (profiles/), so that the
    # line names the same kernel in every run and its frac can be re-derived from profiles/ (the HIP-event durations of kernels that overlap on
    # four chains wander by +- 20 % between runs of equal throughput, and at 128 rows two kernels are within that of each other); without the
    # file, the longest HIP-event duration of this run. `launch_ms` is always this run's own HIP-event figure for that kernel.
    committed = {}
    spath = ""
    for tag in ("r06_final", "r05_final", "r04_final", "r03_final"):  # (the newest committed summary of this build's round)
        cand = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv" if R == 64 else f"{tag}_kernel_stats_s128.csv")
        if os.path.exists(cand):
            spath = cand
            break
    if spath:
        try:
            import csv
            calls = {}
            for row in csv.reader(open(spath)):
                for key, kern in KERNEL_OF.items():
                    if ("cck::" + kern + "<") in row[0] or ("cck::" + kern + "(") in row[0]:
                        committed[key] = float(row[3]) / 1e6
                        calls[key] = int(row[1])
            # (only kernels of a steady-state step: the serial insertion kernel runs in the start-up batch alone)
            most = max(calls.values()) if calls else 0
            committed = {k: v for k, v in committed.items() if calls[k] * 10 >= most * 9}
        except Exception:  # noqa: BLE001
            committed = {}
    if committed:
        dom = max(committed, key=lambda k: committed[k])
    else:
        dom = max(KERNEL_OF, key=lambda k: per_kernel.get(k, 0.0))
    rocprof_ms = committed.get(dom)
    cells_per_launch = float(S * F * R) / launches_per_step
    achieved = cells_per_launch * alg_bytes_per_cell / (max(per_kernel[dom], 1e-9) / launches_per_step * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json" if R == 64 else "traffic_s128.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(KERNEL_OF[dom], {}).get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None
    # Vector-ALU issue against the MEASURED roof (tools/ubench/valu_issue.hip -> profiles/r05_valu_issue.txt: with >= 2 wavefronts per SIMD a wave64
    # VOP2 on VGPR operands issues every 2 clocks, VOP3 / compares / SGPR operands / 64-bit / packed / DPP / readlane every 4). Wave-instructions per
    # step from the committed SQ counter pass of this round (tools/pmc_sq.sh runs tools/solo_run.py on 64 streams: scaled to this run's streams x
    # firings). `frac` is against the 2-clock roof, `frac_4clk` against the 4-clock class most of the path's instructions belong to.
    valu = None
    vpath = ""
    for tag in ("r06_final", "r05_final", "r04_final"):
        cand = os.path.join(ROOT, "profiles", f"{tag}_sq_lds_l2_counters.txt")
        if os.path.exists(cand):
            vpath = cand
            break
    if R == 64 and vpath:
        try:
            import re
            per_kernel_valu = {}
            for line in open(vpath):
                m = re.match(r"(k_\w+) .*'SQ_INSTS_VALU': (\d+)", line)
                if m:
                    per_kernel_valu[m.group(1)] = float(m.group(2))
            # (kernels of a steady-state step; the start-up batch's serial kernels are not part of it; a step kernel without a line in the file
            # makes the figure meaningless: no partial sums)
            need = ["k_insert_par", KERNEL_OF["scan_ms"], "k_seg_scan", "k_assocb"]
            if all(k in per_kernel_valu for k in need):
                tot = sum(per_kernel_valu.get(k, 0.0) for k in need + ["k_assoc3", "k_publish", "k_ego", "k_begin_batch"])
                per_step = tot * (S * F) / (64.0 * 2200.0)
                peak = 1024 * 2.4e9 / 2.0  # wave-instructions per second, all 1024 SIMDs, plain VOP2
                rate = per_step / (elapsed / steps)
                valu = {"wave_instructions_per_step": per_step, "achieved": rate / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
                        "frac": rate / peak, "frac_4clk": rate / (peak / 2.0), "roof_source": "profiles/r05_valu_issue.txt",
                        "count_source": os.path.relpath(vpath, ROOT)}
        except Exception:  # noqa: BLE001
            valu = None
    res = {
        "value": cells / elapsed / 1e6,
        "ms_per_step": elapsed / steps * 1e3,
        "cells_published": cells,
        "clusters_finished": clusters,
        "serial_columns": after["serial_columns"],
        # columns the batch-parallel association kernel took itself / times it stopped at a column it hands to the serial kernel (whole run)
        "association": dict(eng.batch_counters(), kernel_launches_per_step=launches_per_step),
        "kernel_ms_per_step": per_kernel,
        "roofline": {
            "bound": "hbm", "kernel": KERNEL_OF[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": cells_per_launch * alg_bytes_per_cell, "launches_per_step": launches_per_step,
            "launch_ms": per_kernel[dom] / launches_per_step,
            "launch_ms_rocprof_committed": rocprof_ms, "rocprof_source": os.path.relpath(spath, ROOT) if rocprof_ms is not None else None,
            "traffic_source": (os.path.relpath(tpath, ROOT) + " (PMC passes of tools/pmc.sh on this round's build, not measured in this run)") if traffic is not None else None,
            "step_frac": cells * alg_bytes_per_cell / ctx.world / elapsed / 1e9 / HBM_PEAK_GBS,
            "dominant_by": "longest average launch in the committed rocprofv3 summary (rocprof_source)" if committed else "longest HIP-event duration of this run",
            "valu": valu,
        },
        "verified": verified,
        "per_rank": per_rank,
        "streams_total": int(sum(r["streams"] for r in per_rank)),
    }
    return res, eng, (xyz, inten, poses)


def live_multi_stream(torch, cfg, sensor, xyz, inten, poses, local_rank, sizes=(8, 32, 128, 550, 2200)):
    """The metric's second half for configs[2]: S live streams, n firings per stream per engine call. For every n:
    throughput with the calls pipelined (three chains in flight, like the headline) and the latency of ONE call alone — submit ->
    everything of the call associated and published, measured with a sync per call — p50 / p99 over the calls. A column of a live 10 Hz
    sensor (22 000 firings/s) is finishable when the next firing arrives; it then waits for its call to fill (up to n / 22 000 s) and
    for the call to be processed (call latency). `keeps_up` says whether S such sensors can be followed with calls of n firings."""
    from continuous_clustering_amd import Engine
    S, F, R = int(xyz.shape[1]), int(xyz.shape[2]), sensor.num_rows
    rot = min(int(xyz.shape[0]), 4)
    out = {}
    for n in sizes:
        if n > F:
            continue
        per_rot = F // n
        # contiguous [S][n][...] chunks of the first `rot` rotations
        eng = Engine(cfg, R, S, device=local_rank)
        eng.record_events(False)
        calls = []
        max_calls = 400 if n < 550 else rot * per_rot
        for b in range(rot):
            for k in range(per_rot):
                if len(calls) >= max_calls:
                    break
                calls.append((xyz[b][:, k * n:(k + 1) * n].contiguous(), inten[b][:, k * n:(k + 1) * n].contiguous(),
                              poses[b][:, k * n:(k + 1) * n].contiguous()))
        torch.cuda.synchronize()
        half = len(calls) // 2
        # first half: one call at a time (latency), second half: pipelined (throughput)
        lat = []
        warm = max(1, min(10, half // 4))  # (the first call of an engine allocates its per-call buffers: never a latency sample)
        for i, (cx, ci, cp) in enumerate(calls[:half]):
            t1 = time.perf_counter()
            eng.add_firings_device(n, cx, ci, cp)
            rc = eng.sync()
            if rc != 0:
                raise SystemExit(f"engine error {rc}: {eng.last_error()}")
            if i >= warm:
                lat.append(time.perf_counter() - t1)
        t0 = time.perf_counter()
        for cx, ci, cp in calls[half:]:
            eng.add_firings_device(n, cx, ci, cp)
        rc = eng.sync()
        el = time.perf_counter() - t0
        if rc != 0:
            raise SystemExit(f"engine error {rc}: {eng.last_error()}")
        ncalls = len(calls) - half
        lat = np.array(lat) * 1e6
        period_us = el / ncalls * 1e6
        out[str(n)] = {"firings_per_call": n, "calls_timed": ncalls, "Mpoints_per_s": S * n * R * ncalls / el / 1e6, "call_period_us": period_us,
                       "call_latency_us_p50": float(np.percentile(lat, 50)), "call_latency_us_p99": float(np.percentile(lat, 99)),
                       "fill_wait_us_max": n / 22000.0 * 1e6,
                       "column_latency_us_p99_live": float(n / 22000.0 * 1e6 + np.percentile(lat, 99)),
                       "keeps_up_with_10Hz_sensors": bool(period_us <= n / 22000.0 * 1e6)}
        eng.close()
        del calls
        torch.cuda.empty_cache()
    out["note"] = (f"{S} streams x n firings per cc_engine_add_firings_device call, inputs resident in HBM. call_latency = submit -> sync of one call alone "
                   "(all kernels of the path for the call's columns); call_period = pipelined calls back to back. column_latency_us_p99_live = "
                   "time a call of a live 22 kHz sensor takes to fill + p99 call latency: the bound for 'column finishable -> column associated' "
                   "(SURVEY 8d) when S live sensors are served with calls of n firings; keeps_up = the pipelined call period is shorter than the "
                   "time the sensors need to deliver n firings")
    return out


def replay_report(local_rank, frames_scale=0.004, verify_sequences=2):
    """BASELINE.json configs[4] shape on this GPU: the 11 SemanticKITTI train sequences as synthetic KITTI-format data on disk (frame counts of
    kitti_loader.cpp:552-562 scaled down so that the run takes seconds), replayed CONCURRENTLY as the streams of one engine through
    continuous_clustering_amd.replay: .bin -> rows / un-correction / range image / pseudo-firings on the GPU -> hot path -> device frame scatter
    -> label compare on the GPU -> records. frames/s is end to end (file reads and pose interpolation on the host included); device_frames_per_s
    leaves the host's file I/O out. A few sequences are checked against the single-sequence oracle walk (bit-equal records)."""
    import shutil
    import tempfile
    from continuous_clustering_amd import kitti, replay
    counts = {0: 4541, 1: 1101, 2: 4661, 3: 801, 4: 271, 5: 2761, 6: 1101, 7: 1101, 8: 4071, 9: 1591, 10: 1201}
    lengths = {s: max(2, int(round(n * frames_scale))) for s, n in counts.items()}
    root = tempfile.mkdtemp(prefix="cc_replay_")
    try:
        for s, nf in lengths.items():
            kitti.write_synthetic_sequence(root, s, nf, seed=500 + s, motion=(6.0 + 0.5 * s, 0.05 * s, 0.0, 0.1))
        timing = {}
        t0 = time.perf_counter()
        records, totals = replay.replay(root, list(lengths), device=local_rank, timing=timing)
        el = time.perf_counter() - t0
        out = {"sequences": len(lengths), "frames": totals["frames"], "frames_per_sequence": lengths, "seconds": el,
               "frames_per_s": totals["frames"] / el, "device_frames_per_s": totals["frames"] / max(timing.get("device_s", el), 1e-9),
               "host_io_s": timing.get("host_io_s"), "device_s": timing.get("device_s"), "records": len(records),
               "cells_published": totals["cells_published"],
               "note": "11 synthetic KITTI-format sequences (SemanticKITTI train frame counts x %.3g) replayed concurrently on one GPU; frame scatter "
                       "and label compare on the device; at N GPUs sequence i runs on rank i mod N and the records meet in one all_gather" % frames_scale}
        # oracle walk of a few sequences (loader -> clustering -> scatter -> label compare, all CPU) must give the same records, bit for bit
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        try:
            import test_gpu_kitti_replay as tk
            ok = 0
            for s in sorted(lengths, key=lambda k: lengths[k])[:verify_sequences]:
                sc, _, _ = tk.expected_records(os.path.join(root, "sequences", f"{s:02d}"), s, lengths[s])
                sc.finish()
                want = np.array(sc.records)
                got = np.array(sorted([r for r in records if int(r[0]) == s], key=lambda r: r[1]))
                if got.shape != want.shape or not np.array_equal(got.view(np.uint64), want.view(np.uint64)):
                    raise SystemExit(f"replay: records of sequence {s} differ from the oracle walk")
                ok += 1
            out["verified_sequences"] = ok
        except ImportError:
            out["verified_sequences"] = 0
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def single_stream_report(ctx, sensor, cfg, F, xyz, inten, poses):
    """BASELINE.json configs[1]: ONE 64-beam stream on one GPU. Per-column latency of the reference's calling pattern (one firing per call), the
    C++ drop-in class fed like a live sensor, and the single-stream throughput with device-resident inputs."""
    from continuous_clustering_amd import Engine
    local_rank, R = ctx.local_rank, sensor.num_rows
    out = {}
    # ---- single-stream latency (BASELINE.json configs[1] shape): one firing per call through the host API --------
    # (twice: a kernel launch per call — k_small_all —, and the resident kernel, engine option "resident": no dispatch per call, a doorbell in
    # pinned memory; the drop-in class switches it on in its synchronous mode)
    for key, resident in (("latency_us_per_column_single_stream", 0), ("latency_us_per_column_single_stream_resident", 1)):
        e1 = Engine(cfg, R, 1, device=local_rank)
        if resident:
            e1.set_option("resident", 1)
        hx = xyz[0, 0].cpu().numpy()
        hi = inten[0, 0].cpu().numpy()
        hp = poses[0, 0].cpu().numpy()
        e1.add_firings(hx[:200], hi[:200], hp[:200])
        lat = []
        for k in range(200, min(F, 1400)):
            t1 = time.perf_counter()
            e1.add_firings(hx[k:k + 1], hi[k:k + 1], hp[k:k + 1])
            lat.append(time.perf_counter() - t1)
        lat = np.array(lat) * 1e6
        out[key] = {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "p999": float(np.percentile(lat, 99.9)), "max": float(lat.max()),
                    "mode": "1 firing per cc_engine_add_firings call (firing into pinned memory + all kernels of the path + results mirrored back)" +
                            (", resident kernel" if resident else ", one launch per call")}
        if resident:
            out[key].update({k2: v2 for k2, v2 in e1.resident_counters().items() if k2 != "running"})
        e1.close()

    # ---- the same single stream through the C++ drop-in class, fed like a live HDL-64E (22 000 firings per second): a call per firing
    #      (the reference's calling pattern) falls behind, setAdaptiveBatching() hands over what queued up behind the running call ----
    demo = os.path.join(ROOT, "tests", "cpp", "dropin_demo")
    if R == 64 and os.path.exists(demo):
        import re
        import struct
        import subprocess
        import tempfile
        n_batches = int(xyz.shape[0])
        nrt = min(n_batches, 3) * F
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as tf_:
            tf_.write(struct.pack("<iiii", R, cfg.num_columns, nrt, 1))
            tf_.write(xyz[:min(n_batches, 3), 0].reshape(nrt, R, 3).cpu().numpy().astype(np.float32).tobytes())
            tf_.write(inten[:min(n_batches, 3), 0].reshape(nrt, R).cpu().numpy().astype(np.uint8).tobytes())
            tf_.write(poses[:min(n_batches, 3), 0].reshape(nrt, 12).cpu().numpy().astype(np.float64).tobytes())
            path = tf_.name
        rt = {}
        try:
            # reference_api_only: nothing but the reference's calls with its default is_single_threaded = false — the class's asynchronous mode
            # (addFiring enqueues, a worker thread runs the engine and the callbacks); latency there = due time -> ground-view callback
            # (one untimed run first: the first start of the demo binary on a fresh box pages in the libraries and loads the code objects —
            # a 20 - 30 ms stall in the middle of a 0.3 s feed that says nothing about a running system)
            subprocess.run([demo, path, "/dev/null", "-1", "0"], capture_output=True, text=True, timeout=300)
            for key, batch, rate in (("reference_api_only_paced_22kHz", -1, 22000), ("reference_api_only_free_running", -1, 0),
                                     ("adaptive_paced_22kHz", 0, 22000), ("adaptive_free_running", 0, 0), ("one_call_per_firing", 1, 0)):
                r = subprocess.run([demo, path, "/dev/null", str(batch), str(rate)], capture_output=True, text=True, timeout=300)
                m = re.search(r"firings_per_s=(\d+) latency_us_p50=([\d.]+) p99=([\d.]+) max=([\d.]+) p999=([\d.]+) stalls_over_2ms=(\d+)", r.stdout)
                if r.returncode == 0 and m:
                    rt[key] = {"firings_per_s": float(m.group(1)), "latency_us_p50": float(m.group(2)), "latency_us_p99": float(m.group(3)),
                               "latency_us_p999": float(m.group(5)), "latency_us_max": float(m.group(4)), "stalls_over_2ms": int(m.group(6))}
                    tr = [l for l in r.stderr.splitlines() if "async trace" in l]  # (CC_ASYNC_TRACE=1: the worker's longest hand-overs)
                    if tr:
                        rt[key]["trace"] = tr[-1]
        finally:
            os.unlink(path)
        rt["note"] = ("tests/cpp/dropin_demo: continuous_clustering::ContinuousClustering (C++ drop-in class over the C-ABI) with both callbacks "
                      "installed and the range_image_ mirror maintained; latency = firing due time -> return of the call that delivered it; "
                      "a sensor needs 22 000 firings/s")
        out["realtime_single_stream"] = rt

    # ---- configs[1] throughput: the ONE stream with device-resident inputs, one rotation per call (the headline harness at 1 stream) ----
    nb = int(xyz.shape[0])
    k = max(4, min(nb - 3, 40))
    sub = (xyz[:3 + k, :1].contiguous(), inten[:3 + k, :1].contiguous(), poses[:3 + k, :1].contiguous())
    r1, e1, _ = run_throughput(ctx, sensor, cfg, [0], F, k, 3, 0, inputs=sub)
    e1.close()
    out["single_stream"] = {"metric": "Mpoints/s clustered, ONE 64-beam stream on one GPU (BASELINE.json configs[1])", "value": r1["value"],
                            "unit": "Mpoints/s", "ms_per_step": r1["ms_per_step"], "steps": k, "firings_per_call": F,
                            "kernel_ms_per_step": r1["kernel_ms_per_step"],
                            "note": "inputs resident in HBM, one rotation per cc_engine_add_firings_device call, three calls in flight"}
    return out


def few_streams_report(ctx, sensor, cfg, F, xyz, inten, poses, steps, counts=(32, 64, 128), steady_steps=60):
    """What ONE GPU can say about north_star's '256 concurrent streams over the 8 GPUs of a node' (32 streams per GPU): the headline leg
    again with only the first n streams of this GPU's inputs. Same harness, same timing brackets. `value`: the driver's shape (its --steps,
    at most 40); `steady_value`: `steady_steps` timed steps on inputs of their own — at 0.4 ms per step the fill and drain of the three-deep
    pipeline are a fifth of a 20-step leg."""
    out = {}
    S = int(xyz.shape[1])
    nb = int(xyz.shape[0])
    warm = 3
    k = max(4, min(steps, nb - warm, 40))
    for n in counts:
        if n >= S:
            continue
        sub = (xyz[:warm + k, :n].contiguous(), inten[:warm + k, :n].contiguous(), poses[:warm + k, :n].contiguous())
        solo = Ctx(ctx.torch, ctx.dist, False, 1, 0, ctx.dev, ctx.local_rank, ctx.stub)
        rs = []
        for _ in range(3 if n <= 64 else 1):  # (a 20-step leg of 32 streams is 8 ms: median of three, the spread beside it)
            r, e, _ = run_throughput(solo, sensor, cfg, list(range(n)), F, k, warm, 0, inputs=sub)
            e.close()
            rs.append(r)
        r = sorted(rs, key=lambda q: q["value"])[(len(rs) - 1) // 2]
        out[str(n)] = {"streams": n, "value": r["value"], "ms_per_step": r["ms_per_step"], "steps": k,
                       "value_min": min(q["value"] for q in rs), "value_max": max(q["value"] for q in rs)}
        del sub
        if steady_steps > k and n <= 64:
            r2, e2, own = run_throughput(solo, sensor, cfg, [1234 + j for j in range(n)], F, steady_steps, warm, 0)
            e2.close()
            del own
            out[str(n)].update({"steady_value": r2["value"], "steady_ms_per_step": r2["ms_per_step"], "steady_steps": steady_steps})
            if ctx.dev.type == "cuda":
                ctx.torch.cuda.empty_cache()
    return out


def cluttered_report(ctx, sensor, cfg, F, S, steps, headline_value, args):
    """The step on streams that leave the batch-parallel association's fast path (synth.SceneModel.cluttered: vegetation over half the circle —
    more unfinished trees side by side than k_assocb has lanes for): throughput, how often and why groups went to the serial kernel, the share of
    columns the fast path took, the floor with the fast path switched off (assoc_batch = 0: every column by the serial kernels), and the CPU on
    the same streams (mode C, 16 processes)."""
    from continuous_clustering_amd import synth
    torch = ctx.torch
    warm, k = 3, max(4, min(steps, 12))
    solo = Ctx(ctx.torch, ctx.dist, False, 1, 0, ctx.dev, ctx.local_rank, ctx.stub)
    inputs = gen_inputs(torch, ctx.dev, sensor, [4321 + j for j in range(S)], F, warm + k, scene=synth.SceneModel.cluttered(0.1))
    r, e, _ = run_throughput(solo, sensor, cfg, list(range(S)), F, k, warm, 2, inputs=inputs)
    a = r["association"]
    e.close()
    r0, e0, _ = run_throughput(solo, sensor, cfg, list(range(S)), F, k, warm, 0, inputs=inputs, options={"assoc_batch": 0})
    e0.close()
    out = {"value": r["value"], "ms_per_step": r["ms_per_step"], "steps": k, "streams": S,
           "vs_headline": r["value"] / headline_value if headline_value else None,
           "batch_bails": a["batch_bails"], "bail_reasons": a["bail_reasons"],
           "fast_path_share_of_columns": a["batch_columns"] / float((warm + k) * S * F),
           "exact_replay_columns": r["serial_columns"], "verified_streams": len(r["verified"]["streams"]) if r["verified"] else 0,
           "kernel_ms_per_step": r["kernel_ms_per_step"],
           "floor_value": r0["value"], "floor_ms_per_step": r0["ms_per_step"], "floor_vs_headline": r0["value"] / headline_value if headline_value else None}
    if not args.no_cpu_baseline:
        a2 = argparse.Namespace(**vars(args))
        a2.cpu_rotations = min(args.cpu_rotations, 8)
        a2.cpu_procs = 16
        cb = cpu_baseline_report(cfg, sensor, *inputs, S, F, a2, sweep_sizes=(16,))
        out["cpu_baseline_value"] = cb["value"]
        out["cpu_baseline_cores"] = cb["cores"]
    del inputs
    if ctx.dev.type == "cuda":
        torch.cuda.empty_cache()
    return out


def host_fed_report(ctx, sensor, cfg, F, xyz, inten, poses, steps=6):
    """The same step with the firings starting in (pinned) HOST memory, as a front-end that receives sensor packets would hold them: H2D of every
    batch's [S][F][...] arrays on a copy stream, cc_engine_add_firings_device once a batch has arrived. Never `value`: the timed region of the
    headline starts with inputs resident in HBM (the contract); this key is the PCIe-inclusive rate."""
    torch = ctx.torch
    nb = min(int(xyz.shape[0]), steps + 2)
    S, R = int(xyz.shape[1]), sensor.num_rows
    host = [(xyz[b].cpu().pin_memory(), inten[b].cpu().pin_memory(), poses[b].cpu().pin_memory()) for b in range(nb)]
    devb = [(torch.empty_like(xyz[0]), torch.empty_like(inten[0]), torch.empty_like(poses[0])) for _ in range(nb)]
    bytes_per_batch = sum(int(t.numel() * t.element_size()) for t in host[0])
    eng = ctx.engine(cfg, R, S)
    eng.record_events(False)
    copy_stream = torch.cuda.Stream(device=ctx.dev)
    evs = [torch.cuda.Event() for _ in range(nb)]

    def start_copy(b):
        with torch.cuda.stream(copy_stream):
            for d, h in zip(devb[b], host[b]):
                d.copy_(h, non_blocking=True)
            evs[b].record(copy_stream)

    warm = 2
    for b in range(warm):
        start_copy(b)
        evs[b].synchronize()
        eng.add_firings_device(F, *devb[b])
    eng.sync()
    before = eng.totals()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    start_copy(warm)
    for b in range(warm, nb):
        evs[b].synchronize()
        if b + 1 < nb:
            start_copy(b + 1)
        eng.add_firings_device(F, *devb[b])
    rc = eng.sync()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if rc != 0:
        raise SystemExit(f"engine error {rc}: {eng.last_error()}")
    cells = eng.totals()["cells_published"] - before["cells_published"]
    eng.close()
    k = nb - warm
    return {"value": cells / el / 1e6, "unit": "Mpoints/s", "steps": k, "ms_per_step": el / k * 1e3, "h2d_bytes_per_step": bytes_per_batch,
            "pcie_GBs": bytes_per_batch * k / el / 1e9,
            "note": f"{S} streams x {F} firings per step copied from pinned host memory (one copy stream, the next batch's copy overlapping the "
                    "current batch's kernels) and handed to cc_engine_add_firings_device: the PCIe-inclusive rate of the headline workload"}


def main():
    args = parse()
    maybe_spawn(args)
    import torch
    import torch.distributed as dist
    from continuous_clustering_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = args.stub_engine
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
        ndev = torch.cuda.device_count()
        # (a launcher that hands every rank ONE device through HIP_VISIBLE_DEVICES leaves device 0 as the rank's GPU)
        dev_index = local_rank if ndev > local_rank else (0 if ndev == 1 and world > 1 else -1)
        if dev_index < 0:
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {ndev} are visible (--gpus {args.gpus})")
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        local_rank = dev_index
        numa = pin_to_gpu_numa_node(torch, dev_index) if (world > 1 or os.environ.get("CC_BENCH_PIN") == "1") else None
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run the RCCL path is exercised even at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ctx = Ctx(torch, dist, use_dist, world, rank, dev, local_rank, stub)
    solo = Ctx(torch, dist, False, 1, 0, dev, local_rank, stub)  # legs that only rank 0 runs

    sensor = synth.SensorModel.s64() if args.sensor == "s64" else synth.SensorModel.s128()
    cfg = capi.Config.kitti() if args.sensor == "s64" else capi.Config.vls128()
    S, F, R = args.streams, args.firings, sensor.num_rows
    n_verify = 0 if (args.no_verify or stub) else (args.verify_streams if rank == 0 else 0)

    # ================= legs every rank takes part in =================
    # ---- headline: weak scaling, S streams per GPU (BASELINE.json configs[2] per GPU) ----
    # The timed region of the driver's shape (--steps 20) is ~42 ms and the boxes of the pool differ by +- 5 %: the leg runs `--repeats` times
    # (default 3) on the same resident inputs with a new engine each time; the line carries the MEDIAN repeat (value, ms_per_step, kernel times
    # and roofline all of that one repeat) and the spread next to it (value_min / value_max / value_repeats).
    seeds = [1234 + rank * S + k for k in range(S)]
    res, eng, (xyz, inten, poses) = run_throughput(ctx, sensor, cfg, seeds, F, args.steps, args.warmup, n_verify)
    reps = [res]
    for _ in range(max(1, args.repeats) - 1):
        eng.close()
        r_more, eng, _ = run_throughput(ctx, sensor, cfg, seeds, F, args.steps, args.warmup, 0, inputs=(xyz, inten, poses))
        r_more["verified"] = res["verified"]  # (the first repeat checked the streams against the oracle: same inputs, same results)
        reps.append(r_more)
    res = sorted(reps, key=lambda r: r["value"])[(len(reps) - 1) // 2]
    out = None
    if rank == 0:
        out = {
            "metric": "Mpoints/s clustered (64-beam streams)" if R == 64 else f"Mpoints/s clustered ({R}-beam streams)",
            "value": res["value"],
            "value_min": min(r["value"] for r in reps),
            "value_max": max(r["value"] for r in reps),
            "value_repeats": [r["value"] for r in reps],
            "value_is": f"median of {len(reps)} repeats of the timed leg ({args.steps} steps each, same inputs, a new engine per repeat)",
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "stub (launcher test, not a measurement)" if stub else "synthetic",
            "config": {
                "workload": f"{S} concurrent synthetic S{R} streams per GPU ({R} rows x {cfg.num_columns} columns/rotation), {F} firings per stream per step, "
                            f"inputs resident in HBM; BASELINE.json configs[2]",
                "streams_per_gpu": S, "firings_per_step": F, "num_rows": R, "num_columns": cfg.num_columns,
                "sharding": f"stream-per-wavefront, {world} rank(s) x {S} streams, no data-path collective",
            },
            "cells_published": res["cells_published"],
            "clusters_finished": res["clusters_finished"],
            "serial_columns": res["serial_columns"],
            "association": res["association"],
            "kernel_ms_per_step": res["kernel_ms_per_step"],
            "roofline": res["roofline"],
            "verified_streams": len(res["verified"]["streams"]) if res["verified"] else 0,
            "verify": res["verified"],
            "rccl_world": world if use_dist else 0,
            "numa_pin": None if stub else numa,
            "per_rank_value": [r["value"] for r in res["per_rank"]],
            "per_rank": res["per_rank"],
        }
    eng.close()

    # ---- strong split (north_star: "256 concurrent streams" sharded across the GPUs of one node): --total-streams streams in all,
    #      stream s on rank s mod N; the same step, fewer streams per GPU as N grows ----
    T = args.total_streams
    if not args.no_strong_split and T > 0:
        mine = [s for s in range(T) if s % world == rank]
        if world == 1 and T == S:
            if rank == 0:
                out["strong_split"] = {"total_streams": T, "streams_per_gpu": [T], "value": res["value"], "ms_per_step": res["ms_per_step"],
                                       "scaling": "strong", "note": "one GPU holds all the streams: this IS the headline leg (not run twice)"}
        else:
            # rank r's engine stream k is global stream mine[k]; its inputs are the scene of seed 1234 + s like the headline's stream s of rank 0
            sub = None
            if world == 1 and T <= S:
                nbk = args.warmup + args.steps
                sub = (xyz[:nbk, :T].contiguous(), inten[:nbk, :T].contiguous(), poses[:nbk, :T].contiguous())
            else:
                del xyz, inten, poses
                if dev.type == "cuda":
                    torch.cuda.empty_cache()
                xyz = inten = poses = None
            r3, e3, b3 = run_throughput(ctx, sensor, cfg, [1234 + s for s in mine], F, args.steps, args.warmup, 0, inputs=sub)
            e3.close()
            if rank == 0:
                out["strong_split"] = {"total_streams": r3["streams_total"], "streams_per_gpu": [r["streams"] for r in r3["per_rank"]],
                                       "value": r3["value"], "ms_per_step": r3["ms_per_step"], "scaling": "strong",
                                       "per_rank_value": [r["value"] for r in r3["per_rank"]],
                                       "kernel_ms_per_step": r3["kernel_ms_per_step"],
                                       "note": "stream s lives on rank s mod N; no data-path collective; value = cells of all ranks / slowest rank"}
            if xyz is None:
                xyz, inten, poses = b3  # (rank 0's solo legs below use whatever inputs are resident)
            del sub

    # ---- BASELINE.json configs[3]: 128-row VLS-128-shaped streams (same harness, fewer steps; the headline stays S64) ----
    s128_inputs = None
    if args.sensor == "s64" and not args.no_s128:
        sensor2, cfg2 = synth.SensorModel.s128(), capi.Config.vls128()
        steps2, warm2 = max(4, min(args.steps, 10)), 3
        r2, eng2, s128_inputs = run_throughput(ctx, sensor2, cfg2, [5678 + rank * S + k for k in range(S)], 1700, steps2, warm2, min(n_verify, 2))
        eng2.close()
        if rank == 0:
            out["s128"] = {"metric": "Mpoints/s clustered (128-beam streams)", "value": r2["value"], "unit": "Mpoints/s",
                           "ms_per_step": r2["ms_per_step"], "steps": steps2, "warmup": warm2,
                           "workload": f"{S} concurrent synthetic S128 streams per GPU (128 rows x 1700 columns/rotation, per-laser azimuth "
                                       f"offsets: every firing spans ~60 columns; library-default parameters), 1700 firings per stream per step",
                           "kernel_ms_per_step": r2["kernel_ms_per_step"], "roofline": r2["roofline"],
                           "verified_streams": len(r2["verified"]["streams"]) if r2["verified"] else 0,
                           "serial_columns": r2["serial_columns"], "per_rank_value": [r["value"] for r in r2["per_rank"]]}
        if rank != 0 or args.no_cpu_baseline:
            s128_inputs = None

    # ---- real-data acceptance (BASELINE.json configs[0] / [4]): only where SemanticKITTI is mounted; sequence i on rank i mod N,
    #      per-frame records gathered with one all_gather (RCCL) ----
    sk = None
    if args.kitti_root and not stub:
        from continuous_clustering_amd import acceptance
        seqs = [int(v) for v in args.kitti_sequences.split(",") if v.strip()]
        sk = acceptance.run(args.kitti_root, seqs, rank=rank, world=world, device=local_rank, max_frames=args.kitti_max_frames or None)

    # ================= the other ranks are done: everything below is rank 0 alone (the job's timed legs are over) =================
    ctx.barrier()
    if use_dist:
        dist.destroy_process_group()
    if rank != 0:
        return
    Sx = int(xyz.shape[1])  # streams of the inputs that are resident now (S, or this rank's share of the strong split)

    # The legs below are reports beside the headline (measured above). One of them failing must not take the line away from whoever reads it:
    # the failure is printed, named in the line (`leg_errors`), that leg's key is left out and the process exits with 1 AFTER the line.
    leg_errors = {}

    def leg(name, fn):
        try:
            return fn()
        except (Exception, SystemExit) as ex:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            leg_errors[name] = f"{type(ex).__name__}: {ex}"[:300]
            return None

    # ---- what one GPU says about fewer streams per GPU (the strong-split regime) ----
    if not args.no_few_streams and not stub and world == 1:
        out["few_streams"] = leg("few_streams", lambda: few_streams_report(solo, sensor, cfg, F, xyz, inten, poses, args.steps))

    # ---- single-stream latency (BASELINE.json configs[1] shape): one firing per call through the host API --------
    if not args.no_latency and not stub:
        out.update(leg("single_stream", lambda: single_stream_report(solo, sensor, cfg, F, xyz, inten, poses)) or {})

    # ---- configs[2], second half of the metric: live streams served with small calls (throughput and latency vs firings per call) ----
    if not args.no_latency and not stub:
        out["live_multi_stream"] = leg("live_multi_stream", lambda: live_multi_stream(torch, cfg, sensor, xyz, inten, poses, local_rank))

    # ---- the headline workload fed from pinned host memory (PCIe-inclusive; never `value`) ----
    if not args.no_host_fed and not stub:
        out["host_fed"] = leg("host_fed", lambda: host_fed_report(solo, sensor, cfg, F, xyz, inten, poses))

    # ---- streams that leave the association's fast path (vegetation), and the floor without it ----
    if not args.no_cluttered and not args.no_few_streams and not stub and world == 1:  # (--no-few-streams: the tools' "headline leg only")
        out["cluttered"] = leg("cluttered", lambda: cluttered_report(solo, sensor, cfg, F, min(Sx, 256), args.steps, out["value"], args))

    # ---- CPU baseline on this box's host cores (inputs: the bench's own streams, copied back from HBM) ------------
    # (the contract: on rank 0 at N = 1 only — a multi-GPU line carries the per-GPU figure of the N = 1 run)
    if not args.no_cpu_baseline and not stub and world == 1:
        out["cpu_baseline"] = leg("cpu_baseline", lambda: cpu_baseline_report(cfg, sensor, xyz, inten, poses, Sx, F, args))
        if s128_inputs is not None and "s128" in out:
            a2 = argparse.Namespace(**vars(args))
            a2.cpu_rotations = min(args.cpu_rotations, 10)
            out["s128"]["cpu_baseline"] = leg("s128_cpu_baseline", lambda: cpu_baseline_report(capi.Config.vls128(), synth.SensorModel.s128(), *s128_inputs, Sx, 1700, a2,
                                                                                             sweep_sizes=(1, 64)))
    else:
        out["cpu_baseline"] = None
    del xyz, inten, poses, s128_inputs
    if dev.type == "cuda":
        torch.cuda.empty_cache()

    # ---- BASELINE.json configs[4] shape: concurrent replay of KITTI-format sequences, end to end ----
    if not args.no_latency and not stub:
        out["replay"] = leg("replay", lambda: replay_report(local_rank))

    # ---- real-data acceptance (BASELINE.json configs[0] / [4]): only where SemanticKITTI is mounted ----
    if leg_errors:
        out["leg_errors"] = leg_errors
    out["semantic_kitti"] = sk if sk is not None else {"skipped": "no --kitti-root / $SEMANTIC_KITTI_ROOT: the dataset is not in this image"}

    # The driver keeps a few KB of this line: the full record (every leg's detail and the notes on how it was taken) goes to a file, the printed
    # line carries the contract's keys and the numbers (DESIGN.md section 6 says what each leg is).
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as fh:
            json.dump(out, fh)
    except OSError:
        pass
    print(json.dumps(slim_line(out)), flush=True)
    if leg_errors:
        sys.exit(1)  # (the line is out — the headline was measured — but a failed report leg is still a failed run)


def _r(v, nd=5):
    """floats to `nd` significant digits (the line is for reading numbers, not for reproducing them bit by bit)"""
    if isinstance(v, float):
        return float(f"{v:.{nd}g}")
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, list):
        return [_r(x, nd) for x in v]
    return v


def slim_line(o):
    def g(d, *ks):  # nested get that gives None where a leg did not run
        for k in ks:
            if not isinstance(d, dict):
                return None
            d = d.get(k)
        return d

    line = {k: o.get(k) for k in ("metric", "value", "value_min", "value_max", "value_repeats", "value_is", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                  "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfgd = dict(o.get("config") or {})
    line["config"] = cfgd
    rf = dict(o.get("roofline") or {})
    for k in ("rocprof_source", "traffic_source", "dominant_by", "note"):
        rf.pop(k, None)
    if isinstance(rf.get("valu"), dict):
        rf["valu"] = {k: v for k, v in rf["valu"].items() if k in ("achieved", "peak", "unit", "frac", "frac_4clk", "roof_source")}
    line["roofline"] = rf
    cb = o.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": f"mode C: {cb.get('cores')} pinned single-threaded oracle processes x {cb.get('rotations_per_instance')} rotations of the bench's streams",
                                "cpu_model": cb.get("cpu_model"), "physical_cores": cb.get("physical_cores"), "single_core_value": cb.get("single_core_value"),
                                "sweep": {k: v.get("mpoints_per_s") for k, v in (cb.get("sweep") or {}).items()},
                                "mode_b_value": g(cb, "mode_b", "value"),
                                "mode_a_latency_us_per_column": {k: v for k, v in (cb.get("mode_a_latency_us_per_column") or {}).items() if k != "note"}}
    else:
        line["cpu_baseline"] = None
    # flat figures the round's targets are quoted on
    fs = o.get("few_streams") or {}
    line["few_streams"] = {k: {kk: vv for kk, vv in v.items() if kk != "streams"} for k, v in fs.items() if isinstance(v, dict)}
    line["few_streams_32_value"] = g(fs, "32", "value")
    line["few_streams_32_steady_value"] = g(fs, "32", "steady_value")
    s128 = o.get("s128") or {}
    line["s128_value"] = s128.get("value")
    line["s128"] = {"value": s128.get("value"), "ms_per_step": s128.get("ms_per_step"), "steps": s128.get("steps"), "verified_streams": s128.get("verified_streams"),
                    "roofline": {k: g(s128, "roofline", k) for k in ("kernel", "achieved", "frac", "traffic", "step_frac", "launch_ms")},
                    "cpu_baseline_value": g(s128, "cpu_baseline", "value"), "cpu_baseline_cores": g(s128, "cpu_baseline", "cores")}
    line["latency_us_per_column_single_stream"] = {k: v for k, v in (o.get("latency_us_per_column_single_stream") or {}).items() if k != "mode"}
    line["latency_us_per_column_single_stream_resident"] = {k: v for k, v in (o.get("latency_us_per_column_single_stream_resident") or {}).items() if k != "mode"}
    for k in ("verified_streams", "rccl_world", "cells_published", "clusters_finished", "serial_columns", "association", "kernel_ms_per_step", "per_rank_value", "per_rank", "numa_pin"):
        line[k] = o.get(k)
    ss = o.get("strong_split")
    if ss:
        line["strong_split"] = {k: v for k, v in ss.items() if k in ("total_streams", "streams_per_gpu", "value", "ms_per_step", "per_rank_value", "scaling")}
    st = o.get("single_stream")
    if st:
        line["single_stream"] = {k: v for k, v in st.items() if not isinstance(v, str)}
    rt = o.get("realtime_single_stream")
    if rt:
        line["realtime_single_stream"] = {k: ({kk: vv for kk, vv in v.items() if not isinstance(vv, str)} if isinstance(v, dict) else v)
                                          for k, v in rt.items() if k != "note"}
    lm = o.get("live_multi_stream")
    if lm:
        line["live_multi_stream"] = {k: {kk: vv for kk, vv in v.items() if kk in ("value", "call_period_us", "call_latency_us_p50", "call_latency_us_p99",
                                                                                   "column_latency_us_p99_live", "keeps_up")}
                                     for k, v in lm.items() if isinstance(v, dict)}
    cl = o.get("cluttered")
    if cl:
        line["cluttered"] = {k: v for k, v in cl.items() if k != "kernel_ms_per_step"}
    line["cluttered_value"] = g(cl, "value")  # (vegetation-like scenes: the nearest thing to north_star's "SemanticKITTI streams" this image allows)
    hf = o.get("host_fed")
    if hf:
        line["host_fed"] = {k: v for k, v in hf.items() if k in ("value", "ms_per_step", "pcie_GBs", "steps")}
    rp = o.get("replay")
    if rp:
        line["replay"] = {k: v for k, v in rp.items() if not isinstance(v, (str, dict, list))}
    sk = o.get("semantic_kitti")
    line["semantic_kitti"] = "skipped" if isinstance(sk, dict) and "skipped" in sk else sk
    if o.get("leg_errors"):
        line["leg_errors"] = o["leg_errors"]  # legs beside the headline that raised (their keys are missing above)
    line["detail"] = "gpurun_out/bench_detail.json"
    # (the contract's own scalars keep their digits: the driver cross-checks value against cells and time)
    return {k: (v if not isinstance(v, (dict, list)) else _r(v)) for k, v in line.items()}


if __name__ == "__main__":
    main()
