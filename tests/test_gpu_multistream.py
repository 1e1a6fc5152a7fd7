"""GPU: the throughput entry (device-resident inputs, many streams per launch) at and above BASELINE.json sizes,
checked through size-independent properties and against the oracle on sampled streams."""
import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import capi, synth

pytestmark = pytest.mark.gpu


def _device_batches(torch, streams, n_batches, F):
    S = len(streams)
    R = streams[0].sensor.num_rows
    xyz = torch.empty((n_batches, S, F, R, 3), dtype=torch.float32, device="cuda")
    inten = torch.empty((n_batches, S, F, R), dtype=torch.uint8, device="cuda")
    poses = torch.empty((n_batches, S, F, 12), dtype=torch.float64, device="cuda")
    for s, st in enumerate(streams):
        xyz[:, s] = torch.from_numpy(st.xyz[:n_batches * F]).view(n_batches, F, R, 3)
        inten[:, s] = torch.from_numpy(st.intensity[:n_batches * F]).view(n_batches, F, R)
        poses[:, s] = torch.from_numpy(st.poses[:n_batches * F]).view(n_batches, F, 12)
    return xyz, inten, poses


def test_multi_stream_device_path_equals_per_stream_oracle(oracle_lib):
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    sen = synth.SensorModel(num_rows=64, num_columns=720)
    cfg = capi.Config.kitti()
    cfg.num_columns = 720
    S, F, NB = 6, 720, 3
    motions = [synth.Motion.static(), synth.Motion.translate(), synth.Motion.turn()]
    streams = [synth.make_stream(F * NB, seed=100 + s, sensor=sen, motion=motions[s % 3]) for s in range(S)]
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    e = Engine(cfg, 64, S)
    e.record_events(True)
    oracles = [Oracle(cfg, 64) for _ in range(S)]
    for s in range(S):
        assert oracles[s].add_firings(streams[s].xyz, streams[s].intensity, streams[s].poses) == 0
    evo = [o.drain_events() for o in oracles]
    pos = [0] * S
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
        assert e.sync() == 0, e.last_error()
        for s in range(S):
            ev = e.drain_events(s)
            ref = evo[s][pos[s]:pos[s] + len(ev)]
            assert len(ev) == len(ref)
            assert (ev["stream"] == s).all()
            for fld in ("type", "a", "b", "c", "d", "column"):
                assert np.array_equal(ev[fld], ref[fld]), (s, fld)
            pos[s] += len(ev)
            pub = ev[(ev["type"] == capi.EV_PUBLISH_COLUMNS) & (ev["b"] >= ev["a"])]
            if len(pub):
                lo, hi = int(pub["a"].min()), int(pub["b"].max())
                util.compare_columns(oracles[s].read_published(lo, hi), e.read_columns(lo, hi, stream=s), lo)
    for s in range(S):
        assert pos[s] == len(evo[s])
        so, se = oracles[s].state(), e.state(s)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (s, k)


def test_kernel_times_sampled_every_nth_batch_stand_for_all_batches():
    """option "timing_every": the HIP events of every 4th batch, scaled by the engine to all batches, tell the same story as events around
    every batch (same number of batches, per-kernel sums within a factor of two on a quiet 8-stream engine) and change no result."""
    import torch
    from continuous_clustering_amd import Engine
    sen = synth.SensorModel(num_rows=64, num_columns=720)
    cfg = capi.Config.kitti()
    cfg.num_columns = 720
    S, F, NB = 8, 360, 13
    streams = [synth.make_stream(F * NB, seed=300 + s, sensor=sen, motion=synth.Motion.translate()) for s in range(S)]
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    got = {}
    for every in (1, 4):
        e = Engine(cfg, 64, S)
        e.record_events(False)
        e.set_option("timing_every", every)
        e.add_firings_device(F, xyz[0], inten[0], poses[0])
        assert e.sync() == 0, e.last_error()
        e.enable_timing(True)
        for b in range(1, NB):
            e.add_firings_device(F, xyz[b], inten[b], poses[b])
        assert e.sync() == 0, e.last_error()
        kt = e.kernel_times()
        e.enable_timing(False)
        got[every] = (kt, e.totals(), [e.state(s) for s in range(S)])
        e.close()
    k1, k4 = got[1][0], got[4][0]
    assert k1["batches"] == k4["batches"] == NB - 1
    for name in ("prep_ms", "segment_ms", "scan_ms", "assoc_lds_ms", "publish_ms"):
        assert k1[name] > 0 and k4[name] > 0, name
        assert 0.5 < k4[name] / k1[name] < 2.0, (name, k1[name], k4[name])
    assert got[1][1] == got[4][1]
    for a, b in zip(got[1][2], got[4][2]):
        for k in util.STATE_FIELDS:
            assert a[k] == b[k], k


def test_full_size_properties_256_streams():
    """BASELINE.json configs[2] shape: 256 concurrent 64 x 2200 streams. Properties: every stream publishes, totals add
    up, replicated inputs give identical per-stream results (determinism across wavefronts), output planes carry only
    legal labels, and a sampled stream equals the oracle."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    cfg = capi.Config.kitti()
    S, F, NB = 256, 2200, 2
    distinct = 4
    base = [synth.make_stream(F * NB, seed=500 + s, motion=synth.Motion.translate()) for s in range(distinct)]
    streams = [base[s % distinct] for s in range(S)]
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    e = Engine(cfg, 64, S)
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0, e.last_error()
    tot = e.totals()
    states = [e.state(s) for s in range(S)]
    assert tot["firings_consumed"] == S * F * NB
    assert tot["cells_published"] == sum(st["cells_published"] for st in states)
    assert all(st["cells_published"] > 0.9 * 64 * (F * NB - 400) for st in states)
    for s in range(distinct, S):
        for k in util.STATE_FIELDS:
            assert states[s][k] == states[s % distinct][k], (s, k)
    # sampled stream vs oracle
    for s in (0, 3):
        o = Oracle(cfg, 64)
        assert o.add_firings(streams[s].xyz, streams[s].intensity, streams[s].poses) == 0
        so = o.state()
        for k in util.STATE_FIELDS:
            assert so[k] == states[s][k], (s, k)
        # columns still in the ring (not yet cleared) must equal the oracle's published snapshot
        hi = states[s]["first_unpublished_global_column_index"] - 1
        lo = hi - 1500
        for s2 in (s, s + distinct * 7):
            util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s2), lo, mirror=False)
    g_ptr, id_ptr = e.output_planes(0)
    assert g_ptr and id_ptr


def test_full_size_properties_256_streams_s128():
    """BASELINE.json configs[3] shape: 256 concurrent VLS-128-shaped streams (128 rows x 1700 columns, per-laser azimuth offsets: every
    firing spans ~60 columns; library-default parameters), two rotations through the pipelined device path. Same properties as the S64 test:
    totals add up, replicated inputs give identical per-stream results, sampled streams equal the oracle (state + the last published columns)."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    cfg = capi.Config.vls128()
    sensor = synth.SensorModel.s128()
    S, F, NB = 256, 1700, 2
    distinct = 3
    base = [synth.make_stream(F * NB, seed=700 + s, sensor=sensor, motion=synth.Motion.translate(), start_column=40) for s in range(distinct)]
    streams = [base[s % distinct] for s in range(S)]
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    e = Engine(cfg, 128, S)
    e.record_events(False)
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0, e.last_error()
    tot = e.totals()
    states = [e.state(s) for s in range(S)]
    assert tot["firings_consumed"] == S * F * NB
    assert tot["cells_published"] == sum(st["cells_published"] for st in states)
    assert all(st["cells_published"] > 0.8 * 128 * (F * NB - 600) for st in states)
    for s in range(distinct, S):
        for k in util.STATE_FIELDS:
            assert states[s][k] == states[s % distinct][k], (s, k)
    for s in (0, 2):
        o = Oracle(cfg, 128)
        assert o.add_firings(streams[s].xyz, streams[s].intensity, streams[s].poses) == 0
        so = o.state()
        for k in util.STATE_FIELDS:
            assert so[k] == states[s][k], (s, k)
        hi = states[s]["first_unpublished_global_column_index"] - 1
        lo = hi - 1200
        for s2 in (s, s + distinct * 11):
            util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s2), lo, mirror=False)


@pytest.mark.parametrize("pipeline,sub_batch", [(1, 0), (2, 0), (1, 300), (0, 0)])
def test_pipelined_throughput_path_matches_oracle(pipeline, sub_batch, oracle_lib):
    """The bench configuration of the engine (events off: consecutive batches overlap on three or four chains of HIP streams,
    the per-point preparation of the next batch runs ahead, a call may be cut into sub-batches) must leave every stream in the
    oracle's state and with the oracle's published columns."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    sen = synth.SensorModel(num_rows=64, num_columns=720)
    cfg = capi.Config.kitti()
    cfg.num_columns = 720
    S, F, NB = 8, 720, 5
    motions = [synth.Motion.static(), synth.Motion.translate(), synth.Motion.turn()]
    streams = [synth.make_stream(F * NB, seed=300 + s, sensor=sen, motion=motions[s % 3]) for s in range(S)]
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    e = Engine(cfg, 64, S)
    e.record_events(False)
    e.set_option("pipeline", pipeline)
    e.set_option("sub_batch", sub_batch)
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0, e.last_error()
    for s in range(S):
        o = Oracle(cfg, 64)
        assert o.add_firings(streams[s].xyz, streams[s].intensity, streams[s].poses) == 0
        so, se = o.state(), e.state(s)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (s, k)
        hi = se["first_unpublished_global_column_index"] - 1
        lo = max(hi - 600, se["ring_buffer_start_global_column_index"])
        util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s), lo, mirror=False)


@pytest.mark.parametrize("firings,rounds,store_fin", [(240, 0, -1), (97, 0, -1), (240, 1, -1), (720, 2, -1), (240, 0, 0), (240, 0, 1), (720, 2, 0)])
def test_pipelined_path_with_streams_the_batch_parallel_association_stops_at(firings, rounds, store_fin, oracle_lib):
    """Pipelined device path (events off: four chains of HIP streams, batches overlap) over streams made to make k_assocb stop (tests/cases.py
    EXCEPTION_CASES) next to ordinary ones: the hand-over to the serial kernel (adaptive number of rounds, or pinned) happens while the next batch is
    inserted, segmented and scanned. Every stream must end in the oracle's state with the oracle's published columns, and the exception path must
    have been taken."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    names = ["x_s64_slanted_gaps", "s64_translate", "x_s64_slanted_gaps_far", "s64_turn", "x_s64_near_jitter_gaps_3", "s64_static", "x_s64_refused_attach"]
    built = [cases.build_case(n) for n in names]
    cfg = built[0][1]
    keep = [i for i, b in enumerate(built) if bytes(b[1]) == bytes(cfg) and b[2] is None]
    assert len(keep) >= 4, [names[i] for i in keep]
    streams = [built[i][0] for i in keep]
    S = len(streams)
    F = firings
    NB = min(st.n_firings for st in streams) // F
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    e = Engine(cfg, 64, S)
    e.record_events(False)
    e.set_option("assoc_rounds", rounds)
    e.set_option("scan_store_fin", store_fin)  # (the serial kernels read the points' finished_at contributions / recompute them / per launch)
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0, e.last_error()
    bc = e.batch_counters()
    assert bc["batch_bails"] > 0, bc
    for s in range(S):
        o = Oracle(cfg, 64)
        assert o.add_firings(streams[s].xyz[:NB * F], streams[s].intensity[:NB * F], streams[s].poses[:NB * F]) == 0
        so, se = o.state(), e.state(s)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (names[keep[s]], k, so[k], se[k])
        hi = se["first_unpublished_global_column_index"] - 1
        lo = max(hi - 600, se["ring_buffer_start_global_column_index"])
        util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s), lo, mirror=False)


@pytest.mark.parametrize("S,split", [(48, 0), (48, 1), (24, 3), (24, 0), (30, 0), (38, 0), (72, 0), (100, 0)])
def test_insertion_dealt_to_several_blocks_per_stream(S, split, oracle_lib):
    """Launches of at most 96 streams run k_insert_par with 16 wavefronts per block and deal a stream's firings to several blocks (8 up to 24 streams,
    6 up to 32, 4 up to 40, 3 up to 64, else 2; option insert_split_blocks pins the number), k_insert_par_fin closes the stream's state; above 96 streams one block of 8 wavefronts takes a
    stream. Streams with firings out of shape (duplicated / empty / backwards: the run ends there and later blocks' cells are taken back) next to regular
    ones, pipelined device path: every stream ends in the oracle's state with the oracle's published columns."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    from test_gpu_parallel_insert import perturbed_stream
    cfg = capi.Config.kitti()
    distinct = 6
    base = [perturbed_stream(4000 + s) if s % 2 else synth.make_stream(2200 * 2 + 200, seed=900 + s, motion=synth.Motion.translate()) for s in range(distinct)]
    F = 1100
    NB = min(st.n_firings for st in base) // F
    streams = [base[s % distinct] for s in range(S)]
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    e = Engine(cfg, 64, S)
    e.record_events(False)
    if split:
        e.set_option("insert_split_blocks", split)
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0, e.last_error()
    states = [e.state(s) for s in range(S)]
    for s in range(distinct, S):
        for k in util.STATE_FIELDS:
            assert states[s][k] == states[s % distinct][k], (s, k)
    for s in range(distinct):
        o = Oracle(cfg, 64)
        assert o.add_firings(base[s].xyz[:NB * F], base[s].intensity[:NB * F], base[s].poses[:NB * F]) == 0
        so = o.state()
        for k in util.STATE_FIELDS:
            assert so[k] == states[s][k], (s, k, so[k], states[s][k])
        hi = states[s]["first_unpublished_global_column_index"] - 1
        lo = max(hi - 1000, states[s]["ring_buffer_start_global_column_index"])
        for s2 in (s, s + distinct * ((S - 1 - s) // distinct)):
            util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s2), lo, mirror=False)


@pytest.mark.parametrize("F,limit", [(700, 0), (1100, 0), (700, 300)])
def test_insertion_enqueued_ahead_of_the_previous_gate(F, limit, oracle_lib):
    """Few streams (option lazy_gate, default <= 40): a call enqueues its insertion BEFORE the host has read the previous batch's insertion counters; the
    kernels check those counters themselves and return at once when the previous batch still needs the serial insertion kernels, and the host launches
    them again behind those (cc_engine_gate_counters). Regular batches between batches with firings out of shape (so that the engine keeps enqueueing
    ahead), one of them the last of the run, one with the emission limit reached inside the serial kernel (continuation passes between the two launches):
    oracle's state and published columns for every stream."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    cfg = capi.Config.kitti()
    S, NB = 5, 9
    streams = []
    for s in range(S):
        st = synth.make_stream(F * NB, seed=7100 + s, motion=synth.Motion.translate())
        xyz, inten, poses = st.xyz.copy(), st.intensity.copy(), st.poses.copy()
        if s in (1, 3):
            for b in ((2, 5, 8) if s == 1 else (5,)):
                k = b * F + 100 + 37 * s
                xyz[k + 1], inten[k + 1], poses[k + 1] = xyz[k], inten[k], poses[k]  # the same firing twice: the second lands in occupied cells
                xyz[k + 200, ::2] = xyz[k + 201, ::2]  # returns that straddle two columns
                xyz[k + 300] = np.nan  # an empty firing
        streams.append(synth.Stream(xyz=xyz, intensity=inten, poses=poses, sensor=st.sensor))
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    e = Engine(cfg, 64, S)
    e.record_events(False)
    if limit:
        e.set_option("limit_columns", limit)
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0, e.last_error()
    gc = e.gate_counters()
    # (with the limit every batch needs the serial kernel: after two such batches in a row the engine stops enqueueing ahead)
    assert gc["lazy_batches"] >= (2 if limit else NB - 3) and gc["lazy_redone"] >= 2, gc
    for s in range(S):
        o = Oracle(cfg, 64)
        assert o.add_firings(streams[s].xyz, streams[s].intensity, streams[s].poses) == 0
        so, se = o.state(), e.state(s)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (s, k, so[k], se[k])
        hi = se["first_unpublished_global_column_index"] - 1
        lo = max(hi - 1500, se["ring_buffer_start_global_column_index"])
        util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s), lo, mirror=False)
    e.close()


@pytest.mark.parametrize("rows", [16, 48, 100])
def test_pipelined_path_at_other_row_counts(rows, oracle_lib):
    """The throughput path (events off, chains overlapped, block-parallel insertion, batch-parallel association) at row counts between the usual ones."""
    import torch
    from continuous_clustering_amd import Engine
    from oracle.pyoracle import Oracle
    sen = synth.SensorModel(num_rows=rows, num_columns=720, incl_top_deg=6.0, incl_bottom_deg=-26.0)
    cfg = capi.Config.kitti()
    cfg.num_columns = 720
    S, F, NB = 5, 720, 4
    motions = [synth.Motion.static(), synth.Motion.translate(), synth.Motion.turn()]
    streams = [synth.make_stream(F * NB, seed=3100 + rows + s, sensor=sen, motion=motions[s % 3]) for s in range(S)]
    xyz, inten, poses = _device_batches(torch, streams, NB, F)
    e = Engine(cfg, rows, S)
    e.record_events(False)
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0, e.last_error()
    for s in range(S):
        o = Oracle(cfg, rows)
        assert o.add_firings(streams[s].xyz, streams[s].intensity, streams[s].poses) == 0
        so, se = o.state(), e.state(s)
        for k in util.STATE_FIELDS:
            assert so[k] == se[k], (s, k)
        hi = se["first_unpublished_global_column_index"] - 1
        lo = max(hi - 600, se["ring_buffer_start_global_column_index"])
        util.compare_columns(o.read_published(lo, hi), e.read_columns(lo, hi, stream=s), lo, mirror=False)


@pytest.mark.gpu
def test_bench_under_the_distributed_launcher_world_1():
    """The driver's multi-GPU command at the one world size a box offers: bench.py under torch.distributed.run (RCCL process group, barrier,
    max-over-ranks, all-gather of the counts) with 32 streams dealt over the job (north_star's strong split) and the oracle check of the headline."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(root, "bench.py"), "--gpus", "1", "--streams", "64", "--steps", "4", "--warmup", "3", "--total-streams", "32", "--no-s128",
           "--no-cpu-baseline", "--no-latency", "--no-few-streams", "--no-host-fed", "--verify-streams", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CC_BENCH_PIN="1")
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["rccl_world"] == 1 and out["n_gpus"] == 1
    assert out["strong_split"]["streams_per_gpu"] == [32] and out["strong_split"]["total_streams"] == 32
    assert out["verified_streams"] == 2
    assert out["value"] > 0 and out["strong_split"]["value"] > 0
    assert len(json.dumps(out)) < 6144  # (the driver keeps a few KB of the line)
