"""GPU: label compare (cc_eval_frame) against the oracle, bit-exact, and the evaluation data flow end to end:
engine -> published cells -> frame scatter (kitti_demo.cpp:173-224) -> GPU label compare == oracle path."""
import numpy as np
import pytest

import util
from test_eval_cpu import random_frame

pytestmark = pytest.mark.gpu


def test_label_compare_matches_oracle_bit_exact(oracle_lib):
    from continuous_clustering_amd import evaluation
    from oracle import pyoracle
    rng = np.random.default_rng(5)
    for n, n_gt, n_det in ((1, 2, 2), (63, 5, 5), (4097, 30, 40), (123456, 400, 900), (300000, 20000, 30000)):
        f = random_frame(rng, n, n_gt, n_det)
        a, b = evaluation.eval_frame(*f), pyoracle.eval_frame(*f)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (n, a, b)
    # the pair (0xFFFFFFFF, 0xFFFFFFFF) has the bit pattern of a free slot of the device hash table: it is counted beside the table
    f = [x.copy() for x in random_frame(rng, 5000, 10, 10)]
    f[1] = f[1].astype(np.uint32)
    f[3] = f[3].astype(np.uint32)
    f[1][100:400] = 0xFFFFFFFF
    f[3][100:300] = 0xFFFFFFFF
    a, b = evaluation.eval_frame(*f), pyoracle.eval_frame(*f)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (a, b)
    z = np.zeros(0)
    assert evaluation.eval_frame(z, z, z, z).tolist() == [0.0] * 6
    # all unlabeled / all background
    n = 1000
    a = evaluation.eval_frame(np.zeros(n), np.zeros(n), np.ones(n), np.zeros(n))
    assert a.tolist() == [0.0] * 6
    # device-resident variant
    import torch
    f = random_frame(rng, 50000, 50, 80)
    t = [torch.from_numpy(x.astype(dt)).cuda() for x, dt in zip(f, (np.int16, np.int32, np.uint8, np.int32))]
    a = evaluation.eval_frame_device(50000, *t)
    assert np.array_equal(a.view(np.uint64), pyoracle.eval_frame(*f).view(np.uint64))


def test_stream_to_evaluation_end_to_end(oracle_lib):
    """One rotation = one frame. Ground truth from the synthetic scene (hit ids), detection from the engine."""
    from continuous_clustering_amd import Engine, capi, evaluation, synth
    from oracle import pyoracle
    cols, rows, nframes = 720, 64, 3
    sensor = synth.SensorModel(num_rows=rows, num_columns=cols)
    stream = synth.make_stream(cols * nframes + 40, seed=77, sensor=sensor, motion=synth.Motion.translate(5.0))
    cfg = capi.Config.kitti()
    cfg.num_columns = cols
    hit = stream.hit[: cols * nframes].reshape(nframes, cols * rows)
    semantic = [np.where(h == 1, 40, np.where(h == 2, 50, np.where(h >= 3, 80, 0))).astype(np.uint16) for h in hit]
    euclid = [np.where(h >= 3, h - 2, 0).astype(np.uint32) for h in hit]
    F = stream.n_firings
    uidx_by_firing = np.full((F, rows), 2 ** 64 - 1, dtype=np.uint64)
    for f in range(cols * nframes):
        fr, k = divmod(f, cols)
        uidx_by_firing[f] = (np.uint64(3) << np.uint64(48)) | (np.uint64(fr) << np.uint64(32)) | (np.uint64(k * rows) + np.arange(rows, dtype=np.uint64))

    def run(engine_like, read_cols, evaluate):
        sc = evaluation.FrameScatter(3, [cols * rows] * nframes, semantic, euclid, evaluate=evaluate)
        for f0 in range(0, F, 500):
            assert engine_like.add_firings(stream.xyz[f0:f0 + 500], stream.intensity[f0:f0 + 500], stream.poses[f0:f0 + 500]) == 0
            ev = engine_like.drain_events()
            pub = ev[(ev["type"] == capi.EV_PUBLISH_COLUMNS) & (ev["b"] >= ev["a"])]
            if not len(pub):
                continue
            lo, hi = int(pub["a"].min()), int(pub["b"].max())
            a = read_cols(lo, hi)
            src = a["source_firing"]
            uidx = np.where(src >= 0, uidx_by_firing[np.maximum(src, 0), np.arange(rows)[None, :]], np.uint64(2 ** 64 - 1))
            sc.add_columns(uidx, a["ground_point_label"], a["id"])
        return sc.records

    eng = Engine(cfg, rows)
    rec_gpu = run(eng, lambda lo, hi: eng.read_columns(lo, hi, fields=["source_firing", "ground_point_label", "id"]), evaluation.eval_frame)
    orc = pyoracle.Oracle(cfg, rows)
    rec_cpu = run(orc, lambda lo, hi: orc.read_published(lo, hi, fields=["source_firing", "ground_point_label", "id"]), pyoracle.eval_frame)
    assert len(rec_gpu) == len(rec_cpu) == nframes - 1 + (1 if len(rec_cpu) == nframes else 0)
    assert np.array_equal(np.array(rec_gpu).view(np.uint64), np.array(rec_cpu).view(np.uint64))
    s = evaluation.summarize(np.array(rec_gpu)[:, 2:])
    assert 0.9 < s["recall"][0] <= 1.0 and 0.9 < s["accuracy"][0] <= 1.0  # flat synthetic ground is easy


def test_records_gather_over_rccl_through_the_c_abi():
    """cc_eval_gather_records: the one collective of the system reachable from C / C++ harnesses — a communicator made with RCCL's own
    API (world size 1: a box has one GPU), one ncclAllGather of a padded block, records and counts back on the host."""
    import ctypes as C
    from continuous_clustering_amd import load_library
    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    assert rccl is not None, "no RCCL library on this box"

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    L = load_library()
    L.cc_eval_gather_records.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    rec = np.ascontiguousarray(rng.random((37, 8)))
    rec[:, 0], rec[:, 1] = 4, np.arange(37)
    out = np.zeros((1, 64, 8))
    counts = np.zeros(1, dtype=np.int64)
    assert L.cc_eval_gather_records(comm, 1, 0, rec.ctypes.data, 37, 64, out.ctypes.data, counts.ctypes.data) == 0
    assert counts[0] == 37 and np.array_equal(out[0, :37].view(np.uint64), rec.view(np.uint64)) and not out[0, 37:].any()
    assert L.cc_eval_gather_records(comm, 1, 0, rec.ctypes.data, 37, 16, out.ctypes.data, counts.ctypes.data) != 0  # more records than capacity
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
