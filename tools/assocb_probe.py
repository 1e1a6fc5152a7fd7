"""GPU box: the batch-parallel association kernel (k_assocb) on every named parity case — engine vs oracle through tests/util.run_and_compare —
with the share of columns it took and how often it handed a batch to the serial kernel; then kernel times at several stream counts.
Usage: python tools/assocb_probe.py [cases|times|all]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
what = sys.argv[1] if len(sys.argv) > 1 else "all"

if what in ("cases", "all"):
    import util, cases
    from oracle import pyoracle
    pyoracle.build()
    bad = 0
    names = cases.ALL_CASES + cases.RING_WRAP_CASES + cases.GOLDEN_CASES
    if os.environ.get("CC_CASES"):
        names = os.environ["CC_CASES"].split(",")
    for name in names:
        stream, cfg, tf = cases.build_case(name)
        box = {}
        def setup(e, box=box):
            box["e"] = e
            if os.environ.get("CC_AB") is not None:
                e.set_option("assoc_batch", int(os.environ["CC_AB"]))
        for chunks in ([stream.sensor.num_columns], [97, 1, 200]):
            t0 = time.time()
            try:
                summ = util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=tf, engine_setup=setup)
                bc = box["e"].batch_counters()
                es = summ["engine_state"]
                print(f"{name:34s} chunks {str(chunks):14s} ok   published {summ['published_columns']:6d} batch columns {bc['batch_columns']:6d} "
                      f"bails {bc['batch_bails']:4d} {bc['bail_reasons'][1:7]} serial {es['error_b']:5d}  {time.time() - t0:.1f}s", flush=True)
            except AssertionError as ex:
                bad += 1
                bc = box["e"].batch_counters() if "e" in box else {}
                print(f"{name:34s} chunks {str(chunks):14s} FAIL {bc} {str(ex)[:300]}", flush=True)
    print("failures:", bad, flush=True)

if what in ("times", "all"):
    import torch
    from continuous_clustering_amd import Engine, capi, synth
    import bench
    sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
    F, NB = 2200, 5
    for S in (32, 128, 256):
        xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(S)], F, NB)
        torch.cuda.synchronize()
        for ab in (0, 1):
            for pipe in (0, 1):
                e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", pipe); e.set_option("assoc_batch", ab)
                e.add_firings_device(F, xyz[0], inten[0], poses[0]); e.sync()
                e.enable_timing(True)
                t0 = time.time()
                for b in range(1, NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
                e.sync(); dt = (time.time() - t0) / (NB - 1)
                k = e.kernel_times(); bc = e.batch_counters(); tot = e.totals()
                print("streams", S, "assoc_batch", ab, "pipeline", pipe, "ms/step %.3f" % (dt * 1e3), "Mpoints/s %.0f" % (S * F * 64 / dt / 1e6), bc,
                      "serial", tot["serial_columns"], "clusters", tot["clusters_finished"],
                      {n: round(v / k["batches"], 3) for n, v in k.items() if n.endswith("_ms")}, flush=True)
                e.close()
        del xyz, inten, poses
