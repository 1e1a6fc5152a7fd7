"""Where the time of a few-stream leg goes: host timestamps after every cc_engine_add_firings_device call and after the final sync
(bench.py's timed region, call by call). usage: python tools/step_probe.py [streams] [steps] [timing 0/1] [lib]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continuous_clustering_amd as _cca
if len(sys.argv) > 4:
    _cca.LIB_PATH = os.path.join(os.path.dirname(_cca.LIB_PATH), sys.argv[4])
from continuous_clustering_amd import Engine, capi, synth
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
timing = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
F, W = 2200, 3
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(S)], F, W + K)
torch.cuda.synchronize()
mode = os.environ.get("PROBE_MODE", "close")
if mode == "dummy":
    d = Engine(cfg, 64, S); d.close(); del d
keep = []
for rep in range(4):
    if mode == "reuse" and rep > 0:
        e.reset(64)
    else:
        e = Engine(cfg, 64, S); e.record_events(False); bench.engine_options(e)
    for b in range(W): e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0
    before = e.totals()
    if timing: e.enable_timing(True)
    if mode == "heat":  # a second of full-GPU work right before the timed region (does the shader clock / power state matter?)
        a = torch.randn(8192, 8192, device="cuda"); t_h = time.perf_counter()
        while time.perf_counter() - t_h < 1.0:
            for _ in range(10): a = (a @ a) * 1e-4
            torch.cuda.synchronize()
        del a
    if mode == "idle": time.sleep(1.0)
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    for b in range(W, W + K):
        e.add_firings_device(F, xyz[b], inten[b], poses[b]); t.append(time.perf_counter())
    rc = e.sync(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    cells = e.totals()["cells_published"] - before["cells_published"]
    d = [round((t[i + 1] - t[i]) * 1e3, 3) for i in range(len(t) - 1)]
    print(f"streams {S} steps {K} timing {timing} rep {rep}: total {1e3 * (t[-1] - t[0]):.3f} ms = {cells / (t[-1] - t[0]) / 1e6:.0f} Mpoints/s; calls {d[:-2]} sync {d[-2]} devsync {d[-1]}")
    if timing: print("   kernel ms per step", {k: round(v / K, 3) for k, v in e.kernel_times().items() if k.endswith("_ms")})
    if mode == "keep": keep.append(e)
    elif mode != "reuse": e.close()
    if mode == "sleep": time.sleep(2.0)
