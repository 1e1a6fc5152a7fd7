"""CPU: `python bench.py --gpus 2` must start two ranks by itself (torch.distributed.run, rendezvous on 127.0.0.1), rank 0 prints ONE JSON line
whose value is the whole job's (cells of both ranks / slowest rank), with the strong-split leg dealing --total-streams streams s -> rank s mod N.
Runs with tests/bench_stub.py in place of the HIP engine and gloo in place of RCCL (--stub-engine): the launcher and the aggregation are what is
under test, not a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-engine", "--streams", "3", "--firings", "40", "--steps", "4",
                        "--warmup", "1", "--no-s128", *extra], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_aggregates():
    out = run_bench("--gpus", "2", "--total-streams", "5")
    assert out["n_gpus"] == 2 and out["rccl_world"] == 2 and out["scaling"] == "weak"
    assert out["data"].startswith("stub")
    # weak leg: 3 streams per rank, 4 timed steps of 40 firings x 64 rows on each rank
    per_rank_cells = 3 * 40 * 64 * 4
    assert out["cells_published"] == 2 * per_rank_cells
    assert [r["cells"] for r in out["per_rank"]] == [per_rank_cells] * 2 and [r["streams"] for r in out["per_rank"]] == [3, 3]
    assert len(out["per_rank_value"]) == 2
    # the job's time is the slowest rank's (the stub's rank 1 sleeps twice as long per step): >= 4 steps x 20 ms
    assert out["ms_per_step"] >= 19.0
    assert abs(out["value"] - out["cells_published"] / (out["ms_per_step"] * 4 / 1e3) / 1e6) < 1e-6 * out["value"] + 1e-9
    assert out["per_rank_value"][0] > out["per_rank_value"][1] * 1.3
    # strong split: 5 streams in all, stream s on rank s mod 2 -> 3 + 2
    ss = out["strong_split"]
    assert ss["total_streams"] == 5 and ss["streams_per_gpu"] == [3, 2] and ss["scaling"] == "strong"
    assert ss["value"] > 0 and len(ss["per_rank_value"]) == 2


def test_gpus_1_is_one_process_without_a_process_group():
    out = run_bench("--gpus", "1", "--total-streams", "3")
    assert out["n_gpus"] == 1 and out["rccl_world"] == 0
    assert out["cells_published"] == 3 * 40 * 64 * 4
    assert out["strong_split"]["streams_per_gpu"] == [3] and abs(out["strong_split"]["value"] - out["value"]) < 1e-4 * out["value"]


def test_under_a_launcher_the_world_is_the_launchers():
    """The driver's own command shape for N > 1: torch.distributed.run around bench.py --gpus N (RANK set: bench.py must not spawn again)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-engine", "--streams", "2", "--firings", "30",
                        "--steps", "2", "--warmup", "1", "--no-s128", "--no-strong-split"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["cells_published"] == 2 * 2 * 30 * 64 * 2 and "strong_split" not in out
