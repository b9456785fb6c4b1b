"""`python bench.py --gpus N` must start itself: with N > 1 and no launcher around it the parent spawns
`torch.distributed.run` with N ranks as a child before anything touches the GPU, relays rank 0's JSON line and propagates
failure.  Exercised here on CPU: gloo rendezvous on 127.0.0.1, a stand-in step (SPRK_BENCH_STUB), world size 2."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, stub="1", timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SPRK_BENCH_STUB=stub, SPRK_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, cwd=ROOT, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_bench_gpus2_launches_itself_and_reports_the_slowest_rank():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["per_gpu_batch"] == 32 and j["config"]["global_batch"] == 64
    assert j["ms_per_step"] >= 4.0                      # rank 1 sleeps 4 ms per step: the line carries the MAX over ranks
    assert abs(j["value"] - 64 * 5 / (j["ms_per_step"] * 5e-3)) < 1e-6 * j["value"]     # whole-job patches / that time
    assert "self-launch" in r.stderr and "--nproc-per-node 2" in r.stderr


def test_bench_global_batch_is_split_over_the_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--global-batch", "128"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["scaling"] == "strong" and j["config"] == {"workload": "stub", "per_gpu_batch": 64, "global_batch": 128}


def test_bench_exits_nonzero_when_a_rank_fails():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0"], stub="fail1")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 fails on purpose" in r.stderr
