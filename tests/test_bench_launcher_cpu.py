"""`python bench.py --gpus N` must start N ranks by itself (the driver's contract, and what a user types) and report them.
Rehearsed on CPU: VK_BENCH_DRY_RUN=1 replaces the model step by a small gloo all-reduce and keeps the launcher, the rendezvous on
127.0.0.1, the barriers, the max-over-ranks timing and the one JSON line of rank 0 (train_concap.py:114-130,150-155)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = dict(os.environ, VK_BENCH_DRY_RUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, printed by rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 2 * 256 and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and "dry_run" in out


def test_single_rank_needs_no_launcher():
    r = _run(["--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["parallelism"] == "dp1" and out["config"]["global_batch"] == 256


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], extra_env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
