"""`python bench.py --gpus 2` on the one-GPU box: rehearsal mode (VK_BENCH_SHARED_GPU=1: both ranks drive cuda:0, collectives over
gloo) runs the REAL step -- wrap-time broadcast, bucketed reduction with side-stream joins, max-over-ranks timing -- end to end
and must report n_gpus = 2; the timings of such a run mean nothing and the line says so."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_rehearsal_reports_two_gpus():
    env = dict(os.environ, VK_BENCH_SHARED_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "2", "--warmup", "1",
                        "--no-kernel-timing", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    assert "rehearsal" in out and all(x == x for x in out["losses_last_step"])      # finite losses
