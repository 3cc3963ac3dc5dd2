"""Data-parallel wrapper with MORE THAN ONE RANK on a one-GPU box: every rank on cuda:0, collectives over gloo (RCCL refuses two ranks on one
device).  Runs the assertions of tests/test_ddp_rccl2_gpu.py -- wrapper gradients == mean of the ranks' plain backward passes for all-reduce,
reduce-scatter + all-gather and the bf16 wire; replicas bit-identical afterwards -- from processes started BEFORE anything touches the GPU:

    python tools/ddp_check.py [--ranks 2]"""
import argparse
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    a = ap.parse_args()
    if "RANK" not in os.environ:                      # launcher: has not imported torch.cuda, starts the ranks as children
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ranks", str(a.ranks)],
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(a.ranks), VK_DDP_CHECK_PORT=str(port))) for r in range(a.ranks)]
        rcs = [p.wait() for p in procs]
        print("ddp_check: %d ranks on one GPU over gloo: %s" % (a.ranks, "OK" if not any(rcs) else "FAILED %r" % rcs))
        raise SystemExit(1 if any(rcs) else 0)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ddp_rccl2_gpu import _worker
    _worker(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["VK_DDP_CHECK_PORT"]), shared_gpu=True)
    print("rank %s: wrapper gradients equal the mean of the plain backward passes in all three modes; three clip + AdamW steps under "
          "mode zero1 leave master weights, bf16 copies and (gathered) moments equal to the unsharded wrapper's (<= 1e-6: float atomics in the backward) and bit-identical on every rank" % os.environ["RANK"], flush=True)


if __name__ == "__main__":
    main()
