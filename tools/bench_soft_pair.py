"""FFN-up + GELU -> FFN-down as two launches behind the stream-order barrier against the soft boundary (VK_GEMM_SOFT_START + row-block
counters): time per pair over back-to-back repetitions, hot and with a 600 MB flush in front of every pair.
usage: python tools/bench_soft_pair.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from volta_amd import _lib as L, ops
import test_gemm_gpu as T


def time_pairs(up, down, cnt, soft, reps, flush=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    if flush is None:
        e0.record()
        for _ in range(reps):
            T._run_pair(L, ops, up, down, cnt, soft)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    ts = []
    for _ in range(reps):
        flush.fill_(1.0)
        if soft:
            cnt[1:].zero_()
        e0.record()
        ops.gemm_grouped(L.NT, L.EPI_GELU, up, geometry=258)
        ops.gemm_grouped(L.NT, L.EPI_BF16, down, geometry=259 | (L.GEMM_SOFT_START if soft else 0))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    flush = torch.empty(300 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")
    for rows in ((5120, 9472), (5120,)):
        up, down, cnt, sigs, outs, keep = T._ffn_pair(L, ops, rows, 3072, 768, seed=3, soft=True)
        upf, downf, _, _, outsf, keepf = T._ffn_pair(L, ops, rows, 3072, 768, seed=3, soft=False)
        for _ in range(3):
            time_pairs(up, down, cnt, True, 3); time_pairs(upf, downf, None, False, 3)
        for rnd_ in range(3):
            print("rows %-14s round %d: hot fenced %.1f us  soft %.1f us | cold fenced %.1f us  soft %.1f us" % (
                rows, rnd_, time_pairs(upf, downf, None, False, 10), time_pairs(up, down, cnt, True, 10),
                time_pairs(upf, downf, None, False, 7, flush), time_pairs(up, down, cnt, True, 7, flush)), flush=True)


main()
