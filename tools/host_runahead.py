"""How far does the host run ahead of the GPU?  N back-to-back launches of one kernel: wall time of the issuing loop against the GPU time of the launches.
(kernel arguments travel by value: a grouped GEMM launch carries ~4 KB of them, a LayerNorm launch ~300 B)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
M, Nn, K = 5120, 768, 768
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(Nn, K, device="cuda").bfloat16(); Cc = torch.empty(M, Nn, device="cuda", dtype=torch.bfloat16)
prob = (L.GemmProblem * 1)(ops.gemm_problem(A, B, Cc, L.NT, M, Nn, K, bias=torch.zeros(Nn, device="cuda")))
st = ops.stream_ptr()
x = torch.randn(5120 * 768, device="cuda")


def run(name, fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s issue loop %.2f ms, until the GPU is done %.2f ms (%.1f us per launch on the GPU)" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) / N * 1e6))


run("grouped GEMM (1 problem)", lambda: L.check(L.lib.vk_gemm_grouped_ex(L.NT, L.EPI_BF16, prob, 1, 0, st)))
run("torch mul_ (tiny kernarg)", lambda: x.mul_(1.0))
