"""Why does a GEMM take 15-30 % longer inside the step than in a back-to-back loop?  Times ONE launch (HIP events around it)
after different predecessors: itself (hot code, hot data), another kernel (cold instruction cache), a large copy (cold L2 / MALL)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L, ops


def make(layout, epi, M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    A = rnd(M, K)
    B = rnd(N, K) if layout == L.NT else rnd(K, N)
    Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    p = ops.gemm_problem(A, B, Cb, layout, M, N, K, bias=bias)
    arr = (L.GemmProblem * 1)(p)
    keep = (A, B, Cb, bias)
    return lambda: L.check(L.lib.vk_gemm_grouped(layout, epi, arr, 1, ops.stream_ptr())), keep


def timed(fn, before, iters=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for _ in range(3):
        before(); fn()
    torch.cuda.synchronize()
    for e0, e1 in ev:
        before()
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return ts[len(ts) // 2]


def main():
    M = 9472
    d, x = (torch.randn(M, 768, device="cuda").bfloat16() for _ in range(2))
    y, z = (torch.empty(M, 768, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    gam, bet = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ln = lambda: ops.ln_fwd(d, x, gam, bet, y, z, mean, rstd, M, 768)
    big_a = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    big_b = torch.empty_like(big_a)
    flush = lambda: big_b.copy_(big_a)
    other, keep2 = make(L.NN, L.EPI_BF16, 4096, 1024, 1024, 5)          # a different kernel template on other buffers
    tiny = torch.zeros(64, device="cuda")
    nothing = lambda: None
    for name, layout, Mr, N, K in (("text ffn-down fwd", L.NT, 5120, 768, 3072), ("text out fwd", L.NT, 5120, 768, 768),
                                   ("text qkv fwd", L.NT, 5120, 2304, 768), ("text+vis-sized qkv", L.NT, 14592, 2304, 768),
                                   ("text+vis ffn-down", L.NT, 14592, 768, 3072), ("text+vis ffn-up", L.NT, 14592, 3072, 768)):
        fn, keep = make(layout, L.EPI_BF16, Mr, N, K, 1)
        A, Bw = keep[0], keep[1]
        A2, B2 = A.clone(), Bw.clone()
        res = [("itself", timed(fn, fn)), ("nothing (event gap)", timed(fn, nothing)), ("LayerNorm kernel", timed(fn, ln)),
               ("other GEMM template", timed(fn, other)), ("600 MB copy", timed(fn, flush)),
               ("copy then itself", timed(fn, lambda: (flush(), fn()))),
               # what does the producer -> consumer hand-over of the step look like?  A freshly WRITTEN by another kernel / freshly READ
               ("copy, then A rewritten", timed(fn, lambda: (flush(), A.copy_(A2)))),
               ("copy, then A read", timed(fn, lambda: (flush(), A.float().sum()))),
               ("copy, then A and B rewritten", timed(fn, lambda: (flush(), A.copy_(A2), Bw.copy_(B2)))),
               ("copy, then B read", timed(fn, lambda: (flush(), Bw.float().sum()))),
               ("copy, A rewritten, B read", timed(fn, lambda: (flush(), A.copy_(A2), Bw.view(torch.int32).sum()))),
               ("copy, then B rewritten", timed(fn, lambda: (flush(), Bw.copy_(B2))))]
        print("%-20s M=%5d N=%4d K=%4d  " % (name, Mr, N, K) + "  ".join("%s: %.1f us" % r for r in res), flush=True)


main()
