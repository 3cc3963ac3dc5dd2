"""Which operand's residency decides the gap between a "hot" GEMM launch (operands in L2 / Infinity Cache) and a "cold" one (all of them in HBM)?
Single launches bracketed by events, after a 600 MB flush, with chosen operands touched again after the flush:
  cold | A warm | B warm | C (+ R) warm | all warm;  and the cold case with the leading dimension of A padded by 64 elements.
usage: bench_cold_parts.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L, ops

SHAPES = [
    ("dual ffn-down NT K=3072", L.NT, L.EPI_BF16, [(5120, 768, 3072), (9472, 768, 3072)]),
    ("dual ffn-up gelu NT K=768", L.NT, L.EPI_GELU, [(5120, 3072, 768), (9472, 3072, 768)]),
    ("dual ffn-up dgrad NN K=3072 +R", L.NN, L.EPI_ADDR, [(5120, 768, 3072), (9472, 768, 3072)]),
    ("dual qkv fwd NT K=768", L.NT, L.EPI_BF16, [(5120, 2304, 768), (9472, 2304, 768)]),
]
flush = torch.empty(300 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")
sink = torch.zeros(1, device="cuda")


def make(layout, epi, shapes, pad):
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    probs, groups = [], dict(A=[], B=[], C=[])
    for M, N, K in shapes:
        Afull = rnd(M, K + pad)
        A = Afull[:, :K]
        B = rnd(N, K) if layout == L.NT else rnd(K, N)
        Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        C2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == L.EPI_GELU else None
        R = rnd(M, N) if epi in (L.EPI_MULR, L.EPI_ADDR) else None
        bias = torch.zeros(N, device="cuda")
        p = ops.gemm_problem(A, B, Cb, layout, M, N, K, bias=bias, C2=C2, R=R)
        probs.append(p)
        groups["A"].append(Afull); groups["B"].append(B)
        groups["C"] += [t for t in (Cb, C2, R) if t is not None]
    return (L.GemmProblem * len(probs))(*probs), groups


def one(layout, epi, arr, n, groups, warm):
    flush.fill_(1.0)
    for key in warm:
        for t in groups[key]:
            sink.add_(t.sum(dtype=torch.float32))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(L.lib.vk_gemm_grouped_ex(layout, epi, arr, n, 0, ops.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for name, layout, epi, shapes in SHAPES:
    for pad in (0, 64):
        arr, groups = make(layout, epi, shapes, pad)
        row = []
        for warm in ((), ("A",), ("B",), ("C",), ("A", "B"), ("A", "B", "C")):
            ts = sorted(one(layout, epi, arr, len(shapes), groups, warm) for _ in range(7))
            row.append("%s %6.1f" % ("+".join(warm) or "cold", ts[3]))
        print("%-32s lda pad %2d: %s" % (name, pad, "   ".join(row)), flush=True)
