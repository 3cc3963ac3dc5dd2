"""Per-shape A/B of the GEMM tile geometries on the step's GEMM shapes (B = 256), interleaved rounds in one process, next to the
vendor library's GEMM (torch.matmul -> hipBLASLt / rocBLAS: a YARDSTICK for tools only, never on the product path) under the same
cold protocol (600 MB flush in front of every timed launch) and hot (10 back-to-back launches).  The vendor column is the bare product
(no bias, no GELU / residual epilogue; a grouped launch = its problems one after the other), so it is a lower bound for what the fused
launch could cost with the vendor's kernel.
usage: bench_shapes.py [geometry codes ...]   (default: 0 = heuristic, 261)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from volta_amd import _lib as L, ops

GEOS = [int(x) for x in sys.argv[1:]] or [0, 261]
SHAPES = [  # name, layout, epilogue, [(M, N, K), ...]
    ("image projection fwd", L.NT, L.EPI_BF16, [(9472, 768, 2048)]),
    ("text qkv fwd", L.NT, L.EPI_BF16, [(5120, 2304, 768)]),
    ("text out fwd", L.NT, L.EPI_BF16, [(5120, 768, 768)]),
    ("text ffn-up gelu", L.NT, L.EPI_GELU, [(5120, 3072, 768)]),
    ("text ffn-down", L.NT, L.EPI_BF16, [(5120, 768, 3072)]),
    ("text ffn-down dgrad mulr", L.NN, L.EPI_MULR, [(5120, 3072, 768)]),
    ("text ffn-up dgrad addr", L.NN, L.EPI_ADDR, [(5120, 768, 3072)]),
    ("text out dgrad", L.NN, L.EPI_BF16, [(5120, 768, 768)]),
    ("text qkv dgrad addr", L.NN, L.EPI_ADDR, [(5120, 768, 2304)]),
    ("dual qkv fwd", L.NT, L.EPI_BF16, [(5120, 2304, 768), (9472, 2304, 768)]),
    ("dual out fwd", L.NT, L.EPI_BF16, [(5120, 768, 768), (9472, 768, 768)]),
    ("dual ffn-up gelu", L.NT, L.EPI_GELU, [(5120, 3072, 768), (9472, 3072, 768)]),
    ("dual ffn-down", L.NT, L.EPI_BF16, [(5120, 768, 3072), (9472, 768, 3072)]),
    ("dual ffn-down dgrad mulr", L.NN, L.EPI_MULR, [(5120, 3072, 768), (9472, 3072, 768)]),
    ("dual ffn-up dgrad addr", L.NN, L.EPI_ADDR, [(5120, 768, 3072), (9472, 768, 3072)]),
    ("dual out dgrad", L.NN, L.EPI_BF16, [(5120, 768, 768), (9472, 768, 768)]),
    ("dual qkv dgrad addr", L.NN, L.EPI_ADDR, [(5120, 768, 2304), (9472, 768, 2304)]),
    ("ffn wgrad x4 (no split)", L.TN, L.EPI_F32, [(768, 3072, 5120), (768, 3072, 9472), (3072, 768, 5120), (3072, 768, 9472)]),
    ("attn wgrad x4 (no split)", L.TN, L.EPI_F32, [(768, 768, 5120), (768, 768, 9472), (2304, 768, 5120), (2304, 768, 9472)]),
]


def make(layout, epi, shapes):
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    probs, keep, fl, vend = [], [], 0.0, []
    for M, N, K in shapes:
        if layout == L.NT:
            A, B = rnd(M, K), rnd(N, K)
        elif layout == L.NN:
            A, B = rnd(M, K), rnd(K, N)
        else:
            A, B = rnd(K, M), rnd(K, N)
        f32 = epi in (L.EPI_F32, L.EPI_F32_ACC)
        Cb = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        C2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == L.EPI_GELU else None
        R = rnd(M, N) if epi in (L.EPI_MULR, L.EPI_ADDR) else None
        bias = torch.zeros(N, device="cuda") if layout != L.TN else None
        bg = torch.zeros(M, device="cuda") if layout == L.TN else None
        probs.append(ops.gemm_problem(A, B, Cb, layout, M, N, K, bias=bias, C2=C2, R=R, bias_grad=bg))
        keep += [A, B, Cb, C2, R, bias, bg]
        fl += 2.0 * M * N * K
        vend.append((A, B, torch.empty(M, N, device="cuda", dtype=torch.bfloat16)))
    return (L.GemmProblem * len(probs))(*probs), keep, fl, vend


def vendor_call(layout, vend):
    for A, B, out in vend:
        if layout == L.NT:
            torch.matmul(A, B.t(), out=out)
        elif layout == L.NN:
            torch.matmul(A, B, out=out)
        else:
            torch.matmul(A.t(), B, out=out)


def time_vendor(layout, vend, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        vendor_call(layout, vend)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def time_one(layout, epi, arr, n, geo, iters=10):
    st = ops.stream_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.check(L.lib.vk_gemm_grouped_ex(layout, epi, arr, n, geo, st))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# a 600 MB scratch write between rounds pushes operands out of L2 / Infinity Cache: the step runs every GEMM on cold operands
flush = torch.empty(300 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")
print("%-28s %s  vendor cold us (TF/s) hot   own-cold / vendor-cold" % ("shape", "  ".join("geo %-4d us (TF/s)" % g for g in GEOS)))
for name, layout, epi, shapes in SHAPES:
    arr, keep, fl, vend = make(layout, epi, shapes)
    best = {g: [] for g in GEOS}
    vbest = []
    for g in GEOS:
        time_one(layout, epi, arr, len(shapes), g, 2)
    time_vendor(layout, vend, 3)
    for rnd_ in range(5):
        for g in GEOS:
            flush.fill_(1.0)
            best[g].append(time_one(layout, epi, arr, len(shapes), g, 1))
        flush.fill_(1.0)
        vbest.append(time_vendor(layout, vend, 1))
    hot = {g: time_one(layout, epi, arr, len(shapes), g, 10) for g in GEOS}
    vhot = time_vendor(layout, vend, 10)
    vc = sorted(vbest)[2]
    print("%-28s %s  %6.1f (%4.0f) hot %6.1f   %.2f" % (name, "  ".join("%6.1f (%4.0f) hot %6.1f" % (sorted(best[g])[2], fl / sorted(best[g])[2] / 1e6, hot[g]) for g in GEOS),
                                                 vc, fl / vc / 1e6, vhot, sorted(best[GEOS[0]])[2] / vc), flush=True)
