"""Micro-benchmark of vk_gemm_grouped on the GEMM shapes of one ctrl_vilbert_base step (B=256)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L, ops


def bench(name, layout, epi, M, N, K, iters=20):
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    if layout == L.NT:
        A, B = rnd(M, K), rnd(N, K)
    elif layout == L.NN:
        A, B = rnd(M, K), rnd(K, N)
    else:
        A, B = rnd(K, M), rnd(K, N)
    Cb = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == L.EPI_F32 else torch.bfloat16)
    C2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == L.EPI_GELU else None
    bias = torch.zeros(N, device="cuda")
    p = ops.gemm_problem(A, B, Cb, layout, M, N, K, bias=bias, C2=C2)
    import ctypes
    arr = (L.GemmProblem * 1)(p)
    rep = L.lib.vk_gemm_repeat          # launches issued back-to-back from native code (no interpreter time between kernels)
    rep.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    st = ops.stream_ptr()
    assert rep(layout, epi, arr, 1, 3, st) == 0, L.lib.vk_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert rep(layout, epi, arr, 1, iters, st) == 0
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-28s M=%5d N=%5d K=%5d  %8.1f us  %7.1f TF/s" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)


def sweep():
    """A/B of the tile geometries in one process: 128 x 128, 256 x 256 8-phase, legacy 16-wave 256 x 256."""
    import ctypes
    L.lib.vk_gemm_set_tile.argtypes = [ctypes.c_int]
    for edge in ([int(x) for x in sys.argv[2:]] or (258, 259, 256, 128, 258)):
        L.lib.vk_gemm_set_tile(edge)
        print("=== tile", edge, flush=True)
        for Mrows, tag in ((5120, "text"), (9472, "vis")):
            bench(tag + " qkv fwd NT", L.NT, L.EPI_BF16, Mrows, 2304, 768)
            bench(tag + " out fwd NT", L.NT, L.EPI_BF16, Mrows, 768, 768)
            bench(tag + " ffn-up fwd NT gelu", L.NT, L.EPI_GELU, Mrows, 3072, 768)
            bench(tag + " ffn-down fwd NT", L.NT, L.EPI_BF16, Mrows, 768, 3072)
            bench(tag + " ffn-up dgrad NN", L.NN, L.EPI_BF16, Mrows, 768, 3072)
            bench(tag + " ffn-down dgrad NN", L.NN, L.EPI_BF16, Mrows, 3072, 768)
            bench(tag + " ffn wgrad TN", L.TN, L.EPI_F32, 3072, 768, Mrows)
        bench("square 4096", L.NT, L.EPI_BF16, 4096, 4096, 4096)
        bench("square 4096 NN", L.NN, L.EPI_BF16, 4096, 4096, 4096)
        bench("square 4096 TN", L.TN, L.EPI_F32, 4096, 4096, 4096)
        bench("square 8192", L.NT, L.EPI_BF16, 8192, 8192, 8192, iters=5)
    L.lib.vk_gemm_set_tile(0)


def ablate():
    """Ablation of the 4-phase 256 x 256 kernel's main loop (tile 256, the only one that carries the switches): what does a K-tile cost
    without its DMA / MFMA / LDS reads?"""
    import ctypes
    L.lib.vk_gemm_set_tile.argtypes = [ctypes.c_int]
    L.lib.vk_gemm_set_debug.argtypes = [ctypes.c_int]
    L.lib.vk_gemm_set_tile(256)
    for dbg, tag in ((0, "full"), (1, "no DMA in loop"), (2, "no MFMA"), (4, "no LDS reads"), (3, "no DMA, no MFMA"), (5, "no DMA, no LDS reads"), (6, "no MFMA, no LDS reads"), (7, "barriers only"), (16, "exit at once"), (7 + 32, "no K loop"), (7 + 32 + 8, "no K loop, no epilogue"), (7 + 8, "barriers only, no epilogue"), (8, "no epilogue"), (0, "full")):
        L.lib.vk_gemm_set_debug(dbg)
        print("=== ablation:", tag, flush=True)
        bench("text qkv fwd NT", L.NT, L.EPI_BF16, 5120, 2304, 768)
        bench("text ffn-down dgrad NN", L.NN, L.EPI_BF16, 5120, 3072, 768)
        bench("deep K one wave of tiles", L.NT, L.EPI_BF16, 4096, 4096, 8192)
        bench("square 8192", L.NT, L.EPI_BF16, 8192, 8192, 8192, iters=5)
    L.lib.vk_gemm_set_debug(0)
    L.lib.vk_gemm_set_tile(0)


def peak():
    """Sustained MFMA rate on register operands (no memory traffic): the ceiling of any GEMM main loop here."""
    import ctypes
    f = L.lib.vk_mfma_peak
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    seed = torch.randint(1, 2 ** 31 - 1, (64,), device="cuda", dtype=torch.int32)
    out = torch.zeros(4, device="cuda")
    for blocks in (256, 512):
        for iters in (2000, 20000):
            f(seed.data_ptr(), out.data_ptr(), 100, blocks, ops.stream_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f(seed.data_ptr(), out.data_ptr(), iters, blocks, ops.stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            fl = blocks * 8 * iters * 16 * 16384.0
            print("mfma peak: %d blocks x 8 waves, %d iters: %.3f ms  %.0f TF/s" % (blocks, iters, ms, fl / ms / 1e9), flush=True)


def blas(name, layout, M, N, K, iters=20):
    """Vendor BLAS (torch.matmul -> hipBLASLt) on the same shape: a known-good yardstick for the headroom, never part of the product."""
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    if layout == L.NT:
        A, B = rnd(M, K), rnd(N, K).t()
    elif layout == L.NN:
        A, B = rnd(M, K), rnd(K, N)
    else:
        A, B = rnd(K, M).t(), rnd(K, N)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(A, B, out=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(A, B, out=C)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-28s M=%5d N=%5d K=%5d  %8.1f us  %7.1f TF/s  (vendor BLAS)" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)


def vendor():
    for Mrows, tag in ((5120, "text"), (9472, "vis")):
        blas(tag + " qkv fwd NT", L.NT, Mrows, 2304, 768)
        blas(tag + " out fwd NT", L.NT, Mrows, 768, 768)
        blas(tag + " ffn-up fwd NT", L.NT, Mrows, 3072, 768)
        blas(tag + " ffn-down fwd NT", L.NT, Mrows, 768, 3072)
        blas(tag + " ffn-up dgrad NN", L.NN, Mrows, 768, 3072)
        blas(tag + " ffn-down dgrad NN", L.NN, Mrows, 3072, 768)
        blas(tag + " ffn wgrad TN", L.TN, 3072, 768, Mrows)
        blas(tag + " out wgrad TN", L.TN, 768, 768, Mrows)
    blas("square 4096", L.NT, 4096, 4096, 4096)
    blas("square 8192", L.NT, 8192, 8192, 8192, iters=5)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        sweep()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "peak":
        peak()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":
        ablate()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vendor":
        vendor()
        sys.exit(0)
    for Mrows, tag in ((5120, "text"), (9472, "vis")):
        bench(tag + " qkv fwd NT", L.NT, L.EPI_BF16, Mrows, 2304, 768)
        bench(tag + " out fwd NT", L.NT, L.EPI_BF16, Mrows, 768, 768)
        bench(tag + " ffn-up fwd NT gelu", L.NT, L.EPI_GELU, Mrows, 3072, 768)
        bench(tag + " ffn-down fwd NT", L.NT, L.EPI_BF16, Mrows, 768, 3072)
        bench(tag + " ffn-up dgrad NN", L.NN, L.EPI_BF16, Mrows, 768, 3072)
        bench(tag + " ffn-down dgrad NN", L.NN, L.EPI_BF16, Mrows, 3072, 768)
        bench(tag + " ffn wgrad TN", L.TN, L.EPI_F32, 3072, 768, Mrows)
        bench(tag + " out wgrad TN", L.TN, L.EPI_F32, 768, 768, Mrows)
    bench("img-emb NT", L.NT, L.EPI_BF16, 9472, 768, 2048)
    bench("square 4096", L.NT, L.EPI_BF16, 4096, 4096, 4096)
    bench("square 8192", L.NT, L.EPI_BF16, 8192, 8192, 8192, iters=5)
