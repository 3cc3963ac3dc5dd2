"""Micro-benchmarks of the HBM-bound kernels (LayerNorm fwd/bwd, attention fwd/bwd) at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L, ops


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def ln(M, H=768):
    dev = "cuda"
    d, x, dy = (torch.randn(M, H, device=dev).bfloat16() for _ in range(3))
    y, z, dz, dd = (torch.empty(M, H, device=dev, dtype=torch.bfloat16) for _ in range(4))
    g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    dg, db = torch.empty(H, device=dev), torch.empty(H, device=dev)
    part = torch.empty(L.lib.vk_ln_bwd_partial_rows(M) * 2 * H, device=dev)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    for p in (0.0, 0.1):
        drop = L.dropout_cfg(seed.data_ptr(), 3, p)
        tf = timeit(lambda: ops.ln_fwd(d, x, g, b, y, z, mean, rstd, M, H, drop=drop))
        tb = timeit(lambda: ops.ln_bwd(dy, z, mean, rstd, g, dz, dd, part, dg, db, M, H, drop=drop))
        tb2 = timeit(lambda: ops.ln_bwd(dy, z, mean, rstd, g, dz, None, part, dg, db, M, H, drop=drop))
        by = M * H * 2
        print("LN M=%5d p=%.1f  fwd %6.1f us (%.2f TB/s)  bwd %6.1f us (%.2f TB/s)  bwd-no-dd %6.1f us" % (M, p, tf, 4 * by / tf / 1e6, tb, 4 * by / tb / 1e6, tb2), flush=True)


def attn(gate, T=20, R=37, B=256, nh=12, H=768):
    dev = "cuda"
    Ls = [T, R]
    qkv = [torch.randn(B * Ls[m], 3 * H, device=dev).bfloat16() for m in range(2)]
    masks = [torch.zeros(B, Ls[m], device=dev) for m in range(2)]
    ctx = [torch.zeros(B * Ls[m], H, device=dev, dtype=torch.bfloat16) for m in range(2)]
    lse = [torch.zeros(B * nh * Ls[m], device=dev) for m in range(2)]
    dctx = [torch.randn(B * Ls[m], H, device=dev).bfloat16() for m in range(2)]
    dqkv = [torch.zeros(B * Ls[m], 3 * H, device=dev, dtype=torch.bfloat16) for m in range(2)]
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    for p in (0.0, 0.1):
        drops = [[L.dropout_cfg(seed.data_ptr(), 2 * i + j, p) for j in range(2)] for i in range(2)]
        a = ops.attn_args(qkv, Ls, masks, ctx, lse, B, nh, gate, drops, H)
        tf = timeit(lambda: ops.attn_fwd(a))
        tb = timeit(lambda: ops.attn_bwd(a, dctx, dqkv, Ls, B, gate, H))
        print("ATTN gate=%s p=%.1f  fwd %6.1f us  bwd %6.1f us" % (gate, p, tf, tb), flush=True)


def concap(B=256, T=20, Rl=36, F=2048, Cn=1601):
    """ConceptCap batch producer at the benchmark shape: bytes = raw features + class distributions read and written once."""
    from volta_amd.data import ConceptCapBatchProducer
    g = torch.Generator(device="cuda").manual_seed(0)
    caps = [[int(x) for x in torch.randint(1000, 30000, (int(n),))] for n in torch.randint(4, 19, (B + 100,))]
    prod = ConceptCapBatchProducer(caps, T, Rl, 30522)
    feat = torch.rand(B, Rl, F, device="cuda", generator=g)
    cls = torch.softmax(torch.randn(B, Rl, Cn, device="cuda", generator=g), -1)
    xy = torch.rand(B, Rl, 2, device="cuda", generator=g) * 300
    boxes = torch.cat([xy, xy + 50 + torch.rand(B, Rl, 2, device="cuda", generator=g) * 200], -1)
    nb = torch.full((B,), Rl, dtype=torch.int32, device="cuda")
    wh = torch.full((B, 2), 640.0, device="cuda")
    ci = torch.arange(B, dtype=torch.int32, device="cuda")
    us = timeit(lambda: prod(feat, cls, boxes, nb, wh, ci, 7), iters=20)
    by = 2 * 4 * (feat.numel() + cls.numel())
    print("CONCAP producer B=%d: %.1f us per batch (%.2f TB/s over %.0f MB read+written) = %.1f M pairs/s" % (B, us, by / us / 1e6, by / 1e6, B / us), flush=True)


if __name__ == "__main__":
    import ctypes
    if len(sys.argv) > 1 and sys.argv[1] == "ln":
        for M in (5120, 9472, 14592):
            ln(M)
        sys.exit(0)
    if not (len(sys.argv) > 1 and sys.argv[1] == "fwd"):
        concap()
        ln(5120)
        ln(9472)
    L.lib.vk_attn_set_bwd_occupancy.argtypes = [ctypes.c_int]
    L.lib.vk_attn_set_bwd_waves.argtypes = [ctypes.c_int]
    L.lib.vk_attn_set_fwd_occupancy.argtypes = [ctypes.c_int]
    L.lib.vk_attn_set_fwd_waves.argtypes = [ctypes.c_int]
    if len(sys.argv) > 1 and sys.argv[1] == "fwd":          # forward occupancy study
        for occ, wv in ((4, 0), (4, 4), (5, 0), (5, 4), (4, 0)):
            L.lib.vk_attn_set_fwd_occupancy(occ)
            L.lib.vk_attn_set_fwd_waves(wv)
            print("--- attn fwd: waves per SIMD targeted", occ, " waves per workgroup", wv or "one per query tile")
            attn([[1, 0], [0, 0]])
            attn([[1, 0], [0, 1]])
            attn([[0, 1], [1, 0]])
            attn([[1, 1], [1, 1]], T=20, R=37)
        sys.exit(0)
    for occ, wv in ((2, 4), (3, 4), (2, 8), (2, 4)):
        L.lib.vk_attn_set_bwd_occupancy(occ)
        L.lib.vk_attn_set_bwd_waves(wv)
        print("--- attn bwd: waves per SIMD targeted", occ, " waves per workgroup", wv)
        attn([[1, 0], [0, 0]])
        attn([[1, 0], [0, 1]])
        attn([[0, 1], [1, 0]])
