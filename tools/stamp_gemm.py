"""Where does a tile of the persistent GEMM (gemm256p_kernel) spend its time?  Study build only (VK_LIB=study): the kernel writes four
s_memrealtime stamps (100 MHz) per workgroup and tile -- loop top, K loop done, ring drained + next prologue issued, epilogue done --
and this tool prints, per tile index of the walk, the median over workgroups of each segment, next to the launch's HIP-event time.
    VK_LIB=study python tools/stamp_gemm.py
Shapes: the multi-round grouped launches of a ViLBERT step (text 5120 + vision 9472 rows)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L, ops


def group(layout, epi, N, K, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    probs, keep = [], []
    for M in (5120, 9472):
        A = rnd(M, K)
        B = rnd(N, K) if layout == L.NT else rnd(K, N)
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        C2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == L.EPI_GELU else None
        R = rnd(M, N) if epi in (L.EPI_MULR, L.EPI_ADDR) else None
        bias = torch.zeros(N, device="cuda")
        probs.append(ops.gemm_problem(A, B, C, layout, M, N, K, bias=None if epi == L.EPI_MULR else bias, C2=C2, R=R))
        keep += [A, B, C, C2, R, bias]
    return (L.GemmProblem * 2)(*probs), keep


def main():
    L.lib.vk_gemm_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    stamps = torch.zeros(256 * 8 * 4, dtype=torch.int64, device="cuda")
    big_a = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    big_b = torch.empty_like(big_a)
    for name, layout, epi, N, K in (("ffn-up + GELU (f54)", L.NT, L.EPI_GELU, 3072, 768), ("qkv (f57)", L.NT, L.EPI_BF16, 2304, 768),
                                    ("ffn-down dgrad x gelu' (b30)", L.NN, L.EPI_MULR, 3072, 768), ("ffn-down (f55, one round)", L.NT, L.EPI_BF16, 768, 3072)):
        if os.environ.get("STAMP_ONLY") and os.environ["STAMP_ONLY"] not in name:
            continue
        arr, keep = group(layout, epi, N, K)
        variants = [("persistent", L.GEMM_PERSISTENT, 0), ("one tile per wg", L.GEMM_ONE_TILE_PER_WG, 0)]
        if os.environ.get("STAMP_DESYNC"):        # every other workgroup of an XCD starts N us late: are lockstep epilogues what costs?
            variants.insert(1, ("persistent, odd wgs %s us late" % os.environ["STAMP_DESYNC"], L.GEMM_PERSISTENT, int(os.environ["STAMP_DESYNC"]) * 100))
        for geo_name, geo, desync in variants:
            L.lib.vk_gemm_set_desync(desync)
            for cold in (False, True):
                call = lambda: L.check(L.lib.vk_gemm_grouped_ex(layout, epi, arr, 2, geo, ops.stream_ptr()))
                ts = []
                for it in range(8):
                    if cold:
                        big_b.copy_(big_a)
                    stamps.zero_()
                    L.lib.vk_gemm_set_stamp_buffer(stamps.data_ptr() if geo == L.GEMM_PERSISTENT else None)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); call(); e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                L.lib.vk_gemm_set_stamp_buffer(None)
                line = "%-30s %-30s %-5s launch %.1f us" % (name, geo_name, "cold" if cold else "hot", sorted(ts)[len(ts) // 2])
                if geo == L.GEMM_PERSISTENT:
                    st = stamps.view(256, 8, 4).cpu().double() / 100.0          # us
                    t_first = st[:, 0, 0][st[:, 0, 0] > 0].min()
                    on_time = ((torch.arange(256) >> 3) & 1) == 0
                    halves = (("", slice(None)),) if not desync else ((" [on-time]", on_time), (" [late]", ~on_time))
                    for ti, (tag, sel) in ((ti, h) for ti in range(4) for h in halves):
                        s = st[sel][:, ti]
                        live = s[:, 3] > 0
                        if int(live.sum()) == 0:
                            continue
                        s = s[live]
                        seg = [float((s[:, 1] - s[:, 0]).median()), float((s[:, 2] - s[:, 1]).median()), float((s[:, 3] - s[:, 2]).median())]
                        line += " | tile %d%s (%d wgs): K loop %.1f drain+next %.1f epilogue %.1f, ends at %.1f" % (
                            ti, tag, int(live.sum()), seg[0], seg[1], seg[2], float((s[:, 3] - t_first).median()))
                print(line, flush=True)


main()
