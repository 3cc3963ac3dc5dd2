"""When do the workgroups of a soft-started FFN-down run relative to the FFN-up in front of it?  Study build (VK_LIB=study):
gemm256p writes its four phase stamps per tile, gemm256k (the consumer) its start / dependency-met / end stamps.
    VK_LIB=study python tools/stamp_soft.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from volta_amd import _lib as L, ops
import test_gemm_gpu as T


def main():
    L.lib.vk_gemm_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    L.lib.vk_gemm_set_kstamp_buffer.argtypes = [ctypes.c_void_p]
    pst = torch.zeros(256 * 8 * 4, dtype=torch.int64, device="cuda")
    kst = torch.zeros(256 * 4, dtype=torch.int64, device="cuda")
    rows = (5120, 9472)
    for soft in (False, True):
        up, down, cnt, sigs, outs, keep = T._ffn_pair(L, ops, rows, 3072, 768, seed=3, soft=soft)
        for it in range(4):
            pst.zero_(); kst.zero_()
            if soft:
                cnt[1:].zero_()
            torch.cuda.synchronize()
            L.lib.vk_gemm_set_stamp_buffer(pst.data_ptr())
            assert L.lib.vk_gemm_set_kstamp_buffer(kst.data_ptr()) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm_grouped(L.NT, L.EPI_GELU, up, geometry=258)
            ops.gemm_grouped(L.NT, L.EPI_BF16, down, geometry=259 | (L.GEMM_SOFT_START if soft else 0))
            e1.record()
            torch.cuda.synchronize()
            L.lib.vk_gemm_set_stamp_buffer(None)
            L.lib.vk_gemm_set_kstamp_buffer(None)
            p = pst.view(256, 8, 4).cpu().double() / 100.0
            k = kst.view(256, 4).cpu().double() / 100.0
            t0 = p[:, 0, 0][p[:, 0, 0] > 0].min()
            pend = p[:, :, 3].max(dim=1).values - t0                      # each producer workgroup's last epilogue end
            kk = k[k[:, 0] > 0] - t0
            q = lambda x, f: float(x.sort().values[int(f * (len(x) - 1))])
            print("%s it %d: pair %.1f us (events) | producer workgroups end %.1f / %.1f / %.1f (min / median / max) | consumer start %.1f / %.1f / %.1f, "
                  "dependency met %.1f / %.1f / %.1f, end %.1f / %.1f / %.1f" % (
                      "soft  " if soft else "fenced", it, e0.elapsed_time(e1) * 1e3, q(pend, 0), q(pend, .5), q(pend, 1),
                      q(kk[:, 0], 0), q(kk[:, 0], .5), q(kk[:, 0], 1), q(kk[:, 1], 0), q(kk[:, 1], .5), q(kk[:, 1], 1), q(kk[:, 2], 0), q(kk[:, 2], .5), q(kk[:, 2], 1)), flush=True)


def chain():
    """The same pair as ONE persistent launch (vk_gemm_chain): producer rounds in stamp slots 0-3, consumer rounds in slots 4-7."""
    L.lib.vk_gemm_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    pst = torch.zeros(256 * 8 * 4, dtype=torch.int64, device="cuda")
    P, Cn, cnt, sigs, outs, keep = T._chain_pair(L, ops, (5120, 9472), 3072, 768, 3, "fwd")
    flush = torch.empty(300 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")
    for it in range(6):
        pst.zero_(); cnt[1:].zero_()
        if it >= 3:
            flush.fill_(1.0)
        torch.cuda.synchronize()
        L.lib.vk_gemm_set_stamp_buffer(pst.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_chain(L.NT, L.EPI_GELU, P, L.EPI_BF16, Cn)
        e1.record()
        torch.cuda.synchronize()
        L.lib.vk_gemm_set_stamp_buffer(None)
        p = pst.view(256, 8, 4).cpu().double() / 100.0
        t0 = p[:, 0, 0][p[:, 0, 0] > 0].min()
        pend = p[:, :4, 3].max(dim=1).values - t0
        has_c = p[:, 4, 0] > 0
        cs, cw, ce = p[has_c, 4, 0] - t0, p[has_c, 4, 1] - t0, p[has_c, 4, 3] - t0
        q = lambda x, f: float(x.sort().values[int(f * (len(x) - 1))])
        print("chain %s it %d: launch %.1f us (events) | producer tiles end %.1f / %.1f / %.1f (min / median / max) | %d consumer tiles: K loop starts %.1f / %.1f / %.1f, ends %.1f / %.1f / %.1f, tile ends %.1f / %.1f / %.1f" % (
            "cold" if it >= 3 else "hot ", it, e0.elapsed_time(e1) * 1e3, q(pend, 0), q(pend, .5), q(pend, 1), int(has_c.sum()), q(cs, 0), q(cs, .5), q(cs, 1), q(cw, 0), q(cw, .5), q(cw, 1), q(ce, 0), q(ce, .5), q(ce, 1)), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "chain":
    chain()
else:
    main()
