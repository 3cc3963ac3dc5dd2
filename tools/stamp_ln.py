"""Where does the LayerNorm backward spend its time?  Study build only (VK_LIB=study): every workgroup writes four s_memrealtime stamps
(100 MHz) -- entry, first row's statistics done (its operands have arrived), last row stored, record written -- and this tool prints the
distribution over the workgroups of the launch next to the launch's HIP-event time.
    VK_LIB=study python tools/stamp_ln.py
Shape: the paired launch of a ViLBERT sub-layer (text 5120 + vision 9472 rows, 768 wide), dropout on, column reductions deferred."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L, ops


def args(M, H, seed_t, site, defer=True):
    g = torch.Generator(device="cuda").manual_seed(M)
    rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
    t = dict(dy=rnd(M, H), z=rnd(M, H), mean=torch.zeros(M, device="cuda"), rstd=torch.ones(M, device="cuda"), gamma=torch.ones(H, device="cuda"),
             dz=torch.empty(M, H, device="cuda", dtype=torch.bfloat16), dd=torch.empty(M, H, device="cuda", dtype=torch.bfloat16),
             part=torch.empty(L.lib.vk_ln_bwd_partial_rows(M) * 2 * H, device="cuda"), dg=torch.zeros(H, device="cuda"), db=torch.zeros(H, device="cuda"))
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    drop = L.dropout_cfg(seed_t.data_ptr(), site, 0.1)
    a = L.LnBwdArgs(p(t["dy"]), p(t["z"]), p(t["mean"]), p(t["rstd"]), p(t["gamma"]), p(t["dz"]), p(t["dd"]), p(t["part"]), p(t["dg"]), p(t["db"]), None,
                    M, H, M, 0, 1.0, 2 if defer else 0, drop, ops._segs(drop, None))
    return a, t


def main():
    L.lib.vk_ln_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    seed_t = torch.tensor([1234], dtype=torch.int64, device="cuda")
    H = 768
    a, ka = args(5120, H, seed_t, 3)
    b, kb = args(9472, H, seed_t, 4)
    nwg = L.lib.vk_ln_bwd_partial_rows(5120) + L.lib.vk_ln_bwd_partial_rows(9472)
    stamps = torch.zeros(nwg * 4, dtype=torch.int64, device="cuda")
    big_a = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    big_b = torch.empty_like(big_a)
    variants = [(0, 0)] + ([tuple(int(x) for x in v.split(":")) for v in sys.argv[1:]])      # "ticks:mod" pairs: stagger experiments
    for stag, cold in [(v, c) for v in variants for c in (False, True)]:
        L.lib.vk_ln_set_stagger(stag[0], stag[1])
        print("stagger %d ticks x (wg %% %d)" % stag)
        ts = []
        for it in range(8):
            if cold:
                big_b.copy_(big_a)
            stamps.zero_()
            L.lib.vk_ln_set_stamp_buffer(stamps.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(L.lib.vk_ln_bwd_pair(ctypes.byref(a), ctypes.byref(b), L.stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        L.lib.vk_ln_set_stamp_buffer(None)
        st = stamps.view(nwg, 4).cpu().double() / 100.0
        t0 = st[:, 0].min()
        q = lambda x: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % tuple(float(torch.quantile(x, v)) for v in (0.0, 0.1, 0.5, 0.9, 1.0))
        print("%s: launch %.1f us (HIP events), %d workgroups" % ("cold" if cold else "hot", sorted(ts)[len(ts) // 2], nwg))
        print("   workgroup start after the first one   ", q(st[:, 0] - t0))
        print("   entry -> first row's operands arrived ", q(st[:, 1] - st[:, 0]))
        print("   first row -> last row stored          ", q(st[:, 2] - st[:, 1]))
        print("   record reduction + write              ", q(st[:, 3] - st[:, 2]))
        print("   workgroup lifetime                    ", q(st[:, 3] - st[:, 0]))
        print("   last workgroup ends at %.1f us after the first one started" % float(st[:, 3].max() - t0))
        wg = torch.arange(nwg)
        first = st[:, 1] - st[:, 0]
        print("   entry -> first operands, median by blockIdx %% 8 (XCD):", " ".join("%.1f" % float(first[wg % 8 == x].median()) for x in range(8)))
        nb0 = L.lib.vk_ln_bwd_partial_rows(5120)
        print("   ... by job: text %.1f vision %.1f;  by dispatch order (eighths of the grid):" % (float(first[:nb0].median()), float(first[nb0:].median())),
              " ".join("%.1f" % float(first[i * nwg // 8:(i + 1) * nwg // 8].median()) for i in range(8)))
        slow = first > 9.0
        print("   %d workgroups waited > 9 us: blockIdx %% 8 histogram %s, blockIdx // 114 histogram %s" % (int(slow.sum()), torch.bincount(wg[slow] % 8, minlength=8).tolist(),
              torch.bincount(wg[slow] // 114, minlength=8).tolist()))


if __name__ == "__main__":
    main()
