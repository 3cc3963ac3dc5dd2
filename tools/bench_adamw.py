"""AdamW update bandwidth: the full-width launch (vk_adamw_step) against the resident-workgroup form (vk_adamw_step_on) at several
workgroup counts, alone on the GPU.  usage: python tools/bench_adamw.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L

n = 64 * 1024 * 1024
p, g, m, v = (torch.rand(n, device="cuda") for _ in range(4))
sh = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
a = L.AdamwArgs()
a.p, a.g, a.m, a.v, a.shadow, a.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), n
a.cls_lr_mult[0], a.cls_wd[0] = 1.0, 0.01
a.lr, a.beta1, a.beta2, a.eps, a.step_mult, a.grad_scale = 1e-3, 0.9, 0.999, 1e-6, 1.0, 1.0
bytes_ = n * 30


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e-3


t = timed(lambda: L.check(L.lib.vk_adamw_step(C.byref(a), L.stream_ptr())))
print("full width: %.0f us, %.2f TB/s" % (t * 1e6, bytes_ / t / 1e12))
for ncus in (8, 16, 24, 32, 64, 128, 256):
    t = timed(lambda: L.check(L.lib.vk_adamw_step_on(C.byref(a), ncus, L.stream_ptr())))
    print("%3d resident workgroups: %.0f us, %.2f TB/s, %.1f GB/s per CU" % (ncus, t * 1e6, bytes_ / t / 1e12, bytes_ / t / 1e9 / ncus))
