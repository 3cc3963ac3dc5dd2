"""How long does the platform take to WRITE an output of a GEMM's size?  (floor for a GEMM epilogue)"""
import torch
for mb, tag in ((23.6, "5120x2304 bf16"), (31.5, "5120x3072 bf16"), (7.9, "5120x768 bf16"), (58.2, "9472x3072 bf16"), (125.0, "5120x30522 f32 / 5")):
    n = int(mb * 1e6 / 2)
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    y = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for name, fn in (("fill", lambda: x.zero_()), ("copy", lambda: x.copy_(y))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%-22s %5.1f MB %s: %6.1f us  %6.2f TB/s written" % (tag, mb, name, us, mb / us), flush=True)
