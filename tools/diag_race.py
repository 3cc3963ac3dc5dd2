"""Diagnostic: the same forward + backward (same weights, batch and dropout seed) repeated back to back must give the same gradients
up to the summation order of the few atomically accumulated tensors.  Prints every parameter whose gradient moved by more than 1e-4
of its largest element."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_optim_gpu import _build, _args

name = sys.argv[1] if len(sys.argv) > 1 else "vilbert"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
model, rcfg, sd = _build(name)
model.train(); model.materialize()
args = _args(rcfg, B=B)
snaps, losses = [], []
for r in range(reps):
    model.set_dropout_seed(21)
    for p in model.parameters():
        p.grad = None
    out = model(*args)
    sum(out).sum().backward()
    snaps.append(model._arena.grad.clone())
    losses.append(torch.stack([x.detach().reshape(()) for x in out]))
torch.cuda.synchronize()
arena = model._arena
bad = 0
for r in range(1, reps):
    d = (snaps[r] - snaps[0]).abs()
    if float(d.max()) <= 1e-4 * float(snaps[0].abs().max()):
        continue
    bad += 1
    print("rep", r, "losses", losses[0].tolist(), losses[r].tolist(), "max diff %.3e" % float(d.max()))
    for n in arena.params:
        g0, g1 = arena.view(n, "grad"), None
        off = arena.offset[n]; numel = g0.numel()
        dd = d[off:off + numel]
        if float(dd.max()) > 1e-4 * max(float(g0.abs().max()), 1e-6):
            print("   %-60s maxdiff %.3e of %.3e, %d / %d elements" % (n, float(dd.max()), float(snaps[0][off:off + numel].abs().max()), int((dd > 1e-6).sum()), numel))
print(name, "reps", reps, "bad", bad)
