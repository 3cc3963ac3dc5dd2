"""Diagnostic: is a 4-step training run reproducible run to run, and does AdamW(overlap_with_forward=True) change it?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_optim_gpu import _build, _args
from volta_amd.optimization import AdamW, clip_grad_norm_

def run(overlap, lr, steps=4, sync=False):
    model, rcfg, sd = _build("vilbert")
    model.train(); model.set_dropout_seed(21); model.materialize()
    opt = AdamW([{"params": [p], "weight_decay": 0.01} for p in model.parameters()], lr=lr, overlap_with_forward=overlap, overlap_ranges=5)
    args = _args(rcfg)
    losses, snaps = [], []
    for _ in range(steps):
        out = model(*args)
        sum(out).sum().backward()
        clip_grad_norm_(model.parameters(), 5.0, defer_to_optimizer=True)
        snaps.append(model._arena.grad.clone())
        opt.step(); opt.zero_grad()
        if sync:
            opt.synchronize(); torch.cuda.synchronize()
        losses.append([x.detach().clone() for x in out])
    opt.synchronize(); torch.cuda.synchronize()
    return [[float(x) for x in l] for l in losses], snaps, model._arena.master.clone()

for lr in (5e-3, 1e-4):
    a = run(False, lr); b = run(False, lr); c = run(True, lr); d = run(True, lr, sync=True)
    for tag, r in (("serial#2", b), ("overlap", c), ("overlap+sync", d)):
        print("lr", lr, tag)
        for s in range(4):
            gd = float((a[1][s] - r[1][s]).abs().max()); gn = float(a[1][s].abs().max())
            print("  step", s, "loss", a[0][s], r[0][s], "grad maxdiff %.3e (max %.3e)" % (gd, gn))
        print("  master maxdiff %.3e" % float((a[2] - r[2]).abs().max()))
