"""How long does the compute stream wait for the weight-gradient side stream at the end of the backward list?
(exposed tail = time between the last compute-stream kernel of the backward and the JOIN completing)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd.config import BertConfig
from volta_amd.modeling import BertForVLPreTraining
from volta_amd.optimization import AdamW, clip_grad_norm_
from volta_amd import data, _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "ctrl_vilbert_base"
cfg = BertConfig.from_json_file(os.path.join(os.path.dirname(__file__), "..", "config", name + ".json"))
model = BertForVLPreTraining(cfg).cuda()
batch = data.synthetic_batch(cfg, 256, 20, 36, seed=0, device="cuda")
opt = AdamW(model.parameters(), lr=1e-4)
args = data.model_args(batch)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
probe = {"on": False}


def step():
    if probe["on"]:
        ev[0].record()
    lm, img, nsp = model(*args)
    loss = lm + img + nsp
    if probe["on"]:
        ev[1].record()
    loss.backward()
    if probe["on"]:
        ev[4].record()
    clip_grad_norm_(model.parameters(), 5.0, defer_to_optimizer=True)
    opt.step()
    opt.zero_grad()


for _ in range(5):
    step()
eng = model._last[0]
assert eng.bwd.ops[-1][0] == L.OP_JOIN
orig = eng.bwd.run


def run(start=0, end=None):
    n = len(eng.bwd.ops)
    if not probe["on"] or start != 0 or end is not None:
        return orig(start, end)
    orig(0, n - 1)
    ev[2].record()
    orig(n - 1, n)
    ev[3].record()


eng.bwd.run = run
probe["on"] = True
for _ in range(5):
    step()
    torch.cuda.synchronize()
    print("fwd %.2f ms  bwd(compute stream) %.2f ms  wait for side stream %.2f ms  total bwd %.2f ms" % (
        ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), ev[1].elapsed_time(ev[4])), flush=True)
