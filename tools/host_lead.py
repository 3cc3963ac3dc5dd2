"""How far ahead of the GPU is the host at the phase boundaries of a step?  (lead = sync wait after the call returns)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd.config import BertConfig
from volta_amd.modeling import BertForVLPreTraining
from volta_amd.optimization import AdamW, clip_grad_norm_
from volta_amd import data

cfg = BertConfig.from_json_file(os.path.join(os.path.dirname(__file__), "..", "config", "ctrl_vilbert_base.json"))
model = BertForVLPreTraining(cfg).cuda()
batch = data.synthetic_batch(cfg, 256, 20, 36, seed=0, device="cuda")
opt = AdamW(model.parameters(), lr=1e-4)
args = data.model_args(batch)

def step(measure=False):
    out = {}
    t0 = time.perf_counter()
    lm, img, nsp = model(*args)
    loss = lm + img + nsp
    t1 = time.perf_counter()
    if measure == "fwd":
        torch.cuda.synchronize(); out["fwd_lead_ms"] = (time.perf_counter() - t1) * 1e3
    loss.backward()
    t2 = time.perf_counter()
    if measure == "bwd":
        torch.cuda.synchronize(); out["bwd_lead_ms"] = (time.perf_counter() - t2) * 1e3
    clip_grad_norm_(model.parameters(), 5.0, defer_to_optimizer=True)
    opt.step()
    opt.zero_grad()
    t3 = time.perf_counter()
    out.update(host_fwd_ms=(t1 - t0) * 1e3, host_bwd_ms=(t2 - t1) * 1e3, host_opt_ms=(t3 - t2) * 1e3)
    return out

for _ in range(5):
    step()
torch.cuda.synchronize()
for m in ("fwd", "bwd", "fwd", "bwd"):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    print(m, step(m), flush=True)
    torch.cuda.synchronize()
