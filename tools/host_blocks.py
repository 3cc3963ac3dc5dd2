"""Where does the host wait in a steady-state loop (no synchronisation in the loop)?  Wall time of every host call of the step, averaged."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd.config import BertConfig
from volta_amd.modeling import BertForVLPreTraining
from volta_amd.optimization import AdamW, WarmupLinearSchedule, clip_grad_norm_
from volta_amd import data

if "--serial" in sys.argv:
    from volta_amd import _lib as L
    L.lib.vk_side_enable(0)
OVERLAP = "--no-overlap" not in sys.argv
cfg = BertConfig.from_json_file(os.path.join(os.path.dirname(__file__), "..", "config", "ctrl_vilbert_base.json"))
model = BertForVLPreTraining(cfg).cuda().train()
batch = data.synthetic_batch(cfg, 256, 20, 36, seed=0, device="cuda")
opt = AdamW(model.parameters(), lr=1e-4, overlap_with_forward=OVERLAP)
sched = WarmupLinearSchedule(opt, 100, 100000)
args = data.model_args(batch)
names = ["forward", "loss sum", "backward", "clip", "opt.step", "sched.step", "zero_grad"]
acc = [0.0] * len(names)
N = 60
for it in range(N + 10):
    t = [time.perf_counter()]
    lm, img, nsp = model(*args); t.append(time.perf_counter())
    loss = lm + img + nsp; t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    clip_grad_norm_(model.parameters(), 5.0, defer_to_optimizer=True); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    sched.step(); t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    if it >= 10:
        for i in range(len(names)):
            acc[i] += t[i + 1] - t[i]
torch.cuda.synchronize()
for n, a in zip(names, acc):
    print("%-12s %.3f ms" % (n, a / N * 1e3))
print("sum          %.3f ms" % (sum(acc) / N * 1e3))
