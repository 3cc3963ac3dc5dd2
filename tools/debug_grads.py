import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import volta_ref as R
from test_engine_gpu import build, rel_err

name = sys.argv[1] if len(sys.argv) > 1 else "vilbert"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
train = len(sys.argv) > 3 and sys.argv[3] == "train"
which = sys.argv[4] if len(sys.argv) > 4 else "all"
model, rcfg, sd = build(name)
batch = R.synthetic_batch(rcfg, B, 20, 36, seed=7, pad=True)
seed = 0xABCDEF12345
model.train(train); model.set_dropout_seed(seed)
cb = {k: v.cuda() for k, v in batch.items()}
lm, img, nsp = model(cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
                     cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])
sel = lambda a, b, c: {"all": a + b + c, "lm": a, "img": b, "nsp": c}[which]
sel(lm, img, nsp).sum().backward()
torch.cuda.synchronize()
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k not in R.param_aliases(rcfg)}
full = dict(leaves)
for a, t in R.param_aliases(rcfg).items():
    full[a] = leaves[t]
olm, oimg, onsp = R.forward_from_batch(full, rcfg, batch, train=train, philox_seed=seed if train else None)
sel(olm, oimg, onsp).sum().backward()
print("losses", float(lm), float(olm), float(img), float(oimg), float(nsp), float(onsp))
named = dict(model.named_parameters())
rows = []
for k, leaf in leaves.items():
    if leaf.grad is None: continue
    g = named[k].grad.float().cpu(); r = leaf.grad
    cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-20))
    proj = float((g * r).sum() / ((r * r).sum() + 1e-30))
    rows.append((rel_err(g, r), cos, float(r.norm()), float(g.norm()), k, proj))
for e, c, rn, gn, k, pj in rows:
    if rn < 1e-6: continue
    print("%-66s rel %.4f cos %.5f |ref| %.4g proj %.4f" % (k, e, c, rn, pj))
taps = {}
with torch.no_grad():
    R.forward_from_batch(full, rcfg, batch, train=train, philox_seed=seed if train else None, taps=taps)
eng = model._last[0]
for k in ("seq_t", "seq_v", "pooled_t", "pooled_v"):
    got = eng.taps[k].float().cpu().view(taps[k].shape)
    print("TAP", k, "rel", rel_err(got, taps[k]), "absmax", float((got - taps[k]).abs().max()), "refmax", float(taps[k].abs().max()))
itm = eng.bufs["itm_logits"][:, :2].float().cpu()
print("ITM logits got", itm[:6].tolist())
print("ITM logits ref", taps["itm"][:6].tolist())
print("ITM rel", rel_err(itm, taps["itm"]))
pooled = eng.bufs["pooled"].float().cpu()
pref = taps["pooled_t"] * taps["pooled_v"]
print("pooled rel", rel_err(pooled, pref), float(pref.abs().max()))
