"""The HIP engine on the FIVE REAL ctrl_* configurations (full depth, config/ctrl_*.json) against fixtures produced by the
real reference (tests/golden/ctrl_*.npz, oracle/make_golden.py: B=2, T=20, 36 regions / 100 for VL-BERT, eval mode, weights
from the seed generator): the three losses, pooled vectors, hidden-state slices, checksums, gradient slices and norms.
No oracle in between.  bf16 storage + fp32 accumulation through 24-36 sub-layers; tolerances are stated per quantity."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import volta_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CTRL = ["ctrl_vilbert_base", "ctrl_lxmert", "ctrl_uniter_base", "ctrl_visualbert_base", "ctrl_vl-bert_base"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("name", CTRL)
def test_ctrl_config_matches_reference_fixture(golden_dir, name):
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    cd = json.load(open(os.path.join(ROOT, "config", name + ".json")))
    rcfg = R.RefConfig(cd)
    sd = R.make_weights(rcfg, seed=3, std=0.03)
    Rn = 100 if "vl-bert" in name else 36
    batch = R.synthetic_batch(rcfg, B=2, T=20, R=Rn, seed=7)
    model = BertForVLPreTraining(BertConfig.from_json_file(os.path.join(ROOT, "config", name + ".json")))
    assert list(model.state_dict().keys()) == [str(k) for k in z["ref_keys"]]
    assert sum(p.numel() for p in model.parameters()) == int(z["n_params"][0])
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    cb = {k: v.cuda() for k, v in batch.items()}
    lm, img, nsp = model(cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
                         cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])
    (lm + img + nsp).sum().backward()
    torch.cuda.synchronize()
    eng = model._last[0]
    B, T, H = 2, 20, rcfg.hidden_size
    Rv = batch["image_feat"].shape[1]
    seq_t = eng.taps["seq_t"].float().cpu().numpy().reshape(B, T, H)
    seq_v = eng.taps["seq_v"].float().cpu().numpy().reshape(B, Rv, H)
    report = {}
    # ---- losses.  north_star's 1e-3 is asserted at the BASELINE batch (B = 256) and at B = 32 against the same reference
    # (tests/test_fullsize_golden_gpu.py).  Here B = 2: the MLM loss averages ~5 labelled rows and the region loss ~10 of bf16-noisy
    # logits (observed 1e-4 ... 1.1e-3 from config to config and with the summation order of the LayerNorm statistics), the ITM
    # loss two samples.
    for got, key, tol in ((lm, "loss_lm", 1.5e-3), (img, "loss_img", 1.5e-3), (nsp, "loss_nsp", 3e-2)):
        want = float(z["out::" + key])
        report[key] = abs(float(got) - want) / abs(want)
        assert report[key] <= tol, (key, float(got), want)
    # ---- hidden states after the last sub-layer and pooled vectors
    report["seq_t"] = rel(seq_t[:, :, :64], z["out::seq_t_slice"])
    report["seq_v"] = rel(seq_v[:, :8, :64], z["out::seq_v_slice"])
    assert report["seq_t"] <= 3e-2 and report["seq_v"] <= 3e-2, report
    for key in ("pooled_t", "pooled_v"):
        report[key] = rel(eng.taps[key].float().cpu().numpy(), z["out::" + key])
        assert report[key] <= 3e-2, report
    # whole-tensor checksums (sums of ~30 k elements of mixed sign: compare against the tensor's 1-norm scale)
    for got, key in ((seq_t, "seq_t_sum"), (seq_v, "seq_v_sum")):
        assert abs(float(got.astype(np.float64).sum()) - float(z["out::" + key][0])) <= 2e-2 * float(np.abs(got).sum()), key
    # ---- gradients: slices and per-tensor norms of the parameters the fixture recorded, global norm
    named = dict(model.named_parameters())
    for k in z.files:
        if k.startswith("out::gradslice::"):
            pname = k[len("out::gradslice::"):]
            g = named[pname].grad.float().cpu().numpy()
            gs = g.reshape(g.shape[0], -1)[:16, :64] if g.ndim > 1 else g[:64]
            gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
            want_n = float(z["out::gradnorm::" + pname][0])
            itm_only = pname in ("bert.t_pooler.dense.weight", "cls.bi_seq_relationship.weight")      # carry the B = 2 ITM noise
            report["gn::" + pname] = abs(gn - want_n) / want_n
            assert report["gn::" + pname] <= (0.15 if itm_only else 4e-2), (pname, gn, want_n)
            if not itm_only:
                report["gs::" + pname] = rel(gs, z[k])
                assert report["gs::" + pname] <= (0.12 if ".query." in pname else 6e-2), (pname, report["gs::" + pname])
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())))
    report["grad_norm"] = abs(total - float(z["out::grad_norm"][0])) / float(z["out::grad_norm"][0])
    assert report["grad_norm"] <= 3e-2, report
    print(name, {k: float("%.2e" % v) for k, v in report.items()})
