"""Downstream-task model: the oracle's restatement of BertForVLTasks against fixtures written by the REAL
reference model (tests/golden/tasks_*.npz, oracle/make_golden.py tasks) -- every head type, predictions and gradients -- and
the product model's parameter inventory against the reference's.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import volta_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    z = np.load(os.path.join(GOLD, "tasks_%s.npz" % name), allow_pickle=False)
    cd, task_cfg = json.loads(str(z["cfg_json"])), json.loads(str(z["task_cfg_json"]))
    return z, cd, task_cfg


@pytest.mark.parametrize("name", ["tiny_vilbert", "tiny_uniter", "tiny_vilbert_vqa", "tiny_vilbert_sum", "tiny_vilbert_text"])
def test_oracle_tasks_forward_backward_matches_reference(name):
    z, cd, task_cfg = _load(name)
    cfg = R.RefConfig(cd)
    ids = list(task_cfg)
    base = R.make_task_weights(cfg, task_cfg, ids, seed=13)
    batch = R.synthetic_batch(cfg, B=4, T=6, R=4, seed=17, pad=True)
    aliases = R.param_aliases(cfg)
    for t in ids:
        leaves = {k: v.clone().requires_grad_(True) for k, v in base.items() if k not in aliases}
        sd = dict(leaves)
        for a, tgt in aliases.items():
            sd[a] = leaves[tgt]
        pred = R.tasks_forward(sd, cfg, task_cfg, t, batch["input_ids"], batch["image_feat"].clone(), batch["image_loc"], batch["segment_ids"],
                               batch["input_mask"], batch["image_mask"])
        np.testing.assert_allclose(pred.detach().numpy(), z["pred::" + t], rtol=0, atol=2e-5 + 1e-5 * np.abs(z["pred::" + t]).max())
        probe = torch.randn(pred.shape, generator=torch.Generator().manual_seed(sum(map(ord, t))))
        (pred * probe).sum().backward()
        for k in z.files:
            if k.startswith("grad::%s::" % t):
                g = leaves[k.split("::", 2)[2]].grad.numpy()
                np.testing.assert_allclose(g, z[k], rtol=0, atol=2e-6 + 2e-5 * np.abs(z[k]).max())


def test_product_task_model_has_the_reference_parameters():
    pytest.importorskip("volta_amd._lib", reason="libvolta_hip.so not built")
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLTasks
    z, cd, task_cfg = _load("tiny_vilbert")
    model = BertForVLTasks(BertConfig.from_dict(cd), task_cfg, list(task_cfg))
    assert list(model.state_dict().keys()) == [str(k) for k in z["ref_keys"]]
    cfg = R.RefConfig(cd)
    sd = R.make_task_weights(cfg, task_cfg, list(task_cfg), seed=13)
    model.load_state_dict(sd, strict=True)
    with pytest.raises(ValueError):
        BertForVLTasks(BertConfig.from_dict(cd), {"X": {"type": "nope"}}, ["X"])
