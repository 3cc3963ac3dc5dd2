"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/volta_hip.h declares, the ctypes structs match the header's layout, and the host-side mirrors of the
reference API (BertConfig, module tree, schedule, bucket planner) behave like the reference.  No GPU needed."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "volta_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from volta_amd import _lib
    declared = header_functions()
    assert len(declared) >= 35
    missing = [f for f in declared if not hasattr(_lib.lib, f)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == set(declared), (set(declared) ^ set(_lib.EXPORTS))
    assert _lib.lib.vk_version() >= 1 and _lib.lib.vk_device_arch() == b"gfx950"


def test_struct_layouts_match_the_header():
    """sizeof() of every ctypes mirror against a C translation unit compiled from the header itself."""
    import subprocess, tempfile
    from volta_amd import _lib as L
    pairs = {"vk_dropout": L.Dropout, "vk_gemm_problem": L.GemmProblem, "vk_drop_rows": L.DropRows, "vk_ln_args": L.LnArgs,
             "vk_ln_bwd_args": L.LnBwdArgs, "vk_attn_args": L.AttnArgs, "vk_attn_bwd_args": L.AttnBwdArgs, "vk_embed_args": L.EmbedArgs,
             "vk_embed_bwd_args": L.EmbedBwdArgs, "vk_xent_args": L.XentArgs, "vk_kl_args": L.KlArgs, "vk_vis_loss_args": L.VisLossArgs, "vk_adamw_args": L.AdamwArgs,
             "vk_generic_args": L.GenericArgs, "vk_op": L.Op, "vk_concap_args": L.ConcapArgs, "vk_concap_record": L.ConcapRecord, "vk_tail_job": L.TailJob, "vk_gemm_fp8_problem": L.GemmFp8Problem}
    src = '#include <stdio.h>\n#include "volta_hip.h"\nint main(void){' + "".join(
        'printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in pairs) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    for name, cls in pairs.items():
        assert ctypes.sizeof(cls) == sizes[name], (name, ctypes.sizeof(cls), sizes[name])


def test_host_validation_errors_without_gpu():
    """Argument checks happen on the host before any launch and report through vk_last_error()."""
    from volta_amd import _lib as L
    bad = L.GemmProblem(None, None, None, None, None, None, None, None, 128, 128, 64, 60, 64, 128, 0, 0)   # lda not a multiple of 8
    arr = (L.GemmProblem * 1)(bad)
    assert L.lib.vk_gemm_grouped(L.NT, L.EPI_BF16, arr, 1, None) != 0
    assert b"multiples of 8" in L.lib.vk_last_error()
    assert L.lib.vk_gemm_grouped(L.NT, L.EPI_BF16, arr, 0, None) != 0
    assert L.lib.vk_ln_bwd_partial_rows(100) == 7 and L.lib.vk_rows32(33) == 2
    with pytest.raises(L.VoltaHipError):
        L.check(L.lib.vk_run_ops((L.Op * 1)(L.Op(99, 0, 0, 0, None, None, None)), 1, None))


def test_bertconfig_mirror():
    from volta_amd.config import BertConfig
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "config", "ctrl_vilbert_base.json"))
    assert cfg.hidden_size == 768 and cfg.tv_attn_sublayers == [12, 16, 20, 24, 28, 32]
    assert cfg.objective == 0 and cfg.image_head_ln is True and cfg.model == "bert" and cfg.fixed_layers == []   # defaults survive
    assert BertConfig(30522).v_pooler_size == 1024 and BertConfig(30522, hidden_size=512).hidden_size == 512
    assert json.loads(cfg.to_json_string())["vocab_size"] == 30522
    with pytest.raises(ValueError):
        BertConfig(3.5)
    d = BertConfig.from_dict({"vocab_size": 10, "hidden_size": 32})
    assert d.hidden_size == 32 and d.num_attention_heads == 12


def test_model_is_a_drop_in_parameter_container_and_refuses_cpu_execution():
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    z = np.load(os.path.join(ROOT, "tests", "golden", "tiny_lxmert.npz"))
    cfg = BertConfig.from_dict(json.loads(str(z["cfg_json"])))
    model = BertForVLPreTraining(cfg)
    assert set(model.state_dict().keys()) == set(str(k) for k in z["ref_keys"])
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    model.load_state_dict(sd, strict=True)
    assert model.cls.predictions.decoder.weight is model.bert.embeddings.word_embeddings.weight           # tied
    shared = model.bert.encoder.layer[4].attention_self
    assert shared.v_query is shared.query                                                                  # shared sub-layer aliases
    names = [n for n, _ in model.named_parameters()]
    assert any("LayerNorm.weight" in n for n in names) and "bert.embeddings.token_type_embeddings.weight" in names
    with pytest.raises(RuntimeError, match="MI355X"):
        model.materialize()
    with pytest.raises(AssertionError):       # the reference's wiring assertions (encoders.py:842-843)
        BertForVLPreTraining(BertConfig.from_dict(dict(json.loads(str(z["cfg_json"])), t_ff_sublayers=[1, 3, 9])))
    with pytest.raises(ValueError):           # head divisibility (encoders.py:172-176)
        BertForVLPreTraining(BertConfig.from_dict(dict(json.loads(str(z["cfg_json"])), num_attention_heads=5)))


def test_bucket_planner():
    from volta_amd.parallel import plan_buckets
    spans = {"emb": (0, 100), "l0": (128, 50), "l1": (256, 50), "head": (384, 30)}
    ready = {"head": 0, "l1": 1, "l0": 2, "emb": 3}
    b = plan_buckets(spans, ready, 4, cap_bytes=4 * 60)
    assert b == [(1, [(256, 306), (384, 414)]), (3, [(0, 100), (128, 178)])]
    covered = sorted(r for _, rs in b for r in rs)
    assert covered == sorted((o, o + n) for o, n in spans.values())
    assert plan_buckets(spans, ready, 4, cap_bytes=1)[0] == (0, [(384, 414)])


def test_schedule_helpers_match_oracle():
    from oracle import volta_ref as R
    from volta_amd.optimization import WarmupLinearSchedule
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = WarmupLinearSchedule(opt, warmup_steps=3, t_total=10)
    for step in range(12):
        assert abs(opt.param_groups[0]["lr"] - R.warmup_linear(step, 3, 10)) < 1e-12
        opt.step()
        sch.step()


def test_generic_attention_lds_need_is_known_before_the_first_launch():
    """Rows beyond the MFMA attention tiles run on the generic kernels, whose backward keeps two row images of both modalities in LDS: the
    planner asks vk_gated_attn_lds_bytes (host arithmetic) and refuses a plan the backward could not launch (492-512 rows built
    fine and failed at the first backward)."""
    import ctypes as C
    from volta_amd import _lib as L
    aa = L.AttnArgs()
    aa.B, aa.nh, aa.dh, aa.scale = 2, 12, 64, 0.125
    for i in range(2):
        for j in range(2):
            aa.gate[i][j] = 1
    aa.L[0], aa.L[1] = 20, 37
    assert L.lib.vk_gated_attn_lds_bytes(C.byref(aa), 1) == 0            # the MFMA kernels serve the pre-training shapes
    aa.L[0], aa.L[1] = 80, 301
    f, b = L.lib.vk_gated_attn_lds_bytes(C.byref(aa), 0), L.lib.vk_gated_attn_lds_bytes(C.byref(aa), 1)
    assert 0 < f < b <= 160 * 1024
    aa.L[0], aa.L[1] = 80, 420                                            # 500 keys: the forward fits, the backward does not
    assert L.lib.vk_gated_attn_lds_bytes(C.byref(aa), 0) <= 160 * 1024 < L.lib.vk_gated_attn_lds_bytes(C.byref(aa), 1)
