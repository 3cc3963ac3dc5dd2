"""fp8 (OCP e4m3) projection path: the row quantiser against torch's float8_e4m3fn conversion, the GEMM against an fp32 matmul of the
DE-QUANTISED operands (the kernel de-quantises exactly: tolerance = fp32 accumulation order + the bf16 output rounding), and the
end-to-end error of quantise -> GEMM against the un-quantised fp32 product (the stated fp8 tolerance).  GPU only."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    from volta_amd import _lib as L, ops
    return L, ops


def quant_rows(L, ops, x, src_is_f32=False):
    M, K = x.shape
    Kp = (K + 127) // 128 * 128
    q = torch.zeros(M, Kp, dtype=torch.uint8, device="cuda")
    s = torch.empty(M, dtype=torch.float32, device="cuda")
    L.check(L.lib.vk_quant_rows_fp8(L.ptr(x), int(src_is_f32), x.stride(0), L.ptr(q), Kp, L.ptr(s), M, K, None, ops.stream_ptr()))
    return q, s


def fp8_bytes(t):
    """torch's own e4m3fn conversion (done on the host: independent of the GPU's conversion instructions)"""
    return t.detach().float().cpu().to(torch.float8_e4m3fn).view(torch.uint8).to(t.device)


def dequant(q, s, K):
    return q[:, :K].contiguous().cpu().view(torch.float8_e4m3fn).float().to(q.device) * s[:, None]


@pytest.mark.parametrize("M,K", [(5, 768), (300, 3072), (64, 8), (1000, 4096)])
@pytest.mark.parametrize("f32", [False, True])
def test_quant_rows_matches_torch_float8(M, K, f32):
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x = torch.randn(M, K, device="cuda", generator=g) * torch.logspace(-3, 2, M, device="cuda")[:, None]
    x[0] = 0.0                                           # an all-zero row keeps scale 1
    if not f32:
        x = x.bfloat16()
    q, s = quant_rows(L, ops, x, f32)
    torch.cuda.synchronize()
    xf = x.float()
    amax = xf.abs().amax(1)
    want_s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(s, want_s, rtol=1e-6)
    want_q = fp8_bytes((xf * (1.0 / s)[:, None]).clamp(-448, 448))
    got_q = q[:, :K]
    # the reciprocal multiply may sit one ulp of fp32 away from torch's: allow the rare element that lands on a rounding boundary
    diff = (got_q != want_q)
    assert float(diff.float().mean()) < 2e-4, float(diff.float().mean())
    assert float((dequant(q, s, K) - xf).abs().max() / xf.abs().max()) < 0.07          # e4m3: 3 mantissa bits
    assert int((q[:, K:] != 0).sum()) == 0


def test_cast_bf16_fp8_static_scale():
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn(4096 * 8, device="cuda", generator=g) * 5).bfloat16()
    x[:4] = torch.tensor([1000.0, -1000.0, 0.0, 56.0], device="cuda").bfloat16()       # saturation at +-448 / mul
    q = torch.empty(x.numel(), dtype=torch.uint8, device="cuda")
    L.check(L.lib.vk_cast_bf16_fp8(L.ptr(x), L.ptr(q), x.numel(), 8.0, ops.stream_ptr()))
    torch.cuda.synchronize()
    want = fp8_bytes((x.float() * 8.0).clamp(-448, 448))
    assert torch.equal(q, want)


SHAPES = [(128, 128, 128), (256, 384, 256), (300, 200, 768), (1000, 768, 768), (77, 1601, 256), (5120, 2304, 768), (2048, 768, 3072), (64, 8, 1024)]


@pytest.mark.parametrize("geometry", [0, 128, 256])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_fp8_matches_dequantised_fp32(M, N, K, geometry):
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    x = (torch.randn(M, K, device="cuda", generator=g) * 1.5).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    qa, sa = quant_rows(L, ops, x)
    qb, sb = quant_rows(L, ops, w)
    out = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
    p = L.GemmFp8Problem(L.GemmProblem(L.ptr(qa), L.ptr(qb), L.ptr(out), None, L.ptr(bias), None, None, None, M, N, K, qa.stride(0), qb.stride(0), N, 0, 0),
                         L.ptr(sa), L.ptr(sb))
    if N % 4:
        out = torch.full((M, (N + 3) // 4 * 4), 7.0, device="cuda", dtype=torch.bfloat16)
        p.p.C, p.p.ldc = L.ptr(out), out.stride(0)
    arr = (L.GemmFp8Problem * 1)(p)
    L.check(L.lib.vk_gemm_fp8_grouped(L.EPI_BF16, arr, 1, geometry, ops.stream_ptr()))
    torch.cuda.synchronize()
    ref = dequant(qa, sa, K) @ dequant(qb, sb, K).t() + bias
    got = out[:, :N].float()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, err                      # bf16 output rounding (2^-8 relative) on top of the fp32 accumulation order
    if out.shape[1] > N:
        assert float((out[:, N:].float() - 7.0).abs().max()) == 0.0        # nothing written beyond N
    # the fp8 error itself, against the un-quantised product: ~ 2 x 2^-4 / sqrt(3) per product, averaged over K terms
    full = x.float() @ w.float().t() + bias
    rel = float((got - full).norm() / full.norm())
    assert rel < 5e-2, rel


def test_gemm_fp8_gelu_group_and_dyn_rows():
    """Two problems in one launch, GELU epilogue with derivative output, a device-side row count."""
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(11)
    K, N = 768, 3072
    probs, refs, outs = [], [], []
    n_dyn = torch.tensor([301], dtype=torch.int32, device="cuda")
    for M, dyn in ((512, None), (700, n_dyn)):
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.04).bfloat16()
        bias = torch.randn(N, device="cuda", generator=g) * 0.1
        qa, sa = quant_rows(L, ops, x)
        qb, sb = quant_rows(L, ops, w)
        h = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        gp = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        probs.append(L.GemmFp8Problem(L.GemmProblem(L.ptr(qa), L.ptr(qb), L.ptr(h), L.ptr(gp), L.ptr(bias), None, None, L.ptr(dyn), M, N, K, qa.stride(0), qb.stride(0), N, 0, 0),
                                      L.ptr(sa), L.ptr(sb)))
        u = (dequant(qa, sa, K) @ dequant(qb, sb, K).t() + bias).double()
        rows = M if dyn is None else int(dyn)
        cdf = 0.5 * (1 + torch.erf(u / 2 ** 0.5))
        refs.append(((u * cdf)[:rows].float(), (cdf + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5)[:rows].float(), rows))
        outs.append((h, gp, qa, qb, sa, sb, bias))
    arr = (L.GemmFp8Problem * 2)(*probs)
    L.check(L.lib.vk_gemm_fp8_grouped(L.EPI_GELU, arr, 2, 0, ops.stream_ptr()))
    torch.cuda.synchronize()
    for (h, gp, *_), (rh, rg, rows) in zip(outs, refs):
        assert float((h[:rows].float() - rh).abs().max()) < 3e-2 and float((gp[:rows].float() - rg).abs().max()) < 2e-2
        assert float(h[rows:].abs().max()) == 0.0 if rows < h.shape[0] else True      # rows past the device-side count untouched (256-row tile granularity aside)


# ---------------------------------------------------------------------------------------------------- the engine with fp8 projections
# Gates = at most 2 x the values observed on MI355X (round 3, gpurun_out -> profiles/r03_fp8_observed.txt):
#   reduced-depth models   hidden 0.090 / 0.076 / 0.067, MLM loss 3.2e-3 / 3.1e-3 / 2.8e-3, region loss 7.8e-3 / 2.3e-3 / 1.1e-3,
#                          smallest gradient cosine 0.964 / 0.989 / 0.987   (vilbert / uniter / vlbert)
#   ctrl_vl-bert_base, 100 regions, real-reference fixture   MLM 1.0e-3, region 2.3e-3, ITM 2.5e-2 (B = 2), hidden states 0.107
FP8_HIDDEN_TOL = 0.15       # relative L2 of hidden states after 4-8 sub-layers: every e4m3 product carries ~5 % relative noise (3 mantissa bits on both
                            # operands, independent per term), which LayerNorm and the residual stream pass on; observed <= 0.090
FP8_LOSS_TOL = 6.5e-3       # relative, MLM loss; observed <= 3.2e-3
FP8_IMG_LOSS_TOL = 1.6e-2   # relative, region loss; observed <= 7.8e-3


@pytest.mark.parametrize("name", ["vilbert", "uniter", "vlbert"])
def test_engine_fp8_forward_backward_against_oracle(name):
    """Reduced-depth models at true width: fp8 forward projections + bf16 backward (straight-through) against the fp32 oracle."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import build, rel_err
    from oracle import volta_ref as R
    model, rcfg, sd = build(name)
    model.set_projection_dtype("fp8")
    model.eval()
    batch = R.synthetic_batch(rcfg, 4, 20, 36, seed=7, pad=True)
    cb = {k: v.cuda() for k, v in batch.items()}
    lm, img, nsp = model(cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
                         cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])
    (lm + img).sum().backward()
    torch.cuda.synchronize()
    eng = model._last[0]
    assert eng.fp8 and any(op[0] == 18 for op in eng.fwd.ops)          # OP_GEMM_FP8 launches are in the forward list
    aliases = R.param_aliases(rcfg)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k not in aliases}
    full = dict(leaves)
    for a, t in aliases.items():
        full[a] = leaves[t]
    taps = {}
    olm, oimg, onsp = R.forward_from_batch(full, rcfg, batch, taps=taps)
    (olm + oimg).sum().backward()
    worst = 0.0
    for key, ref in taps.items():
        if key in eng.taps and ref is not None and ref.dim() >= 2:
            worst = max(worst, rel_err(eng.taps[key].float().cpu().view(ref.shape), ref.detach()))
    report = {"hidden": worst, "lm": abs(float(lm) - float(olm)) / abs(float(olm)), "img": abs(float(img) - float(oimg)) / abs(float(oimg))}
    named = dict(model.named_parameters())
    cos = []
    for k, leaf in leaves.items():
        if leaf.grad is None or float(leaf.grad.norm()) < 1e-6:
            continue
        gg = named[k].grad.float().cpu()
        assert torch.isfinite(gg).all(), k
        cos.append(float((gg * leaf.grad).sum() / (gg.norm() * leaf.grad.norm())))
    report["min_grad_cos"] = min(cos)
    print(name, {k: float("%.3g" % v) for k, v in report.items()})
    assert report["hidden"] <= FP8_HIDDEN_TOL and report["lm"] <= FP8_LOSS_TOL and report["img"] <= FP8_IMG_LOSS_TOL, report
    assert report["min_grad_cos"] >= 0.95, report          # straight-through backward over the noisy forward activations; observed >= 0.964 (1 - cos <= 1.4 x observed)


def test_ctrl_vlbert_100_regions_fp8_against_reference_fixture(golden_dir):
    """BASELINE.json configs[4]: ctrl_vl-bert_base, 100 regions, fp8 projections, against the REAL reference's fixture (B = 2)."""
    import json, os
    import numpy as np
    from oracle import volta_ref as R
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "ctrl_vl-bert_base"
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    rcfg = R.RefConfig(json.load(open(os.path.join(ROOT, "config", name + ".json"))))
    sd = R.make_weights(rcfg, seed=3, std=0.03)
    batch = R.synthetic_batch(rcfg, B=2, T=20, R=100, seed=7)
    model = BertForVLPreTraining(BertConfig.from_json_file(os.path.join(ROOT, "config", name + ".json")))
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    model.set_projection_dtype("fp8")
    cb = {k: v.cuda() for k, v in batch.items()}
    with torch.no_grad():
        lm, img, nsp = model(cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
                             cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])
    torch.cuda.synchronize()
    eng = model._last[0]
    report = {}
    for got, key in ((lm, "loss_lm"), (img, "loss_img"), (nsp, "loss_nsp")):
        want = float(z["out::" + key][0])
        report[key] = abs(float(got) - want) / abs(want)
    seq_t = eng.taps["seq_t"].float().cpu().numpy().reshape(2, 20, 768)
    ref = z["out::seq_t_slice"]
    report["seq_t"] = float(np.linalg.norm(seq_t[:, :, :64] - ref) / np.linalg.norm(ref))
    print(name, "fp8", {k: float("%.3g" % v) for k, v in report.items()})
    assert report["loss_lm"] <= 2.1e-3 and report["loss_img"] <= 4.5e-3 and report["loss_nsp"] <= 5.1e-2, report       # observed 1.0e-3 / 2.3e-3 / 2.5e-2 (ITM: 2 samples)
    assert report["seq_t"] <= 0.15, report          # 24 sub-layers of fp8 projections; observed 0.107


def test_ln_fwd_writes_row_quantised_copy():
    """vk_ln_fwd with y8 / y8_scale: the LayerNorm output leaves the kernel a second time as e4m3 rows (the next projection's A operand)."""
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(5)
    M, H = 333, 768
    d, x = (torch.randn(M, H, device="cuda", generator=g).bfloat16() for _ in range(2))
    gamma, beta = 1 + 0.1 * torch.randn(H, device="cuda", generator=g), 0.1 * torch.randn(H, device="cuda", generator=g)
    y, z = (torch.empty(M, H, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    y8 = torch.zeros(M, H, dtype=torch.uint8, device="cuda")
    s8 = torch.zeros(M, device="cuda")
    drop = L.dropout_cfg(None, 0, 0.0)
    a = L.LnArgs(L.ptr(d), L.ptr(x), None, L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(z), L.ptr(mean), L.ptr(rstd), None, M, H, M, 0, 1.0, drop, ops._segs(drop, None))
    a.y8, a.y8_scale, a.ld8 = y8.data_ptr(), s8.data_ptr(), H
    L.check(L.lib.vk_ln_fwd(C.byref(a), ops.stream_ptr()))
    torch.cuda.synchronize()
    yf = y.float()
    amax = yf.abs().amax(1)
    assert torch.allclose(s8, amax / 448.0, rtol=1e-2)                 # the scale is taken before the bf16 rounding of y
    deq = dequant(y8, s8, H)
    assert float((deq - yf).abs().max() / yf.abs().max()) < 0.07
    assert float((deq - yf).norm() / yf.norm()) < 0.04


def test_gelu_epilogue_writes_static_scale_fp8_copy():
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(9)
    M, N, K, mul = 700, 3072, 768, 8.0
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.04).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    qa, sa = quant_rows(L, ops, x)
    qb, sb = quant_rows(L, ops, w)
    h, gp = (torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    h8 = torch.zeros(M, N, dtype=torch.uint8, device="cuda")
    p = L.GemmFp8Problem(L.GemmProblem(L.ptr(qa), L.ptr(qb), L.ptr(h), L.ptr(gp), L.ptr(bias), None, None, None, M, N, K, qa.stride(0), qb.stride(0), N, 0, 0),
                         L.ptr(sa), L.ptr(sb), h8.data_ptr(), mul, N)
    for geometry in (128, 256):
        h8.zero_()
        L.check(L.lib.vk_gemm_fp8_grouped(L.EPI_GELU, (L.GemmFp8Problem * 1)(p), 1, geometry, ops.stream_ptr()))
        torch.cuda.synchronize()
        deq = h8.cpu().view(torch.float8_e4m3fn).float().cuda() / mul
        hf = h.float()
        assert float(((deq - hf).abs() - (0.07 * hf.abs() + 2.0 ** -9 / mul + 4e-3)).clamp(min=0).max()) == 0.0     # e4m3 rounding (+ h's own bf16 rounding)
        assert float((deq - hf).norm() / hf.norm()) < 0.04
