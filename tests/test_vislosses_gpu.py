"""csrc/vislosses.hip through the C ABI against plain torch (fp32 autograd) and the oracle: the mse / huber / xent(+confidence) / nce
row losses and their logit gradients on compacted labelled rows, the negatives' generator bit for bit, the pooled-vector fusions with
their backward through dropout and the poolers' ReLUs, VLBertTextPooler's row index and VL-BERT's per-region word index."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import volta_ref as R

pytestmark = pytest.mark.gpu


def _lib():
    from volta_amd import _lib as L
    return L


def _setup(B, Rn, V, seed):
    g = torch.Generator().manual_seed(seed)
    label = torch.where(torch.rand(B, Rn, generator=g) < 0.3, 1, -1)
    label[0, 1] = 1
    pos = torch.nonzero(label.view(-1) == 1).view(-1).int()
    n = pos.numel()
    Vp = -(-V // 64) * 64
    logits = torch.zeros(B * Rn, Vp)
    logits[:n, :V] = torch.randn(n, V, generator=g)
    return g, label, pos, n, Vp, logits


def _run(L, kind, logits, pos, n, V, Vp, weight, target=None, labels=None, conf=None, neg=None, gscale=0.7):
    dev = "cuda"
    rows = logits.shape[0]
    d = dict(logits=logits.to(dev), pos=torch.cat([pos, torch.zeros(rows - n, dtype=torch.int32)]).to(dev), count=torch.tensor([n], dtype=torch.int32, device=dev),
             lse=torch.zeros(rows, device=dev), aux=torch.zeros(rows, L.NCE_MAX_SAMPLES, device=dev), loss=torch.zeros(1, device=dev),
             dlog=torch.full((rows, Vp), 7.0, dtype=torch.bfloat16, device=dev), g=torch.tensor([gscale], device=dev))
    for k, v in (("target", target), ("labels", labels), ("conf", conf), ("neg", neg)):
        d[k] = None if v is None else v.to(dev).contiguous()
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    a = L.VisLossArgs(p(d["logits"]), p(d["target"]), p(d["labels"]), p(d["conf"]), p(d["pos"]), p(d["count"]), p(d["neg"]), p(d["lse"]), p(d["aux"]), p(d["loss"]),
                      weight, V, Vp, rows, kind, 0 if neg is None else neg.shape[-1])
    L.check(L.lib.vk_vis_loss_fwd(C.byref(a), L.stream_ptr()))
    L.check(L.lib.vk_vis_loss_bwd(C.byref(a), p(d["dlog"]), Vp, p(d["g"]), L.stream_ptr()))
    torch.cuda.synchronize()
    return float(d["loss"]) / max(n, 1), d["dlog"].float().cpu()


def _check(got_loss, got_d, want_loss, x, n, V, gscale=0.7):
    want_loss.backward()
    want_d = x.grad * gscale
    assert abs(got_loss - float(want_loss)) <= 2e-5 * abs(float(want_loss)) + 1e-6, (got_loss, float(want_loss))
    e = float((got_d[:n, :V] - want_d).norm() / want_d.norm())
    assert e <= 6e-3, e                                    # bf16 output rounding
    assert float(got_d[:n, V:].abs().max()) == 0.0 if got_d.shape[1] > V else True
    assert float((got_d[n:] - 7.0).abs().max()) == 0.0     # rows beyond the device-side count are not touched


@pytest.mark.parametrize("kind", ["mse", "huber"])
def test_regression_targets(kind):
    L = _lib()
    B, Rn, V = 5, 9, 2048
    g, label, pos, n, Vp, logits = _setup(B, Rn, V, 3)
    feat = torch.randn(B, Rn, V, generator=g) * (2.0 if kind == "huber" else 1.0)
    got = _run(L, L.VIS_MSE if kind == "mse" else L.VIS_HUBER, logits, pos, n, V, Vp, 1.7, target=feat.view(-1, V))
    x = logits[:n, :V].clone().requires_grad_(True)
    pred = torch.zeros(B * Rn, V).index_add(0, pos.long(), x).view(B, Rn, V)       # the labelled predictions at their grid positions
    want = (R.mse_2048 if kind == "mse" else R.huber_2048)(pred, 1.7, label, feat)
    _check(*got, want, x, n, V)


@pytest.mark.parametrize("V,with_conf", [(1600, True), (400, True), (1601, False)])
def test_hard_label_targets(V, with_conf):
    L = _lib()
    B, Rn = 6, 11
    g, label, pos, n, Vp, logits = _setup(B, Rn, V, 5)
    labels = torch.randint(0, V, (B, Rn), generator=g)
    conf = torch.rand(B, Rn, generator=g) if with_conf else None
    got = _run(L, L.VIS_XENT, logits, pos, n, V, Vp, 6.667, labels=labels.view(-1), conf=None if conf is None else conf.view(-1))
    x = logits[:n, :V].clone().requires_grad_(True)
    pred = torch.zeros(B * Rn, V).index_add(0, pos.long(), x).view(B, Rn, V)
    want = R.xent_hard(pred, 6.667, label, labels, conf)
    _check(*got, want, x, n, V)


def test_nce_target_and_negatives():
    L = _lib()
    B, Rn, V = 4, 7, 2048
    g, label, pos, n, Vp, logits = _setup(B, Rn, V, 9)
    logits = logits * 0.05
    feat = torch.randn(B, Rn, V, generator=g)
    seed, site = 0x5EEDF00D1234, 17
    seed_t = torch.tensor([seed], dtype=torch.int64, device="cuda")
    nneg = L.NCE_ACROSS + L.NCE_INSIDE
    neg = torch.zeros(B * Rn, nneg, dtype=torch.int32, device="cuda")
    L.check(L.lib.vk_nce_negatives(L.rng_cfg(seed_t.data_ptr(), site), B, Rn, C.c_void_p(neg.data_ptr()), L.stream_ptr()))
    torch.cuda.synchronize()
    want_idx = R.nce_negative_index(R.nce_draws(seed, site, B, Rn), B, Rn)
    assert torch.equal(neg.cpu().view(B, Rn, nneg).long(), want_idx)
    b = torch.arange(B)[:, None, None]
    assert bool((want_idx[..., :L.NCE_ACROSS] // Rn != b).all())                       # across: another image
    assert bool((want_idx[..., L.NCE_ACROSS:] // Rn == b).all()) and bool((want_idx[..., L.NCE_ACROSS:] % Rn != torch.arange(Rn)[None, :, None]).all())
    got = _run(L, L.VIS_NCE, logits, pos, n, V, Vp, 1.5, target=feat.view(-1, V), neg=neg.cpu())
    x = logits[:n, :V].clone().requires_grad_(True)
    pred = torch.zeros(B * Rn, V).index_add(0, pos.long(), x).view(B, Rn, V)
    want = R.nce_2048(pred, 1.5, label, feat, want_idx)
    _check(*got, want, x, n, V)


@pytest.mark.parametrize("mode", ["mul", "sum", "text"])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_pool_fusion(mode, p):
    L = _lib()
    B, P = 6, 1024
    g = torch.Generator().manual_seed(1)
    pt, pv = torch.relu(torch.randn(B, P, generator=g)).bfloat16(), torch.relu(torch.randn(B, P, generator=g)).bfloat16()
    dp = torch.randn(B, P, generator=g).bfloat16()
    seed, site = 991, 4
    seed_t = torch.tensor([seed], dtype=torch.int64, device="cuda")
    drop = L.dropout_cfg(seed_t.data_ptr(), site, p)
    m = {"mul": L.FUSE_MUL, "sum": L.FUSE_SUM, "text": L.FUSE_TEXT}[mode]
    dev = lambda t: t.cuda()
    a, b, out, d = dev(pt), dev(pv), torch.zeros(B, P, dtype=torch.bfloat16, device="cuda"), dev(dp)
    dyt, dyv = torch.zeros_like(out), torch.zeros_like(out)
    vp = lambda t: C.c_void_p(t.data_ptr())
    L.check(L.lib.vk_pool_fuse_fwd(vp(a), None if mode == "text" else vp(b), vp(out), B, P, m, drop, L.stream_ptr()))
    L.check(L.lib.vk_pool_fuse_bwd(vp(d), P, vp(a), None if mode == "text" else vp(b), vp(dyt), None if mode == "text" else vp(dyv), B, P, m, drop, L.stream_ptr()))
    torch.cuda.synchronize()
    keep = R.philox_keep_mask(seed, site, (B, P), p).float() / (1.0 - p) if p else torch.ones(B, P)
    x, y = pt.float().requires_grad_(True), pv.float().requires_grad_(True)
    fused = {"mul": x * y, "sum": x + y, "text": x}[mode] * keep
    assert float((out.float().cpu() - fused.detach()).abs().max()) <= 2e-2 * float(fused.abs().max())
    (fused * dp.float()).sum().backward()
    wt = x.grad * (pt.float() > 0)
    assert float((dyt.float().cpu() - wt).abs().max()) <= 2e-2 * float(wt.abs().max())
    if mode != "text":
        wv = y.grad * (pv.float() > 0)
        assert float((dyv.float().cpu() - wv).abs().max()) <= 2e-2 * float(wv.abs().max())


def test_text_end_rows_and_vlbert_word_ids():
    L = _lib()
    B, T = 7, 20
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(2, T + 1, (B,), generator=g)
    lens[0], lens[1] = T, 2
    ids = torch.randint(1, 3000, (B, T), generator=g) * (torch.arange(T)[None] < lens[:, None])
    rows, cnt = torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    idc = ids.cuda()
    L.check(L.lib.vk_text_end_rows(C.c_void_p(idc.data_ptr()), B, T, C.c_void_p(rows.data_ptr()), C.c_void_p(cnt.data_ptr()), L.stream_ptr()))
    torch.cuda.synchronize()
    assert int(cnt) == B and torch.equal(rows.cpu().long(), torch.arange(B) * T + (ids != 0).sum(1) - 2)
    M, K = 4 * 9, 9
    z = (torch.rand(M, generator=g) < 0.3).int()
    out = torch.zeros(M, dtype=torch.int64, device="cuda")
    zc = z.cuda()
    L.check(L.lib.vk_vlbert_obj_ids(C.c_void_p(zc.data_ptr()), C.c_void_p(out.data_ptr()), M, K, L.stream_ptr()))
    torch.cuda.synchronize()
    want = torch.where(torch.arange(M) % K == K - 1, 1, torch.where(z == 1, 2, 0))
    assert torch.equal(out.cpu(), want.long())


@pytest.mark.parametrize("B,T,K", [(7, 20, 9), (1, 5, 2), (1100, 38, 37), (3, 16, 101)])
def test_vlbert_position_ids(B, T, K):
    """vk_vlbert_positions against the oracle's statement of embeddings.py:278-292 (the stride-0 shift quirk included):
    ragged captions, one full-length caption, an all-pad caption, more samples than the work-group has threads."""
    L = _lib()
    g = torch.Generator().manual_seed(B + T)
    lens = torch.randint(2, T + 1, (B,), generator=g)
    lens[0] = T
    if B > 2:
        lens[1], lens[2] = 0, 2
    ids = torch.randint(1, 3000, (B, T), generator=g) * (torch.arange(T)[None] < lens[:, None])
    idc = ids.cuda()
    tpos = torch.full((B, T), -1, dtype=torch.int64, device="cuda")
    opos = torch.full((B, K), -1, dtype=torch.int64, device="cuda")
    L.check(L.lib.vk_vlbert_positions(C.c_void_p(idc.data_ptr()), B, T, K, C.c_void_p(tpos.data_ptr()), C.c_void_p(opos.data_ptr()), L.stream_ptr()))
    torch.cuda.synchronize()
    text_end = (ids != 0).sum(1, keepdim=True)
    ar = torch.arange(T)
    shifted = (ar[None] >= text_end).any(0)
    want_t = (ar + K * shifted.long())[None].expand(B, T)
    want_o = text_end.expand(B, K).clone()
    want_o[:, -1] += 1
    assert torch.equal(tpos.cpu(), want_t) and torch.equal(opos.cpu(), want_o)
