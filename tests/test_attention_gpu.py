"""Gated bimodal attention kernels against an fp32 restatement of volta/encoders.py:258-340 with the
Philox dropout masks replayed (oracle.philox_keep_mask).  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GATES = {"tt": [[1, 0], [0, 0]], "tt+vv": [[1, 0], [0, 1]], "tv+vt": [[0, 1], [1, 0]], "all": [[1, 1], [1, 1]],
         "tt+tv": [[1, 1], [0, 0]]}


def reference(q, k, v, masks, gate, B, nh, L, keep, dh=64):
    """q/k/v[m]: [B, nh, L[m], dh] fp32 leaf tensors.  Returns ctx[m] [B, L[m], nh*dh]."""
    out = [None, None]
    for mq in range(2):
        blocks = [mk for mk in range(2) if gate[mq][mk]]
        if not blocks:
            continue
        sc = [q[mq] @ k[mk].transpose(-1, -2) / (dh ** 0.5) + masks[mk][:, None, None, :] for mk in blocks]
        pr = torch.softmax(torch.cat(sc, -1), -1).split([s.shape[-1] for s in sc], -1)
        ctx = 0
        for p, mk in zip(pr, blocks):
            ctx = ctx + (p * keep[mq][mk]) @ v[mk]
        out[mq] = ctx.transpose(1, 2).reshape(B, L[mq], nh * dh)
    return out


@pytest.mark.parametrize("gname", list(GATES))
@pytest.mark.parametrize("T,R", [(20, 37), (38, 37), (20, 101), (6, 4)])
@pytest.mark.parametrize("train", [False, True])
def test_gated_attention(gname, T, R, train):
    _run(gname, T, R, train, 12, 64)


@pytest.mark.parametrize("gname", list(GATES))
@pytest.mark.parametrize("T,R,nh,dh", [(20, 37, 8, 128), (38, 101, 8, 128), (6, 4, 3, 32), (20, 37, 4, 96)])
@pytest.mark.parametrize("train", [False, True])
def test_gated_attention_other_head_sizes(gname, T, R, nh, dh, train):
    """config/vilbert_base.json's 8 heads of 128 (vision stream and co-attention sub-layers) and the other sizes of the generic kernels
    (csrc/attention_generic.hip): same contract, same dropout stream as the 64-wide MFMA kernels."""
    _run(gname, T, R, train, nh, dh)


@pytest.mark.parametrize("gname", list(GATES))
@pytest.mark.parametrize("T,R,nh,dh", [(80, 37, 12, 64), (20, 201, 12, 64), (30, 307, 12, 64), (38, 257, 12, 64), (40, 201, 8, 128)])
@pytest.mark.parametrize("train", [False, True])
def test_gated_attention_long_rows(gname, T, R, nh, dh, train):
    """Rows beyond the MFMA kernels' tiles (more than 64 text tokens / 128 regions): the task configs' lengths -- VCR 80 tokens, Visual7W
    and FlickrGrounding 200 regions (+ global), GuessWhatPointing 256 / 306 (config_tasks/all_tasks.yml:59-70,95-105,306-335) -- on the
    generic kernels, forward and backward, every gate pattern, with the dropout stream replayed."""
    _run(gname, T, R, train, nh, dh)


def _run(gname, T, R, train, nh, dh):
    from volta_amd import _lib as L_, ops
    from oracle import volta_ref as Rf
    gate = GATES[gname]
    B, H = 3, nh * dh
    Ls = [T, R]
    g = torch.Generator().manual_seed(T * 100 + R)
    qkv = [(torch.randn(B * Ls[m], 3 * H, generator=g) * 1.5).bfloat16() for m in range(2)]
    lens = [torch.randint(max(1, Ls[m] // 2), Ls[m] + 1, (B,), generator=g) for m in range(2)]
    masks = [((torch.arange(Ls[m])[None] >= lens[m][:, None]).float() * -10000.0).contiguous() for m in range(2)]
    dctx = [torch.randn(B * Ls[m], H, generator=g).bfloat16() for m in range(2)]
    p, seed = 0.1, 0xC0FFEE123
    dev = "cuda"
    seed_t = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.set_seed(seed_t, seed)
    sites = [[3, 4], [6, 5]]
    drops = [[L_.dropout_cfg(seed_t.data_ptr(), sites[i][j], p if train else 0.0) for j in range(2)] for i in range(2)]
    qkv_d = [t.to(dev) for t in qkv]
    ctx_d = [torch.zeros(B * Ls[m], H, device=dev, dtype=torch.bfloat16) for m in range(2)]
    lse_d = [torch.zeros(B * nh * Ls[m], device=dev) for m in range(2)]
    a = ops.attn_args(qkv_d, Ls, [m.to(dev) for m in masks], ctx_d, lse_d, B, nh, gate, drops, H, dh=dh)
    ops.attn_fwd(a)
    dqkv_d = [torch.zeros(B * Ls[m], 3 * H, device=dev, dtype=torch.bfloat16) for m in range(2)]
    ops.attn_bwd(a, [t.to(dev) for t in dctx], dqkv_d, Ls, B, gate, H)
    torch.cuda.synchronize()

    def heads(t, m, i):
        return t.float()[:, i * H:(i + 1) * H].reshape(B, Ls[m], nh, dh).transpose(1, 2).contiguous().requires_grad_(True)

    q = [heads(qkv[m], m, 0) for m in range(2)]
    k = [heads(qkv[m], m, 1) for m in range(2)]
    v = [heads(qkv[m], m, 2) for m in range(2)]
    keep = [[1.0, 1.0], [1.0, 1.0]]
    if train:
        keep = [[Rf.philox_keep_mask(seed, sites[i][j], (B, nh, Ls[i], Ls[j]), p).float() / (1 - p) for j in range(2)] for i in range(2)]
    ref = reference(q, k, v, masks, gate, B, nh, Ls, keep, dh)
    loss = 0
    for m in range(2):
        if ref[m] is not None:
            got = ctx_d[m].float().cpu().view(B, Ls[m], H)
            np.testing.assert_allclose(got.numpy(), ref[m].detach().numpy(), atol=3e-2, rtol=2e-2)
            loss = loss + (ref[m] * dctx[m].float().view(B, Ls[m], H)).sum()
    loss.backward()
    for m in range(2):
        got = dqkv_d[m].float().cpu()
        for i, leaf in enumerate((q[m], k[m], v[m])):
            if leaf.grad is None:
                continue
            want = leaf.grad.transpose(1, 2).reshape(B * Ls[m], H)
            have = got[:, i * H:(i + 1) * H]
            scale = max(1.0, float(want.abs().max()))
            err = float((have - want).abs().max())
            assert err <= 4e-2 * scale, (gname, m, "qkv"[i], err, scale)
            cos = float((have * want).sum() / (have.norm() * want.norm() + 1e-12))
            assert cos > 0.999, (gname, m, "qkv"[i], cos)
