import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from volta_amd import _lib as L_, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_attention_gpu import GATES, reference

for gname in ("tt", "tt+vv", "all"):
    gate = GATES[gname]
    B, nh, H, Ls = 2, 12, 768, [20, 37]
    g = torch.Generator().manual_seed(1)
    qkv = [(torch.randn(B * Ls[m], 3 * H, generator=g)).bfloat16() for m in range(2)]
    masks = [torch.zeros(B, Ls[m]) for m in range(2)]
    dctx = [torch.randn(B * Ls[m], H, generator=g).bfloat16() for m in range(2)]
    dev = "cuda"
    qkv_d = [t.to(dev) for t in qkv]
    ctx_d = [torch.zeros(B * Ls[m], H, device=dev, dtype=torch.bfloat16) for m in range(2)]
    lse_d = [torch.zeros(B * nh * Ls[m], device=dev) for m in range(2)]
    dctx_d = [t.to(dev) for t in dctx]
    a = ops.attn_args(qkv_d, Ls, [m.to(dev) for m in masks], ctx_d, lse_d, B, nh, gate, None, H)
    ops.attn_fwd(a)
    dqkv_d = [torch.zeros(B * Ls[m], 3 * H, device=dev, dtype=torch.bfloat16) for m in range(2)]
    ops.attn_bwd(a, dctx_d, dqkv_d, Ls, B, gate, H)
    torch.cuda.synchronize()
    def heads(t, m, i):
        return t.float()[:, i * H:(i + 1) * H].reshape(B, Ls[m], nh, 64).transpose(1, 2).contiguous().requires_grad_(True)
    q = [heads(qkv[m], m, 0) for m in range(2)]; k = [heads(qkv[m], m, 1) for m in range(2)]; v = [heads(qkv[m], m, 2) for m in range(2)]
    keep = [[1.0, 1.0], [1.0, 1.0]]
    ref = reference(q, k, v, masks, gate, B, nh, Ls, keep)
    loss = 0
    for m in range(2):
        if ref[m] is not None:
            print(gname, "ctx", m, float((ctx_d[m].float().cpu().view(B, Ls[m], H) - ref[m]).abs().max()))
            loss = loss + (ref[m] * dctx[m].float().view(B, Ls[m], H)).sum()
            # lse check
            blocks = [mk for mk in range(2) if gate[m][mk]]
            sc = torch.cat([q[m] @ k[mk].transpose(-1, -2) / 8.0 for mk in blocks], -1)
            want = torch.logsumexp(sc, -1)
            print(gname, "lse", m, float((lse_d[m].cpu().view(B, nh, Ls[m]) - want).abs().max()))
    loss.backward()
    for m in range(2):
        got = dqkv_d[m].float().cpu()
        for i, leaf in enumerate((q[m], k[m], v[m])):
            if leaf.grad is None: continue
            want = leaf.grad.transpose(1, 2).reshape(B * Ls[m], H)
            have = got[:, i * H:(i + 1) * H]
            d = (have - want).abs()
            bad = (d > 0.05 * max(1, float(want.abs().max())))
            rows = bad.any(1).nonzero()[:, 0].tolist()[:12]
            cols = bad.any(0).nonzero()[:, 0].tolist()[:24]
            print(gname, "d" + "qkv"[i], m, "maxerr", float(d.max()), "scale", float(want.abs().max()), "nan", int(torch.isnan(have).sum()), "badrows", rows, "badcols", cols)
