"""Fused dropout + residual + LayerNorm kernels against the oracle's layer_norm + autograd.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,H", [(37, 768), (1000, 768), (130, 1024), (9, 64)])
@pytest.mark.parametrize("mode", ["eval", "pre", "post"])
def test_ln_fwd_bwd(M, H, mode):
    from volta_amd import _lib as L, ops
    from oracle import volta_ref as R
    g = torch.Generator().manual_seed(M + H)
    d = torch.randn(M, H, generator=g).bfloat16()
    x = torch.randn(M, H, generator=g).bfloat16()
    gamma = 1 + 0.1 * torch.randn(H, generator=g)
    beta = 0.1 * torch.randn(H, generator=g)
    dy = torch.randn(M, H, generator=g).bfloat16()
    p, seed, site, split = 0.1, 0x1234ABCD5678, 11, (M * 2) // 3
    dev = "cuda"
    seed_t = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.set_seed(seed_t, seed)
    drop = L.dropout_cfg(seed_t.data_ptr(), site, 0.0 if mode == "eval" else p)
    post = 1 if mode == "post" else 0
    out_scale = 0.5 if mode == "post" else 1.0
    dd_, xd, yd, zd = d.to(dev), x.to(dev), torch.empty(M, H, device=dev, dtype=torch.bfloat16), torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.ln_fwd(dd_, xd, gamma.to(dev), beta.to(dev), yd, zd, mean, rstd, M, H, drop=drop, split_row=split, post=post, out_scale=out_scale)
    # oracle with the replayed Philox masks
    df, xf = d.float().requires_grad_(True), x.float().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    keep = torch.ones(M, H)
    if mode != "eval":
        keep = torch.cat([R.philox_keep_mask(seed, site, (split, H), p), R.philox_keep_mask(seed, site + 1, (M - split, H), p)]).float() / (1 - p)
    if post:
        zf = df + xf
        yf = R.layer_norm(zf, gm, bt) * keep * out_scale
    else:
        zf = df * keep + xf
        yf = R.layer_norm(zf, gm, bt) * out_scale
    torch.cuda.synchronize()
    np.testing.assert_allclose(yd.float().cpu().numpy(), yf.detach().numpy(), atol=3e-2, rtol=1e-2)
    np.testing.assert_allclose(mean.cpu().numpy(), zf.mean(-1).detach().numpy(), atol=1e-5)
    yf.backward(dy.float())
    partial = torch.empty(L.lib.vk_ln_bwd_partial_rows(M) * 2 * H, device=dev)
    dz, ddd = torch.empty_like(yd), torch.empty_like(yd)
    dgam, dbet = torch.empty(H, device=dev), torch.empty(H, device=dev)
    ops.ln_bwd(dy.to(dev), zd, mean, rstd, gamma.to(dev), dz, ddd, partial, dgam, dbet, M, H, drop=drop, split_row=split, post=post, out_scale=out_scale)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dz.float().cpu().numpy(), xf.grad.numpy(), atol=3e-2, rtol=2e-2)
    np.testing.assert_allclose(ddd.float().cpu().numpy(), df.grad.numpy(), atol=3e-2, rtol=2e-2)
    tol = 2e-2 * max(1.0, float(gm.grad.abs().max()))
    np.testing.assert_allclose(dgam.cpu().numpy(), gm.grad.numpy(), atol=tol)
    np.testing.assert_allclose(dbet.cpu().numpy(), bt.grad.numpy(), atol=tol)


def test_side_tail_matches_separate_reductions():
    """vk_side_tail (one launch for a sub-layer's slab sums and LayerNorm column reductions) against torch sums."""
    import ctypes as C
    from volta_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(0)
    H, rows = 768, 57
    slabs = torch.randn(3 * 1004, device="cuda", generator=g)           # 3 slabs of 1002 elements at stride 1004 (ragged tail of 2)
    part = torch.randn(rows, 2, H, device="cuda", generator=g)
    big = torch.randn(4, 70000, device="cuda", generator=g)
    d0, d1 = torch.zeros(1002, device="cuda"), torch.zeros(70000, device="cuda")
    dg, db = torch.ones(H, device="cuda"), torch.ones(H, device="cuda")
    part2 = torch.randn(20, 2, H, device="cuda", generator=g)          # a second set of partial records for a shared LayerNorm
    dg2, db2 = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    jobs = (L.TailJob * 4)(L.TailJob(d0.data_ptr(), None, slabs.data_ptr(), None, 1004, 1002, 0, 3, 0, 0),
                           L.TailJob(dg.data_ptr(), db.data_ptr(), part.data_ptr(), None, 0, H, 1, rows, 1, 0),
                           L.TailJob(d1.data_ptr(), None, big.data_ptr(), None, 70000, 70000, 0, 4, 0, 0),
                           L.TailJob(dg2.data_ptr(), db2.data_ptr(), part.data_ptr(), part2.data_ptr(), 20, H, 1, rows, 0, 0))
    L.check(L.lib.vk_side_tail(jobs, 4, L.stream_ptr()))
    torch.cuda.synchronize()
    want0 = sum(slabs[s * 1004:s * 1004 + 1002] for s in range(3))
    assert torch.allclose(d0, want0, atol=1e-5)
    assert torch.allclose(d1, big.sum(0), atol=1e-5)
    assert torch.allclose(dg, 1.0 + part[:, 0].sum(0), atol=1e-4) and torch.allclose(db, 1.0 + part[:, 1].sum(0), atol=1e-4)
    assert torch.allclose(dg2, part[:, 0].sum(0) + part2[:, 0].sum(0), atol=1e-4) and torch.allclose(db2, part[:, 1].sum(0) + part2[:, 1].sum(0), atol=1e-4)
