"""Thin tensor-level wrappers over the C ABI (one call = one kernel launch on the current stream).
Used by the parity tests and by the plan builder; they validate shapes on the host because a wrong
shape in a hand-written kernel is a GPU fault, not an exception."""
import ctypes as C

import torch

from . import _lib as L
from . import _lib as L_
from ._lib import check, ptr, stream_ptr


def _bf16(t):
    assert t.dtype == torch.bfloat16 and t.is_cuda, "bf16 cuda tensor expected"
    return t


def gemm_problem(A, B, Cout, layout, M, N, K, bias=None, R=None, C2=None, bias_grad=None, dyn=None, n_store=0,
                 lda=None, ldb=None, ldc=None, ldr=None):
    """Builds one vk_gemm_problem after checking that every operand really covers what the kernel reads."""
    lda = lda if lda is not None else A.stride(-2)
    ldb = ldb if ldb is not None else B.stride(-2)
    ldc = ldc if ldc is not None else Cout.stride(-2)
    a_rows, a_cols = (K, M) if layout == L.TN else (M, K)
    b_rows, b_cols = (N, K) if layout == L.NT else (K, N)
    for name, t, rows, cols, ld in (("A", A, a_rows, a_cols, lda), ("B", B, b_rows, b_cols, ldb)):
        _bf16(t)
        need = (rows - 1) * ld + cols if rows > 0 else 0
        have = t.numel() - 0 if t.is_contiguous() else t.untyped_storage().nbytes() // 2 - t.storage_offset()
        assert need <= have, "%s operand too small: need %d elements, have %d" % (name, need, have)
    ncols = max(N, n_store)
    need_c = (M - 1) * ldc + ncols if M > 0 else 0
    have_c = Cout.untyped_storage().nbytes() // Cout.element_size() - Cout.storage_offset()
    assert need_c <= have_c, "C too small: need %d have %d" % (need_c, have_c)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N
    if R is not None:
        _bf16(R)
        ldr = ldr if ldr is not None else R.stride(-2)
        assert (M - 1) * ldr + N <= R.untyped_storage().nbytes() // 2 - R.storage_offset()
    if bias_grad is not None:
        assert bias_grad.dtype == torch.float32 and bias_grad.numel() >= M
    if dyn is not None:
        assert dyn.dtype == torch.int32
    return L.GemmProblem(ptr(A), ptr(B), ptr(Cout), ptr(C2), ptr(bias), ptr(R), ptr(bias_grad), ptr(dyn),
                         M, N, K, lda, ldb, ldc, ldr or 0, n_store)


def split_geometry(N):
    """Tile geometry a split accumulation runs on: 256 x 192 tiles when they cover N without a ragged last column tile and 256-wide ones
    would not (N = 768: 4 x 192), 256 x 256 otherwise."""
    return 259 if (N % 192 == 0 and N % 256 != 0) else 258


def split_workspace(layout, M, N, nparts, geometry, device):
    """(ws uint8 tensor, cnt int32 tensor) of one split accumulation; cnt starts (and is left by every launch) at zero."""
    tiles = C.c_int(0)
    nbytes = L.lib.vk_gemm_split_workspace_bytes(layout, M, N, nparts, geometry, C.byref(tiles))
    assert nbytes > 0, "split accumulation: nparts >= 2 and geometry 258 / 259"
    return torch.empty(nbytes, dtype=torch.uint8, device=device), torch.zeros(tiles.value, dtype=torch.int32, device=device)


def split_parts(problem, layout, slices, ws, cnt):
    """K-slices of ONE product as the problems of a split accumulation.  `problem`: the whole product (gemm_problem); `slices`: list of
    (A view, B view, K) -- operand pointers and contraction length of each part (a part may come from other tensors altogether: the
    weight gradient of a parameter shared by two modalities sums row chunks of both)."""
    out = []
    for i, (A, B, K) in enumerate(slices):
        q = L.GemmProblem.from_buffer_copy(problem)
        q.A, q.B, q.K = ptr(A), ptr(B), K
        q.ws, q.cnt, q.part, q.nparts = ptr(ws), ptr(cnt), i, len(slices)
        out.append(q)
    return out


def k_slices(A, B, layout, K, nparts, lda=None, ldb=None):
    """Even K-slices (multiples of 64 elements / rows) of the operands of one product, for split_parts()."""
    step = -(-K // nparts)
    step = -(-step // 64) * 64
    out, k0 = [], 0
    while k0 < K:
        kc = min(step, K - k0)
        a = A[k0:] if layout == L.TN else A[:, k0:]
        b = B[:, k0:] if layout == L.NT else B[k0:]
        out.append((a, b, kc))
        k0 += step
    return out


default_geometry = 0      # what gemm_grouped() passes when the caller names none (the parity tests sweep it; 0 = the library's heuristic)


def gemm_grouped(layout, epilogue, problems, geometry=None):
    """geometry: tile code of vk_gemm_grouped_ex (128 / 258 / 259 / 260, | L.GEMM_PERSISTENT / L.GEMM_ONE_TILE_PER_WG), 0 = heuristic."""
    arr = (L.GemmProblem * len(problems))(*problems)
    check(L.lib.vk_gemm_grouped_ex(layout, epilogue, arr, len(problems), default_geometry if geometry is None else geometry, stream_ptr()))


def gemm_chain(layout, epi_p, producers, epi_c, consumers):
    """vk_gemm_chain: producers (256 x 256 tiles, `sig`) and the consumers of their outputs (256 x 192 tiles, `dep`) in one persistent launch."""
    ap, ac = (L.GemmProblem * len(producers))(*producers), (L.GemmProblem * len(consumers))(*consumers)
    check(L.lib.vk_gemm_chain(layout, epi_p, ap, len(producers), epi_c, ac, len(consumers), stream_ptr()))


def cast_f32_bf16(src, dst):
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    check(L.lib.vk_cast_f32_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()))


def set_seed(seed_t, value):
    assert seed_t.dtype == torch.int64 and seed_t.is_cuda
    check(L.lib.vk_set_seed(ptr(seed_t), C.c_uint64(value & 0xFFFFFFFFFFFFFFFF), stream_ptr()))


def _segs(drop, segs):
    """segs: None -> identity rows on drop.site; else [(site, div, mul, off), (site, div, mul, off)]."""
    arr = (L.DropRows * 2)()
    if segs is None:
        site = drop.site if drop is not None else 0
        segs = [(site, 0, 0, 0), (site + 1, 0, 0, 0)]
    for i, sg in enumerate(segs):
        arr[i] = L.DropRows(*sg)
    return arr


def ln_fwd(d, x, gamma, beta, y, z, mean, rstd, M, H, drop=None, split_row=None, post=0, out_scale=1.0, addvec=None,
           dyn=None, segs=None):
    for t in (d, y):
        _bf16(t)
        assert t.numel() >= M * H
    assert gamma.dtype == torch.float32 and gamma.numel() == H and beta.numel() == H
    assert mean.numel() >= M and rstd.numel() >= M and mean.dtype == torch.float32
    drop = drop or L.dropout_cfg(None, 0, 0.0)
    a = L.LnArgs(ptr(d), ptr(x), ptr(addvec), ptr(gamma), ptr(beta), ptr(y), ptr(z), ptr(mean), ptr(rstd), ptr(dyn), M, H,
                 split_row if split_row is not None else M, post, out_scale, drop, _segs(drop, segs))
    check(L.lib.vk_ln_fwd(C.byref(a), stream_ptr()))


def ln_bwd(dy, z, mean, rstd, gamma, dz, dd, partial, dgamma, dbeta, M, H, drop=None, split_row=None, post=0,
           out_scale=1.0, dyn=None, segs=None):
    for t in (dy, z, dz):
        _bf16(t)
        assert t.numel() >= M * H
    assert partial.numel() >= L.lib.vk_ln_bwd_partial_rows(M) * 2 * H and partial.dtype == torch.float32
    assert dgamma.numel() == H and dbeta.numel() == H
    drop = drop or L.dropout_cfg(None, 0, 0.0)
    a = L.LnBwdArgs(ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz), ptr(dd), ptr(partial), ptr(dgamma),
                    ptr(dbeta), ptr(dyn), M, H, split_row if split_row is not None else M, post, out_scale, 0, drop,
                    _segs(drop, segs))
    check(L.lib.vk_ln_bwd(C.byref(a), stream_ptr()))


def attn_args(qkv, L, masks, ctx, lse, B, nh, gate, drops=None, H=None, dh=64):
    """qkv[m]: [B*L[m], 3H] fused projection output (Q | K | V column blocks); ctx[m]: [B*L[m], H]; H = nh * dh."""
    a = L_.AttnArgs()
    H = H or nh * dh
    assert H == nh * dh
    for m in range(2):
        used_q = gate[m][0] or gate[m][1]
        used_k = gate[0][m] or gate[1][m]
        if not (used_q or used_k):
            continue
        t = _bf16(qkv[m])
        assert t.shape == (B * L[m], 3 * H) and t.is_contiguous()
        es = 2
        a.q[m] = t.data_ptr()
        a.k[m] = t.data_ptr() + H * es
        a.v[m] = t.data_ptr() + 2 * H * es
        a.ld[m] = 3 * H
        a.L[m] = L[m]
        if used_k:
            assert masks[m].dtype == torch.float32 and masks[m].shape == (B, L[m]) and masks[m].is_contiguous()
            a.mask[m] = masks[m].data_ptr()
        if used_q:
            assert _bf16(ctx[m]).shape == (B * L[m], H) and lse[m].numel() == B * nh * L[m] and lse[m].dtype == torch.float32
            a.ctx[m] = ctx[m].data_ptr()
            a.ldo[m] = H
            a.lse[m] = lse[m].data_ptr()
    a.B, a.nh, a.scale, a.dh = B, nh, 1.0 / (dh ** 0.5), dh
    for i in range(2):
        for j in range(2):
            a.gate[i][j] = int(bool(gate[i][j]))
            a.drop[i][j] = drops[i][j] if drops is not None else L_.dropout_cfg(None, 0, 0.0)
    a._refs = (qkv, masks, ctx, lse)      # the struct only holds raw pointers: keep the tensors alive
    return a


def attn_fwd(a):
    check(L_.lib.vk_gated_attn_fwd(C.byref(a), stream_ptr()))


def attn_bwd(a, dctx, dqkv, L, B, gate, H):
    b = L_.AttnBwdArgs()
    for m in range(2):
        if gate[m][0] or gate[m][1]:
            assert _bf16(dctx[m]).shape == (B * L[m], H)
            b.dctx[m] = dctx[m].data_ptr()
        if gate[m][0] or gate[m][1] or gate[0][m] or gate[1][m]:
            t = _bf16(dqkv[m])
            assert t.shape == (B * L[m], 3 * H) and t.is_contiguous()
            b.dq[m] = t.data_ptr()
            b.dk[m] = t.data_ptr() + H * 2
            b.dv[m] = t.data_ptr() + 2 * H * 2
            b.ldg[m] = 3 * H
    check(L_.lib.vk_gated_attn_bwd(C.byref(a), C.byref(b), stream_ptr()))
