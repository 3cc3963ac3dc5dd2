"""GEMM kernels (vk_gemm_grouped) against fp32 matmuls of the same bf16 operands.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    from volta_amd import _lib as L, ops
    return L, ops


GEOMETRIES = {"auto": 0, "tile128x128": 128, "tile256x256": 258, "tile256x192": 259, "tile256x128": 260, "tile256x128_4wave": 261, "tile128x128_4wave_ring": 262,
              "tile256x256_persistent": 258 | 0x1000, "tile256x192_persistent": 259 | 0x1000}


@pytest.fixture(params=list(GEOMETRIES.values()), ids=list(GEOMETRIES), autouse=True)
def tile_edge(request):
    """Every GEMM test runs under the tile heuristic and with each tile geometry named explicitly (vk_gemm_grouped_ex)."""
    from volta_amd import ops
    ops.default_geometry = request.param
    yield request.param
    ops.default_geometry = 0


def rnd(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


def ref_mm(layout, A, B, L):
    a, b = A.float(), B.float()
    if layout == L.NT:
        return a @ b.t()
    if layout == L.NN:
        return a @ b
    return a.t() @ b


SHAPES = [(128, 128, 64), (256, 384, 128), (300, 200, 192), (1000, 768, 768), (77, 1601, 256), (5120, 2304, 768), (64, 8, 1024)]


@pytest.mark.parametrize("layout_name", ["NT", "NN", "TN"])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_f32_out(layout_name, M, N, K):
    L, ops = _mods()
    layout = getattr(L, layout_name)
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    if layout == L.TN:
        K = K + 37          # contraction over rows: any count is legal, zero-filled by the bounds check
    Np = (N + 7) // 8 * 8   # row-major operands with N columns need an 8-element aligned leading dim
    Mp = (M + 7) // 8 * 8
    if layout == L.NT:
        A, B = rnd((M, K), g), rnd((N, K), g)
        Av, Bv = A, B
    elif layout == L.NN:
        A = rnd((M, K), g)
        B = torch.zeros(K, Np, device="cuda", dtype=torch.bfloat16)
        B[:, :N] = rnd((K, N), g)
        Av, Bv = A, B[:, :N]
    else:
        A = torch.zeros(K, Mp, device="cuda", dtype=torch.bfloat16)
        A[:, :M] = rnd((K, M), g)
        B = torch.zeros(K, Np, device="cuda", dtype=torch.bfloat16)
        B[:, :N] = rnd((K, N), g)
        Av, Bv = A[:, :M], B[:, :N]
    ld = (N + 63) // 64 * 64
    Cbuf = torch.full((M, ld), float("nan"), device="cuda")
    bias = torch.randn(N, generator=g, device="cuda")
    p = ops.gemm_problem(A, B, Cbuf, layout, M, N, K, bias=bias, n_store=ld)
    ops.gemm_grouped(layout, L.EPI_F32, [p])
    torch.cuda.synchronize()
    ref = ref_mm(layout, Av, Bv, L) + bias
    err = (Cbuf[:, :N] - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-3 * max(scale, 1.0), (err, scale)
    assert torch.all(Cbuf[:, N:] == 0), "pad columns must be written as zeros"


@pytest.mark.parametrize("epi", ["BF16", "GELU", "MULR", "ADDR", "RELU"])
def test_gemm_epilogues(epi):
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 333, 3072, 768
    A, B = rnd((M, K), g, 0.5), rnd((N, K), g, 0.05)
    bias = torch.randn(N, generator=g, device="cuda") * 0.1
    R = rnd((M, N), g)
    Cb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    C2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    e = getattr(L, "EPI_" + epi)
    p = ops.gemm_problem(A, B, Cb, L.NT, M, N, K, bias=None if epi == "MULR" else bias, R=R if epi in ("MULR", "ADDR") else None,
                         C2=C2 if epi == "GELU" else None)
    ops.gemm_grouped(L.NT, e, [p])
    torch.cuda.synchronize()
    u = A.float() @ B.float().t()
    if epi == "BF16":
        ref = u + bias
    elif epi == "GELU":
        x = (u + bias).double()
        ref = torch.nn.functional.gelu(x).float()
        x.requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        assert (C2.float() - x.grad.float()).abs().max().item() < 1e-2
    elif epi == "MULR":
        ref = u * R.float()
    elif epi == "ADDR":
        ref = u + bias + R.float()
    else:
        ref = torch.relu(u + bias)
    err = (Cb.float() - ref).abs()
    assert (err <= 1e-2 * ref.abs() + 2e-2).all(), err.max().item()


def test_gemm_grouped_dynamic_rows_and_bias_grad():
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(9)
    # two forward problems in one launch, the second with a device-side row count
    A1, B1 = rnd((512, 768), g), rnd((768, 768), g, 0.05)
    A2, B2 = rnd((700, 768), g), rnd((768, 768), g, 0.05)
    C1 = torch.zeros(512, 768, device="cuda", dtype=torch.bfloat16)
    C2 = torch.full((700, 768), 7.0, device="cuda", dtype=torch.bfloat16)
    n_dev = torch.tensor([130], device="cuda", dtype=torch.int32)
    ops.gemm_grouped(L.NT, L.EPI_BF16, [ops.gemm_problem(A1, B1, C1, L.NT, 512, 768, 768),
                                        ops.gemm_problem(A2, B2, C2, L.NT, 700, 768, 768, dyn=n_dev)])
    torch.cuda.synchronize()
    r1 = A1.float() @ B1.float().t()
    r2 = A2.float() @ B2.float().t()
    assert (C1.float() - r1).abs().max().item() < 0.05 * r1.abs().max().item()
    assert (C2[:130].float() - r2[:130]).abs().max().item() < 0.05 * r2.abs().max().item()
    assert torch.all(C2[130:] == 7.0), "rows beyond the device count must stay untouched"
    # wgrad with fused bias gradient and a device-side contraction length
    dY, X = rnd((900, 768), g), rnd((900, 3072), g)
    dW = torch.zeros(768, 3072, device="cuda")
    db = torch.zeros(768, device="cuda")
    k_dev = torch.tensor([555], device="cuda", dtype=torch.int32)
    ops.gemm_grouped(L.TN, L.EPI_F32, [ops.gemm_problem(dY, X, dW, L.TN, 768, 3072, 900, bias_grad=db, dyn=k_dev)])
    torch.cuda.synchronize()
    ref = dY[:555].float().t() @ X[:555].float()
    assert (dW - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    assert (db - dY[:555].float().sum(0)).abs().max().item() < 1e-2


def test_gemm_strided_views():
    """Q/K/V slices of a fused [M, 3H] buffer as A operand (lda = 3H) and as C (ldc = 3H)."""
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(3)
    M, H = 200, 768
    qkv = rnd((M, 3 * H), g)
    W = rnd((H, H), g, 0.05)
    out = torch.zeros(M, 3 * H, device="cuda", dtype=torch.bfloat16)
    A = qkv[:, H:2 * H]
    Cv = out[:, 2 * H:]
    ops.gemm_grouped(L.NT, L.EPI_BF16, [ops.gemm_problem(A, W, Cv, L.NT, M, H, H)])
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t()
    assert (Cv.float() - ref).abs().max().item() < 0.05 * ref.abs().max().item()
    assert torch.all(out[:, :2 * H] == 0)


@pytest.mark.parametrize("K", [1, 2, 3, 1601, 3129])
def test_gemm_nn_ragged_contraction_keeps_the_last_row(K):
    """dX = dlogits[M, K] . W[K, N] with the logits' leading dimension padded to 64 and the pad written as 0 (the heads' dgrad): every
    K, odd ones included, must use the last element of the LAST row -- raw buffer loads are range-checked per dword, so a bound that ends
    inside a dword drops it (K = 1, the one-column region-logit head, lost the whole last row)."""
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(K)
    M, N = 148, 768
    ld = (K + 63) // 64 * 64
    A = torch.zeros(M, ld, device="cuda", dtype=torch.bfloat16)
    A[:, :K] = rnd((M, K), g)
    B = rnd((K, N), g)
    C = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    p = ops.gemm_problem(A, B, C, L.NN, M, N, K)
    ops.gemm_grouped(L.NN, L.EPI_BF16, [p])
    torch.cuda.synchronize()
    ref = A[:, :K].float() @ B.float()
    err = (C.float() - ref).abs().max().item()
    assert err <= 1e-2 * max(ref.abs().max().item(), 1.0), (K, err)
    last = (C[-1].float() - ref[-1]).abs().max().item()
    assert last <= 1e-2 * max(ref[-1].abs().max().item(), 1e-3), (K, "last row", last)


@pytest.mark.parametrize("layout_name,M,N,K,nparts", [("NT", 700, 768, 3072, 3), ("NN", 1000, 768, 2304, 4), ("TN", 768, 768, 5120, 5),
                                                      ("TN", 2304, 768, 2000, 2), ("NT", 256, 3072, 1024, 2)])
def test_split_accumulation(layout_name, M, N, K, nparts, tile_edge):
    """K-slices of one product in one launch: partial tiles through the workspace, the last arriver of a tile sums them in part order
    and runs the epilogue (bias / residual / bias gradient once); bitwise reproducible, counters left at zero."""
    if tile_edge != 0:
        pytest.skip("the split path names its own geometry")
    L, ops = _mods()
    layout = getattr(L, layout_name)
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    if layout == L.NT:
        A, B = rnd((M, K), g, 0.5), rnd((N, K), g, 0.1)
    elif layout == L.NN:
        A, B = rnd((M, K), g, 0.5), rnd((K, N), g, 0.1)
    else:
        A, B = rnd((K, M), g, 0.5), rnd((K, N), g, 0.1)
    geo = ops.split_geometry(N)
    ws, cnt = ops.split_workspace(layout, M, N, nparts, geo, "cuda")
    ws.fill_(0xFF)                                          # NaN patterns: every word the reducer reads must have been written
    if layout == L.TN:
        epi, bias, R = L.EPI_F32, None, None
        Cb = torch.full((M, N), float("nan"), device="cuda")
        bg = torch.full((M,), float("nan"), device="cuda")
    else:
        epi = L.EPI_ADDR
        bias, R = torch.randn(N, generator=g, device="cuda"), rnd((M, N), g)
        Cb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        bg = None
    whole = ops.gemm_problem(A, B, Cb, layout, M, N, K, bias=bias, R=R, bias_grad=bg)
    slices = ops.k_slices(A, B, layout, K, nparts)
    assert len(slices) == nparts
    parts = ops.split_parts(whole, layout, slices, ws, cnt)
    ops.gemm_grouped(layout, epi, parts, geometry=geo)
    torch.cuda.synchronize()
    first = Cb.clone()
    ref = ref_mm(layout, A, B, L)
    if layout == L.TN:
        assert (Cb - ref).abs().max().item() <= 2e-3 * max(ref.abs().max().item(), 1.0)
        assert (bg - A.float().sum(0)).abs().max().item() <= 2e-3 * max(A.float().sum(0).abs().max().item(), 1.0)
    else:
        want = ref + bias + R.float()
        assert ((Cb.float() - want).abs() <= 1e-2 * want.abs() + 2e-2).all()
    assert int(cnt.abs().sum()) == 0, "every tile's counter must be back at zero"
    for _ in range(3):                                      # same bits whoever arrives last
        Cb.zero_()
        ops.gemm_grouped(layout, epi, parts, geometry=geo)
        torch.cuda.synchronize()
        assert torch.equal(Cb, first)
    # the parts of another product in the same launch do not disturb it
    ws2, cnt2 = ops.split_workspace(layout, M, N, nparts, geo, "cuda")
    C2b = torch.zeros_like(Cb)
    bg2 = torch.zeros_like(bg) if bg is not None else None
    whole2 = ops.gemm_problem(A, B, C2b, layout, M, N, K, bias=bias, R=R, bias_grad=bg2)
    both = parts + ops.split_parts(whole2, layout, slices, ws2, cnt2)
    Cb.zero_()
    ops.gemm_grouped(layout, epi, both, geometry=geo)
    torch.cuda.synchronize()
    assert torch.equal(Cb, first) and torch.equal(C2b, first)


def test_split_accumulation_rejects_incomplete_groups():
    L, ops = _mods()
    g = torch.Generator(device="cuda").manual_seed(1)
    A, B = rnd((256, 512), g), rnd((256, 512), g)
    Cb = torch.zeros(256, 256, device="cuda", dtype=torch.bfloat16)
    ws, cnt = ops.split_workspace(L.NT, 256, 256, 2, 258, "cuda")
    parts = ops.split_parts(ops.gemm_problem(A, B, Cb, L.NT, 256, 256, 512), L.NT, ops.k_slices(A, B, L.NT, 512, 2), ws, cnt)
    with pytest.raises(RuntimeError):
        ops.gemm_grouped(L.NT, L.EPI_BF16, parts[:1], geometry=258)      # a part is missing
    with pytest.raises(RuntimeError):
        ops.gemm_grouped(L.NT, L.EPI_BF16, parts, geometry=128)          # not a 256-row geometry


# ---- soft boundaries: a GEMM enqueued without the stream-order barrier, its tiles guarded by row-block counters of the GEMM in front of it
def _ffn_pair(L, ops, rows, I, H, seed, layout=None, soft=True):
    """FFN-up + GELU -> FFN-down as the engine lists them: problems per stream of `rows`, plus the counters of the hand-off."""
    layout = L.NT if layout is None else layout
    g = torch.Generator(device="cuda").manual_seed(seed)
    nrb = [M // 256 for M in rows]
    cnt = torch.zeros(1 + 32 * len(rows) + sum(nrb), device="cuda", dtype=torch.int32)
    up, down, keep, outs, sigs = [], [], [], [], []
    cur = 32
    for M in rows:
        x, W1, W2 = rnd((M, H), g, 0.5), rnd((I, H), g, 0.05), rnd((H, I), g, 0.05)
        b1, b2 = torch.randn(I, generator=g, device="cuda") * 0.1, torch.randn(H, generator=g, device="cuda") * 0.1
        h = torch.full((M, I), float("nan"), device="cuda", dtype=torch.bfloat16)
        gp = torch.full((M, I), float("nan"), device="cuda", dtype=torch.bfloat16)
        d = torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16)
        pu = ops.gemm_problem(x, W1, h, L.NT, M, I, H, bias=b1, C2=gp)
        pd = ops.gemm_problem(h, W2, d, L.NT, M, H, I, bias=b2)
        if soft:
            sig = cnt.data_ptr() + 4 * cur
            pu.sig, pu.err = sig, cnt.data_ptr()
            pd.dep, pd.err, pd.dep_need = sig, cnt.data_ptr(), I // 256
            sigs.append(cnt[cur:cur + M // 256])
            cur += (M // 256 + 31) // 32 * 32
        up.append(pu); down.append(pd); keep += [x, W1, W2, b1, b2]; outs.append((h, gp, d))
    return up, down, cnt, sigs, outs, keep


def _run_pair(L, ops, up, down, cnt, soft):
    if soft:
        cnt[1:].zero_()
    ops.gemm_grouped(L.NT, L.EPI_GELU, up, geometry=258)
    ops.gemm_grouped(L.NT, L.EPI_BF16, down, geometry=259 | (L.GEMM_SOFT_START if soft else 0))


@pytest.mark.parametrize("rows", [(5120, 9472), (5120,), (256,), (1024, 256)])
def test_soft_boundary_ffn_pair_equals_the_fenced_pair(rows):
    """The consumer's tiles start while the producer still runs and read row blocks as they are signalled: same bits as the two launches
    behind the stream-order barrier, on every one of 20 back-to-back runs (whoever finishes first), counters at their column-tile count,
    error word clear."""
    L, ops = _mods()
    I, H = 3072, 768
    up, down, cnt, sigs, outs, keep = _ffn_pair(L, ops, rows, I, H, seed=11, soft=True)
    upf, downf, _, _, outsf, keepf = _ffn_pair(L, ops, rows, I, H, seed=11, soft=False)
    _run_pair(L, ops, upf, downf, None, False)
    torch.cuda.synchronize()
    for (h, gp, d), (hf, gpf, df) in zip(outs, outsf):
        assert torch.isfinite(df.float()).all()
    for rep in range(20):
        for h, gp, d in outs:
            d.fill_(float("nan")); h.fill_(float("nan"))
        _run_pair(L, ops, up, down, cnt, True)
        torch.cuda.synchronize()
        assert int(cnt[0]) == 0, "a guarded tile gave up waiting"
        for sig in sigs:
            assert bool((sig == I // 256).all()), sig.tolist()
        for (h, gp, d), (hf, gpf, df) in zip(outs, outsf):
            assert torch.equal(h, hf) and torch.equal(gp, gpf), "producer output differs (rep %d)" % rep
            assert torch.equal(d, df), "consumer read a row block before it was complete (rep %d): %d elements differ" % (rep, int((d != df).sum()))


def test_soft_boundary_under_uneven_load():
    """A second stream holds 64 CUs while the pair runs (late producer workgroups, consumer workgroups that start long before their row
    blocks exist): still the fenced pair's bits, nobody hangs, no poll gives up."""
    L, ops = _mods()
    rows, I, H = (5120, 9472), 3072, 768
    up, down, cnt, sigs, outs, keep = _ffn_pair(L, ops, rows, I, H, seed=5, soft=True)
    upf, downf, _, _, outsf, keepf = _ffn_pair(L, ops, rows, I, H, seed=5, soft=False)
    _run_pair(L, ops, upf, downf, None, False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for rep, (nwg, usec) in enumerate([(64, 150), (128, 60), (200, 100), (64, 400), (255, 30)]):
        for h, gp, d in outs:
            d.fill_(float("nan")); h.fill_(float("nan"))
        cnt[1:].zero_()
        torch.cuda.synchronize()
        L.check(L.lib.vk_hold_cus(nwg, usec, 1, ctypes_ptr(side)))
        ops.gemm_grouped(L.NT, L.EPI_GELU, up, geometry=258)
        ops.gemm_grouped(L.NT, L.EPI_BF16, down, geometry=259 | L.GEMM_SOFT_START)
        L.check(L.lib.vk_hold_cus(nwg, usec // 2, 1, ctypes_ptr(side)))
        torch.cuda.synchronize()
        assert int(cnt[0]) == 0, "a guarded tile gave up waiting"
        for (h, gp, d), (hf, gpf, df) in zip(outs, outsf):
            assert torch.equal(h, hf) and torch.equal(d, df), "rep %d (%d CUs held for %d us)" % (rep, nwg, usec)


def ctypes_ptr(stream):
    import ctypes
    return ctypes.c_void_p(stream.cuda_stream)


def test_soft_boundary_rejects_what_it_cannot_guard():
    L, ops = _mods()
    up, down, cnt, sigs, outs, keep = _ffn_pair(L, ops, (512,), 512, 768, seed=1, soft=True)
    with pytest.raises(RuntimeError):
        ops.gemm_grouped(L.NT, L.EPI_GELU, up, geometry=128)                       # not a 256-row geometry
    with pytest.raises(RuntimeError):
        ops.gemm_grouped(L.NT, L.EPI_GELU, up, geometry=259)                       # 192-wide producer tiles: partial lines
    with pytest.raises(RuntimeError):
        ops.gemm_grouped(L.NT, L.EPI_BF16, down, geometry=128 | L.GEMM_SOFT_START)
    down[0].dep_need = 0
    with pytest.raises(RuntimeError):
        ops.gemm_grouped(L.NT, L.EPI_BF16, down, geometry=259 | L.GEMM_SOFT_START)
    upr, downr, cntr, _, _, keepr = _ffn_pair(L, ops, (300,), 512, 768, seed=1, soft=True)      # ragged rows
    with pytest.raises(RuntimeError):
        ops.gemm_grouped(L.NT, L.EPI_GELU, upr, geometry=258)


# ---- vk_gemm_chain: producer group + consumer group in ONE persistent launch
def _chain_pair(L, ops, rows, I, H, seed, which):
    """which = "fwd": FFN-up + GELU -> FFN-down (NT); "bwd": FFN-down dgrad x gelu' -> FFN-up dgrad + residual gradient (NN)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    cnt = torch.zeros(32 + 64 * len(rows) + sum(M // 256 for M in rows), device="cuda", dtype=torch.int32)
    P, Cn, outs, keep, sigs = [], [], [], [], []
    cur = 32
    for M in rows:
        if which == "fwd":
            x, W1, W2 = rnd((M, H), g, 0.5), rnd((I, H), g, 0.05), rnd((H, I), g, 0.05)
            b1, b2 = torch.randn(I, generator=g, device="cuda") * 0.1, torch.randn(H, generator=g, device="cuda") * 0.1
            mid, mid2 = (torch.full((M, I), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(2))
            out = torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16)
            pp = ops.gemm_problem(x, W1, mid, L.NT, M, I, H, bias=b1, C2=mid2)
            pc = ops.gemm_problem(mid, W2, out, L.NT, M, H, I, bias=b2)
            keep += [x, W1, W2, b1, b2]
        else:
            dd, Wd, Wu = rnd((M, H), g, 0.5), rnd((H, I), g, 0.05), rnd((I, H), g, 0.05)
            gp, dz = rnd((M, I), g, 0.5), rnd((M, H), g, 0.5)
            mid = torch.full((M, I), float("nan"), device="cuda", dtype=torch.bfloat16)
            mid2 = None
            out = torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16)
            pp = ops.gemm_problem(dd, Wd, mid, L.NN, M, I, H, R=gp)
            pc = ops.gemm_problem(mid, Wu, out, L.NN, M, H, I, R=dz)
            keep += [dd, Wd, Wu, gp, dz]
        sig = cnt.data_ptr() + 4 * cur
        pp.sig, pp.err = sig, cnt.data_ptr()
        pc.dep, pc.err, pc.dep_need = sig, cnt.data_ptr(), I // 256
        sigs.append(cnt[cur:cur + M // 256])
        cur += (M // 256 + 31) // 32 * 32
        P.append(pp); Cn.append(pc); outs.append((mid, mid2, out))
    return P, Cn, cnt, sigs, outs, keep


def _plain(p):
    """The same problem without the hand-off fields: for the two-launch reference."""
    import ctypes
    q = type(p)()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(p))
    q.sig = q.dep = q.err = None
    q.dep_need = 0
    return q


@pytest.mark.parametrize("which", ["fwd", "bwd"])
@pytest.mark.parametrize("rows", [(5120, 9472), (5120,), (256,), (1024, 256)])
def test_gemm_chain_equals_the_two_launches(which, rows, tile_edge):
    """Producers and consumers in one persistent launch, consumers behind row-block polls: the two launches' bits, on every one of 20
    back-to-back runs, counters at their column-tile count, error word clear."""
    if tile_edge != 0:
        pytest.skip("the chain names its own geometries")
    L, ops = _mods()
    I, H = 3072, 768
    layout, ep, ec = (L.NT, L.EPI_GELU, L.EPI_BF16) if which == "fwd" else (L.NN, L.EPI_MULR, L.EPI_ADDR)
    P, Cn, cnt, sigs, outs, keep = _chain_pair(L, ops, rows, I, H, 21, which)
    ops.gemm_grouped(layout, ep, [_plain(p) for p in P], geometry=258)
    ops.gemm_grouped(layout, ec, [_plain(p) for p in Cn], geometry=259)
    torch.cuda.synchronize()
    want = [(m.clone(), m2.clone() if m2 is not None else None, o.clone()) for m, m2, o in outs]
    assert all(torch.isfinite(o.float()).all() for _, _, o in want)
    for rep in range(20):
        for m, m2, o in outs:
            m.fill_(float("nan")); o.fill_(float("nan"))
        cnt[1:].zero_()
        ops.gemm_chain(layout, ep, P, ec, Cn)
        torch.cuda.synchronize()
        assert int(cnt[0]) == 0, "a guarded tile gave up waiting"
        assert all(bool((s == I // 256).all()) for s in sigs)
        for (m, m2, o), (wm, wm2, wo) in zip(outs, want):
            assert torch.equal(m, wm) and (m2 is None or torch.equal(m2, wm2)), "producer output differs (rep %d)" % rep
            assert torch.equal(o, wo), "a consumer tile read a row block before it was complete (rep %d): %d elements differ" % (rep, int((o != wo).sum()))


def test_gemm_chain_under_uneven_load_and_with_reserved_cus():
    """A second stream holds CUs while the chain runs (workgroups that start late, consumers that wait long), and the chain on a grid
    that leaves CUs unclaimed (vk_gemm_reserve_cus): the same bits, nobody hangs."""
    L, ops = _mods()
    rows, I, H = (5120, 9472), 3072, 768
    P, Cn, cnt, sigs, outs, keep = _chain_pair(L, ops, rows, I, H, 8, "fwd")
    ops.gemm_grouped(L.NT, L.EPI_GELU, [_plain(p) for p in P], geometry=258)
    ops.gemm_grouped(L.NT, L.EPI_BF16, [_plain(p) for p in Cn], geometry=259)
    torch.cuda.synchronize()
    want = [o.clone() for _, _, o in outs]
    side = torch.cuda.Stream()
    try:
        for rep, (nwg, usec, reserve) in enumerate([(64, 150, 0), (128, 60, 0), (200, 100, 0), (255, 30, 0), (0, 0, 32), (64, 200, 64)]):
            L.lib.vk_gemm_reserve_cus(reserve)
            for m, m2, o in outs:
                m.fill_(float("nan")); o.fill_(float("nan"))
            cnt[1:].zero_()
            torch.cuda.synchronize()
            if nwg:
                L.check(L.lib.vk_hold_cus(nwg, usec, 1, ctypes_ptr(side)))
            ops.gemm_chain(L.NT, L.EPI_GELU, P, L.EPI_BF16, Cn)
            if nwg:
                L.check(L.lib.vk_hold_cus(nwg, usec // 2, 1, ctypes_ptr(side)))
            torch.cuda.synchronize()
            assert int(cnt[0]) == 0, "a guarded tile gave up waiting"
            for (m, m2, o), wo in zip(outs, want):
                assert torch.equal(o, wo), "rep %d (%d CUs held for %d us, %d reserved)" % (rep, nwg, usec, reserve)
    finally:
        L.lib.vk_gemm_reserve_cus(0)


def test_gemm_chain_rejects_what_it_cannot_run():
    L, ops = _mods()
    P, Cn, cnt, sigs, outs, keep = _chain_pair(L, ops, (512,), 512, 768, 1, "fwd")
    with pytest.raises(RuntimeError):
        ops.gemm_chain(L.NT, L.EPI_BF16, P, L.EPI_BF16, Cn)            # not a built epilogue pair
    bad = _plain(Cn[0])
    with pytest.raises(RuntimeError):
        ops.gemm_chain(L.NT, L.EPI_GELU, P, L.EPI_BF16, [bad])         # a consumer without dep
    Cn[0].dep_need = 1
    with pytest.raises(RuntimeError):
        ops.gemm_chain(L.NT, L.EPI_GELU, P, L.EPI_BF16, Cn)            # does not wait for every column tile
    P2, Cn2, *_ = _chain_pair(L, ops, (300,), 512, 768, 1, "fwd")
    with pytest.raises(RuntimeError):
        ops.gemm_chain(L.NT, L.EPI_GELU, P2, L.EPI_BF16, Cn2)          # ragged rows
