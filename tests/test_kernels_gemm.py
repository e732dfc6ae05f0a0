"""CPU: segtran_amd/csrc/gemm.hip executed lane-by-lane on the fiber emulator vs torch fp64."""
import ctypes
import pytest
import torch
from segtran_amd import segx


def _ref(A, B):
    return A.double() @ B.double().transpose(-1, -2)


@pytest.mark.parametrize('M,N,K', [(128, 128, 32), (64, 96, 40), (130, 70, 33), (200, 136, 64), (1, 1, 1), (3, 5, 2)])
@pytest.mark.parametrize('akc,bkc', [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_layouts(backend, M, N, K, akc, bkc):
    L = backend.L
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g, device='cpu').to(backend.dev)
    B = torch.randn(N, K, generator=g, device='cpu').to(backend.dev)             # asymmetric random operands: catches transposes
    Am = A if akc else A.t().contiguous()          # storage [M,K] or [K,M]
    Bm = B if bkc else B.t().contiguous()
    a_str = (0, 0, K, 1) if akc else (0, 0, 1, M)
    b_str = (0, 0, K, 1) if bkc else (0, 0, 1, N)
    C = torch.full((M, N), float('nan'))
    L.gemm(Am, Bm, C, M, N, K, a_str, b_str, (0, 0, N))
    ref = _ref(A, B)
    assert (C.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('tile', [segx.TILE_128x128, segx.TILE_64x64, segx.TILE_128x32, segx.TILE_32x128, segx.TILE_64x128])
@pytest.mark.parametrize('M,N,K,akc,bkc,sk', [(200, 136, 72, True, True, 1), (36, 260, 100, True, False, 1), (132, 24, 200, False, False, 3),
                                              (68, 68, 64, False, True, 2)])
def test_gemm_every_tile_shape_gives_the_same_result(backend, tile, M, N, K, akc, bkc, sk):
    """The workgroup tile is a tuning knob: every TileCfg must produce the k-ordered fp32 result (bias + split-K included)."""
    L = backend.L
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    nb = 2
    A = torch.randn(nb, M, K, generator=g, device='cpu').to(backend.dev)
    B = torch.randn(nb, N, K, generator=g, device='cpu').to(backend.dev)
    bias = torch.randn(M, generator=g, device='cpu').to(backend.dev)
    Am = A if akc else A.transpose(1, 2).contiguous()
    Bm = B if bkc else B.transpose(1, 2).contiguous()
    a_str = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    b_str = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
    C = torch.full((nb, M, N), float('nan'))
    ws = torch.empty(sk * nb * M * N) if sk > 1 else None
    L.gemm(Am, Bm, C, M, N, K, a_str, b_str, (0, M * N, N), nb=(1, nb), alpha=0.5, bias=bias, bias_mode=segx.BIAS_M,
           splitk=sk, workspace=ws, tile=tile)
    ref = 0.5 * _ref(A, B) + bias.double()[None, :, None]
    assert (C.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_gemm_batched_modes_bias_alpha_gmax(backend):
    """squeeze-out QK^T view: Q [B,N,4*d], K [B,A,4*d] -> S [4,B,N,A] (mode-major), scaled, max tracked."""
    L = backend.L
    g = torch.Generator(device='cpu').manual_seed(5)
    Bn, N, A, Mo, d = 2, 70, 24, 4, 12
    Q = torch.randn(Bn, N, Mo * d, generator=g, device='cpu').to(backend.dev)
    Kt = torch.randn(Bn, A, Mo * d, generator=g, device='cpu').to(backend.dev)
    S = torch.zeros(Mo, Bn, N, A)
    gmax = torch.zeros(1)
    L.gemm(Q, Kt, S, N, A, d, (N * Mo * d, d, Mo * d, 1), (A * Mo * d, d, Mo * d, 1), (N * A, Bn * N * A, A),
           nb=(Bn, Mo), alpha=0.25, gmax=gmax)
    ref = torch.einsum('bnmd,bamd->mbna', Q.view(Bn, N, Mo, d).double(), Kt.view(Bn, A, Mo, d).double()) * 0.25
    assert (S.double() - ref).abs().max().item() < 1e-4
    assert abs(gmax.item() - max(ref.max().item(), 0.0)) < 1e-4


def test_gemm_gelu_epilogue_grouped_bias(backend):
    """grouped (per-mode) linear with per-mode bias + GELU epilogue writing the pre-activation."""
    L = backend.L
    g = torch.Generator(device='cpu').manual_seed(6)
    Mo, R, F = 4, 50, 36
    H = torch.randn(Mo, R, F, generator=g, device='cpu').to(backend.dev)
    W = torch.randn(Mo, F, F, generator=g, device='cpu').to(backend.dev) * 0.3
    b = torch.randn(Mo, F, generator=g, device='cpu').to(backend.dev)
    Y = torch.zeros(Mo, R, F); T = torch.zeros(Mo, R, F)
    L.gemm(H, W, Y, R, F, F, (0, R * F, F, 1), (0, F * F, F, 1), (0, R * F, F), nb=(1, Mo), bias=b,
           bias_mode=segx.BIAS_N, bias_b1=F, epilogue=segx.EPI_GELU, aux=T)
    Tref = torch.einsum('mrf,mgf->mrg', H.double(), W.double()) + b[:, None, :].double()
    assert (T.double() - Tref).abs().max().item() < 1e-4
    assert (Y.double() - torch.nn.functional.gelu(Tref)).abs().max().item() < 1e-4


def test_gemm_gelu_dropout_is_mask_times_scale(backend):
    L = backend.L
    g = torch.Generator(device='cpu').manual_seed(7)
    R, F = 64, 32
    H = torch.randn(R, F, generator=g, device='cpu').to(backend.dev); W = torch.randn(F, F, generator=g, device='cpu').to(backend.dev) * 0.3
    Y = torch.zeros(R, F); T = torch.zeros(R, F); Y2 = torch.zeros(R, F)
    kw = dict(epilogue=segx.EPI_GELU, aux=T, dropout_p=0.25, seed=123, offset=9)
    L.gemm(H, W, Y, R, F, F, (0, 0, F, 1), (0, 0, F, 1), (0, 0, F), **kw)
    L.gemm(H, W, Y2, R, F, F, (0, 0, F, 1), (0, 0, F, 1), (0, 0, F), **kw)
    assert torch.equal(Y, Y2)                                   # counter-based: reproducible
    full = torch.nn.functional.gelu(T)
    kept = Y != 0
    assert torch.allclose(Y[kept], full[kept] / 0.75, atol=1e-5)
    frac = 1.0 - kept.float().mean().item()
    assert 0.17 < frac < 0.33


def test_gemm_splitk_and_bias_m(backend):
    """weight-gradient shape: dW = dY^T X with K = rows (TN), split-K 3, + conv-style per-row bias."""
    L = backend.L
    g = torch.Generator(device='cpu').manual_seed(8)
    R, Fo, Fi = 210, 40, 24
    dY = torch.randn(R, Fo, generator=g, device='cpu').to(backend.dev); X = torch.randn(R, Fi, generator=g, device='cpu').to(backend.dev); bias = torch.randn(Fo, generator=g, device='cpu').to(backend.dev)
    dW = torch.zeros(Fo, Fi); ws = torch.zeros(3 * Fo * Fi)
    L.gemm(dY, X, dW, Fo, Fi, R, (0, 0, 1, Fo), (0, 0, 1, Fi), (0, 0, Fi), alpha=0.5, bias=bias,
           bias_mode=segx.BIAS_M, splitk=3, workspace=ws)
    ref = 0.5 * dY.double().t() @ X.double() + bias[:, None].double()
    assert (dW.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize('nb,sk,M,N', [((2, 1), 1, 40, 24), ((3, 2), 1, 40, 24), ((5, 4), 1, 33, 17), ((2, 2), 3, 70, 50), ((2, 1), 1, 600, 520)])
def test_gemm_batch_reduce_sums_over_the_batch(backend, nb, sk, M, N):
    """batch_reduce: ONE [M, N] result = alpha * sum over the batch of A_z B_z^T (+ bias once) -- the shape of a weight gradient whose operand
    is broadcast over the batch.  The cases walk the three slab-reduction kernels (16 / 4 threads per output, one thread per output)."""
    L = backend.L
    g = torch.Generator(device='cpu').manual_seed(9)
    K = 26
    A = torch.randn(nb[0], nb[1], M, K, generator=g, device='cpu').to(backend.dev)
    B = torch.randn(nb[0], nb[1], N, K, generator=g, device='cpu').to(backend.dev)
    bias = torch.randn(N, generator=g, device='cpu').to(backend.dev)
    C = torch.zeros(M, N)
    ws = torch.zeros(sk * nb[0] * nb[1] * M * N)
    L.gemm(A, B, C, M, N, K, (nb[1] * M * K, M * K, K, 1), (nb[1] * N * K, N * K, K, 1), (0, 0, N), nb=nb, alpha=0.5, bias=bias, bias_mode=segx.BIAS_N,
           splitk=sk, workspace=ws, batch_reduce=True)
    ref = 0.5 * torch.einsum('xymk,xynk->mn', A.double(), B.double()) + bias.double()[None]
    assert (C.double() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item() / 10)


@pytest.mark.parametrize('nb,M,N,K', [((2, 1), 192, 32, 2048), ((3, 1), 144, 24, 1024), ((2, 1), 32, 192, 2048), ((3, 2), 24, 48, 1024), ((4, 1), 24, 24, 3072),
                                      ((2, 1), 3, 160, 2048), ((2, 1), 56, 32, 4096), ((2, 1), 100, 7, 2048), ((1, 1), 128, 32, 4096)])
def test_gemm_skinny_weight_gradient_streams(backend, nb, M, N, K):
    """gemm_skinny.hip (segx_gemm_plan -> SEGX_TILE_SKINNY_NT): the batch-reduced product of two k-contiguous operands, one of <= 32 rows, over a long K -- the weight
    gradients of the backbone's first pointwise convolutions (efficientnet/model.py:96, 113) -- in one streaming pass with the batch walked inside the kernel; against
    fp64, and against the tile kernels' split-K slabs (knob 18 = 0) which it replaces.  A persistent grid of 8 (knob 9) keeps the emulator run short and makes every
    workgroup cross a batch-member boundary or end inside one."""
    L = backend.L
    if backend.name == 'emu' and (M, N) in ((32, 192), (24, 24), (56, 32), (100, 7)):
        pytest.skip('emulator: one shape per block arrangement is enough (6 x 1, 5 x 1, 1 x 2, 1 x 5, 4 x 1); the device runs all nine')
    g = torch.Generator(device='cpu').manual_seed(11 + M + N)
    A = torch.randn(nb[0], nb[1], M, K, generator=g, device='cpu').to(backend.dev)
    B = torch.randn(nb[0], nb[1], N, K, generator=g, device='cpu').to(backend.dev)
    bias = torch.randn(M, generator=g, device='cpu').to(backend.dev)
    ref = 0.25 * torch.einsum('xymk,xynk->mn', A.double(), B.double()).cpu() + bias.double().cpu()[:, None]
    args = (M, N, K, (nb[1] * M * K, M * K, K, 1), (nb[1] * N * K, N * K, K, 1), (0, 0, N))
    out = {}
    assert L.c.segx_tune(9, 8) == 0
    try:
        for knob in (1, 0):
            assert L.c.segx_tune(18, knob) == 0
            d = segx.GemmDesc()
            d.M, d.N, d.K, d.nb0, d.nb1 = M, N, K, nb[0], nb[1]
            d.a_b0, d.a_b1, d.a_m, d.a_k = args[3]; d.b_b0, d.b_b1, d.b_n, d.b_k = args[4]
            d.batch_reduce = 1
            t, sk = ctypes.c_int(0), ctypes.c_int(0)
            assert L.c.segx_gemm_plan(A.data_ptr(), B.data_ptr(), ctypes.byref(d), ctypes.byref(t), ctypes.byref(sk)) == 0
            assert (t.value == segx.TILE_SKINNY_NT) == (knob == 1), (t.value, sk.value)
            C = torch.full((M, N), float('nan'))
            L.gemm(A, B, C, *args, nb=nb, alpha=0.25, bias=bias, bias_mode=segx.BIAS_M, splitk=0, batch_reduce=True)
            out[knob] = C.double().cpu()
            assert (out[knob] - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    finally:
        assert L.c.segx_tune(18, 1) == 0 and L.c.segx_tune(9, 256) == 0
    assert (out[1] - out[0]).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # a caller that names the tile for a product the kernel does not serve (K not a multiple of 64) quietly gets the planner's tile
    A2, B2 = A[..., :K - 4].contiguous(), B[..., :K - 4].contiguous()
    C2 = torch.zeros(M, N); ws = torch.zeros(nb[0] * nb[1] * M * N)
    L.gemm(A2, B2, C2, M, N, K - 4, (nb[1] * M * (K - 4), M * (K - 4), K - 4, 1), (nb[1] * N * (K - 4), N * (K - 4), K - 4, 1), (0, 0, N), nb=nb, splitk=1, workspace=ws,
           tile=segx.TILE_SKINNY_NT, batch_reduce=True)
    ref2 = torch.einsum('xymk,xynk->mn', A2.double(), B2.double()).cpu()
    assert (C2.double().cpu() - ref2).abs().max().item() < 2e-5 * max(1.0, ref2.abs().max().item())


def test_gemm_rejects_bad_strides(backend):
    L = backend.L
    A = torch.zeros(8, 8); C = torch.zeros(8, 8)
    with pytest.raises(RuntimeError, match='unit stride'):
        L.gemm(A, A, C, 8, 8, 4, (0, 0, 8, 2), (0, 0, 8, 1), (0, 0, 8))


@pytest.mark.parametrize('shape', [(2, 12, 5, 7), (1, 8, 3, 4, 5), (3, 20, 6, 6)])
@pytest.mark.parametrize('bias', [True, False])
def test_conv1x1_autograd_vs_torch(backend, shape, bias):
    """Pointwise conv on NC[D]HW through the batched GEMM (weights broadcast over the batch, per-row bias)."""
    from segtran_amd import functional as SF
    import torch.nn.functional as F
    g = torch.Generator(device='cpu').manual_seed(sum(shape))
    Cin, Cout = shape[1], 10
    x = torch.randn(*shape, generator=g, device='cpu').to(backend.dev).requires_grad_(True)
    wshape = (Cout, Cin) + (1,) * (len(shape) - 2)
    w = torch.randn(*wshape, generator=g, device='cpu').to(backend.dev).requires_grad_(True)
    b = torch.randn(Cout, generator=g, device='cpu').to(backend.dev).requires_grad_(True) if bias else None
    G = torch.randn(shape[0], Cout, *shape[2:], generator=g, device='cpu').to(backend.dev)
    y = SF.conv1x1(x, w, b)
    y.backward(G)
    got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone() if bias else None)
    x.grad = None; w.grad = None
    if bias:
        b.grad = None
    yr = (F.conv2d if len(shape) == 4 else F.conv3d)(x, w, b)
    yr.backward(G)
    for a, r in zip(got, (yr.detach(), x.grad, w.grad, b.grad if bias else None)):
        if r is not None:
            assert (a - r).abs().max().item() < 1e-4 * max(1.0, r.abs().max().item())


@pytest.mark.parametrize('p', [0.0, 0.3])
def test_fusion_gemm_with_gelu_dropout_epilogue_mode_major(backend, p):
    """The squeeze-out fusion h[m,b] = dropout(gelu(P[m,b] @ u[b,:,m,:] + bias)): A k-contiguous, B row-contiguous, batch (B, M)
    written mode-major.  The dropout mask is keyed by the element's offset in C, so the backward pass (which regenerates it over
    the contiguous tensor) sees the forward's mask: values and both operand gradients against PyTorch with the inferred mask."""
    from segtran_amd import functional as SF
    import torch.nn.functional as F
    B, M, U1, U2, Fd = 2, 3, 40, 12, 36
    g = torch.Generator(device='cpu').manual_seed(77)
    mk = lambda *sh: torch.randn(*sh, generator=g, device='cpu').to(backend.dev)      # noqa: E731
    P = mk(M, B, U1, U2).requires_grad_(True); u = mk(B, U2, M, Fd).requires_grad_(True); bias = mk(Fd).requires_grad_(True)
    spec = SF.GemmSpec(U1, Fd, U2, (U1 * U2, B * U1 * U2, U2, 1), (U2 * M * Fd, Fd, 1, M * Fd), (U1 * Fd, B * U1 * Fd, Fd), (M, B, U1, Fd),
                       nb=(B, M), bias_mode=SF.BIAS_N)
    SF.manual_seed(3)
    h = SF.bgemm(P, u, spec, bias=bias, gelu=True, drop_p=p)
    Pr, ur, br = (t.detach().clone().requires_grad_(True) for t in (P, u, bias))
    pre = torch.einsum('mbik,bkmf->mbif', Pr, ur) + br
    act = F.gelu(pre)
    keep = (h.detach() != 0) | (act.detach() == 0)
    if p > 0:
        assert abs(keep.float().mean().item() - (1 - p)) < 0.05
    href = act * keep / (1 - p)
    assert (h.detach() - href.detach()).abs().max().item() <= 2e-5 * href.detach().abs().max().item()
    G = mk(M, B, U1, Fd)
    h.backward(G); href.backward(G)
    for a, b in ((P, Pr), (u, ur), (bias, br)):
        assert (a.grad - b.grad).abs().max().item() <= 1e-4 * b.grad.abs().max().item()


# ---- bf16x6 tile engine (gemm_x6.h): same entry point, same descriptor, fp32-equivalent results -----------------------------------------
def _x6_case(L, dev, M, N, K, akc, bkc, nb=1, sk=1, tile=segx.TILE_AUTO, seed=0, **kw):
    g = torch.Generator(device='cpu').manual_seed(seed + M * 7 + N * 3 + K)
    A = torch.randn(nb, M, K, generator=g, device='cpu').to(dev)
    B = (torch.randn(nb, N, K, generator=g, device='cpu') * 0.3).to(dev)
    Am = A if akc else A.transpose(1, 2).contiguous()
    Bm = B if bkc else B.transpose(1, 2).contiguous()
    a_str = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    b_str = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
    C = torch.full((nb, M, N), float('nan'), device=dev)
    ws = torch.empty(sk * nb * M * N, device=dev) if sk > 1 else None
    L.gemm(Am, Bm, C, M, N, K, a_str, b_str, (0, M * N, N), nb=(1, nb), splitk=sk, workspace=ws, tile=tile, **kw)
    return A, B, C


@pytest.mark.parametrize('tile', [segx.TILE_128x128, segx.TILE_64x128, segx.TILE_64x64, segx.TILE_256x128, segx.TILE_WS128x128])
@pytest.mark.parametrize('M,N,K,akc,bkc,sk', [(200, 136, 72, True, True, 1), (132, 260, 100, True, False, 1), (132, 84, 200, False, False, 3),
                                              (68, 68, 64, False, True, 2), (256, 128, 32, False, False, 1), (52, 64, 8, True, False, 1)])
def test_x6_engine_matches_fp64_every_tile_and_layout(backend, tile, M, N, K, akc, bkc, sk):
    """bf16x6 engine: all four operand layouts, ragged edges in M, N and K, batches, alpha, per-row bias, split-K -- error against fp64
    at fp32-rounding level (3e-6 of the result scale; the fp32 MFMA engine itself measures 1e-6 on these sizes)."""
    L = backend.L
    prev = L.set_engine('x6')
    try:
        L.x6_launches()
        bias = torch.randn(M, generator=torch.Generator(device='cpu').manual_seed(1), device='cpu').to(backend.dev)
        A, B, C = _x6_case(L, backend.dev, M, N, K, akc, bkc, nb=2, sk=sk, tile=tile, alpha=0.5, bias=bias, bias_mode=segx.BIAS_M)
        assert L.x6_launches() == 1, 'the GEMM did not run on the bf16x6 engine'
    finally:
        L.set_engine(prev)
    ref = 0.5 * _ref(A, B) + bias.double()[None, :, None]
    err = (C.double() - ref).abs().max().item()
    assert err < 3e-6 * max(1.0, ref.abs().max().item()), err


def test_x6_engine_planner_gmax_gelu_and_fallbacks(backend):
    L = backend.L
    dev = backend.dev
    prev = L.set_engine('x6')
    try:
        L.x6_launches()
        # (a) library-planned tile + split-K (splitk=0 path of the binding) on a weight-gradient shape
        A, B, C = _x6_case(L, dev, 96, 80, 2048, False, False, nb=1, sk=0)
        assert L.x6_launches() == 1
        assert (C.double() - _ref(A, B)).abs().max().item() < 3e-6 * _ref(A, B).abs().max().item()
        # (b) running max of the scores (N5 clip flag) in the epilogue
        gmax = torch.zeros(1, device=dev)
        A, B, C = _x6_case(L, dev, 72, 64, 48, True, True, gmax=gmax, alpha=0.25)
        ref = 0.25 * _ref(A, B)
        assert abs(gmax.item() - max(ref.max().item(), 0.0)) < 1e-5 and (C.double() - ref).abs().max().item() < 3e-6 * ref.abs().max().item()
        # (c) fused bias + GELU + dropout epilogue (pre-activation in aux), per-mode bias
        Mo, R, F = 2, 100, 64
        g = torch.Generator(device='cpu').manual_seed(6)
        H = torch.randn(Mo, R, F, generator=g, device='cpu').to(dev); W = (torch.randn(Mo, F, F, generator=g, device='cpu') * 0.3).to(dev)
        b = torch.randn(Mo, F, generator=g, device='cpu').to(dev)
        Y = torch.zeros(Mo, R, F, device=dev); T = torch.zeros(Mo, R, F, device=dev)
        L.gemm(H, W, Y, R, F, F, (0, R * F, F, 1), (0, F * F, F, 1), (0, R * F, F), nb=(1, Mo), bias=b, bias_mode=segx.BIAS_N, bias_b1=F,
               epilogue=segx.EPI_GELU, aux=T)
        Tref = torch.einsum('mrf,mgf->mrg', H.double(), W.double()) + b[:, None, :].double()
        assert (T.double() - Tref).abs().max().item() < 1e-5 and (Y.double() - torch.nn.functional.gelu(Tref)).abs().max().item() < 1e-5
        assert L.x6_launches() == 2
        # (d) skinny / odd operands stay on the fp32 engine
        _x6_case(L, dev, 24, 256, 64, True, True)              # 24 rows: streamed through the 32-row fp32 tile
        _x6_case(L, dev, 130, 70, 33, True, True)              # K = 33: not float4-legal
        assert L.x6_launches() == 0
    finally:
        L.set_engine(prev)


def test_x6_engine_k1792_accuracy_vs_fp32_engine(backend):
    """The dominant contraction length of the model (K = 1792): both engines against fp64 -- the bf16x6 engine must be as accurate as the
    fp32 MFMA (the parity bar of the whole step rests on this)."""
    L = backend.L
    M, N, K = 64, 64, 1792
    prev = L.set_engine('f32')
    try:
        A, B, C32 = _x6_case(L, backend.dev, M, N, K, True, True, seed=3)
        L.set_engine('x6')
        _, _, C6 = _x6_case(L, backend.dev, M, N, K, True, True, seed=3)
    finally:
        L.set_engine(prev)
    ref = _ref(A, B)
    e32 = (C32.double() - ref).abs().max().item() / ref.abs().max().item()
    e6 = (C6.double() - ref).abs().max().item() / ref.abs().max().item()
    # device: measured 1.2e-6 vs 1.0e-6 (profiles/r01_l_bf16x6_proto.txt).  The emulator models the bf16 MFMA pessimistically -- as a
    # sequential fp32 fmaf chain over all 16 products, six MFMAs per 16 k = 6x the rounding steps of the fp32 engine -- hence its looser bound
    tol = 3e-6 if backend.name == 'hip' else 1.2e-5
    assert e6 < tol and (backend.name != 'hip' or e6 < 3 * e32 + 1e-7), (e6, e32)


def test_x6_split_early_schedule_gives_the_same_result(backend):
    """segx_tune knob 6, value 6: the split-early schedule of the 128 x 128 k-contiguous kernel (conversion arithmetic of the next tile dealt
    out between the matrix instructions of the current one) is a scheduling variant -- bit-identical results to the product schedule."""
    L = backend.L
    prev = L.set_engine('x6')
    try:
        out = []
        for v in (0, 6, 7):
            assert L.c.segx_tune(6, v) == 0
            A, B, C = _x6_case(L, backend.dev, 200, 136, 104, True, True, nb=2, tile=segx.TILE_128x128, seed=11)
            out.append(C.clone())
        assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])
        assert (out[0].double() - _ref(A, B)).abs().max().item() < 3e-6 * _ref(A, B).abs().max().item()
    finally:
        L.c.segx_tune(6, 0)
        L.set_engine(prev)


@pytest.mark.parametrize('engine,tile', [('x6', segx.TILE_128x128), ('x6', segx.TILE_64x64), ('x6', segx.TILE_256x128), ('x6', segx.TILE_WS128x256), ('f32', segx.TILE_128x128)])
def test_tile_walk_order_gives_the_same_result(backend, engine, tile):
    """segx_tune knob 19 (gemm_core.h tile_walk): a small A operand with several row-tiles against a B operand more than twice its size is walked M fastest inside
    an XCD's run (the tiles sharing a B column-panel next to each other) instead of N fastest -- a different ORDER of the same tiles: bit-identical results, on the
    4-wave kernels (tile_coord) and on the persistent wave-specialised ones (ws_item_coord), with ragged edges, a batch and split-K slabs."""
    L = backend.L
    prev = L.set_engine(engine)
    try:
        assert L.c.segx_tune_get(19) == 1                                   # the default
        if tile in (segx.TILE_256x128, segx.TILE_WS128x256):
            assert L.c.segx_tune(9, 8) == 0                                 # 8 persistent workgroups: several rounds of items
        out = []
        for v in (1, 0):
            assert L.c.segx_tune(19, v) == 0
            A, B, C = _x6_case(L, backend.dev, 300, 1100, 64, True, False, nb=2, sk=2, tile=tile, seed=19)
            out.append(C.clone())
        assert L.c.segx_tune(19, 2) == -1                                   # an unknown setting is an error
    finally:
        L.c.segx_tune(19, 1); L.c.segx_tune(9, 256)
        L.set_engine(prev)
    assert torch.equal(out[0], out[1])
    ref = _ref(A, B)
    assert (out[0].double() - ref).abs().max().item() < 3e-6 * ref.abs().max().item()


@pytest.mark.parametrize('tile', [segx.TILE_256x128, segx.TILE_WS128x128, segx.TILE_WS128x256, segx.TILE_WS64x256, segx.TILE_WS96x256, segx.TILE_WS256x96])
@pytest.mark.parametrize('M,N,K,akc,bkc,sk,nb', [(520, 264, 96, True, True, 1, 3), (300, 392, 128, True, False, 2, 2), (260, 136, 192, False, False, 3, 2),
                                                 (264, 260, 96, False, True, 1, 2), (264, 136, 104, True, True, 1, 2)])
def test_x6_wave_specialised_persistent_stream(backend, tile, M, N, K, akc, bkc, sk, nb):
    """gemm_x6ws.h: producers / consumers of a persistent workgroup walk a STREAM of work items (here 8 workgroups for 12-72 items, ragged
    edges, batches, split-K slabs; K = 104 is not a whole number of 32-k stages and must quietly take the 4-wave kernel, and so must a 96-row tile whose 96-row
    side is a row-contiguous operand): results must equal the 4-wave bf16x6 kernel bit for bit (same products,
    same order per accumulator) and fp64 to fp32 rounding."""
    L = backend.L
    prev = L.set_engine('x6')
    try:
        assert L.c.segx_tune(9, 8) == 0
        L.x6_launches()
        bias = torch.randn(N, generator=torch.Generator(device='cpu').manual_seed(2), device='cpu').to(backend.dev)
        A, B, C = _x6_case(L, backend.dev, M, N, K, akc, bkc, nb=nb, sk=sk, tile=tile, alpha=0.25, bias=bias, bias_mode=segx.BIAS_N, seed=5)
        assert L.x6_launches() == 1
        _, _, C0 = _x6_case(L, backend.dev, M, N, K, akc, bkc, nb=nb, sk=sk, tile=segx.TILE_128x128, alpha=0.25, bias=bias, bias_mode=segx.BIAS_N, seed=5)
    finally:
        L.c.segx_tune(9, 256)
        L.set_engine(prev)
    ref = 0.25 * _ref(A, B) + bias.double()[None, None, :]
    assert (C.double() - ref).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())
    assert torch.equal(C, C0)


def _split3_host(x):
    """x = hi + mid + lo in bf16, each step rounded to nearest even (what split3_pair does in registers); returns the three int16 bit planes."""
    x = x.detach().float().cpu()
    hi = x.to(torch.bfloat16); r1 = x - hi.float()
    mid = r1.to(torch.bfloat16); r2 = r1 - mid.float()
    lo = r2.to(torch.bfloat16)
    return [t.view(torch.int16) for t in (hi, mid, lo)]


@pytest.mark.parametrize('kcontig', [True, False])
def test_x6_presplit_planes_are_the_kernel_split(backend, kcontig):
    """segx_x6_presplit: [nb][3][rows][K] bf16 planes of an fp32 operand in either unit-stride layout, batch strides, a shared (stride-0) batch dim;
    bit-identical to the round-to-nearest-even three-way split done on the host."""
    L = backend.L
    g = torch.Generator(device='cpu').manual_seed(3)
    rows, K, nb = 20, 48, 3
    W = torch.randn(nb, rows, K, generator=g, device='cpu') * torch.logspace(-3, 3, K, device='cpu')[None, None, :]
    Wd = (W if kcontig else W.transpose(1, 2).contiguous()).to(backend.dev)
    s_row, s_k = (K, 1) if kcontig else (1, rows)
    planes, b0, b1 = L.x6_presplit(Wd, rows, K, s_row, s_k, nb=(2, nb), s_b=(0, rows * K))
    assert (b0, b1) == (0, 3 * rows * K) and planes.numel() == nb * 3 * rows * K
    got = planes.cpu().view(nb, 3, rows, K)
    for z in range(nb):
        for p, ref in enumerate(_split3_host(W[z])):
            assert torch.equal(got[z, p], ref), (z, p)
    hi, mid, lo = (got[:, p].view(torch.bfloat16).double() for p in range(3))
    assert ((hi + mid + lo) - W.double()).abs().max() <= 2.0 ** -23 * W.abs().max()


@pytest.mark.parametrize('tile', [segx.TILE_256x128, segx.TILE_WS128x256])
@pytest.mark.parametrize('M,N,K,akc,bkc,sk,nb,shared', [(300, 392, 128, True, True, 1, 2, False), (264, 520, 96, False, True, 2, 3, True),
                                                        (260, 136, 192, True, False, 1, 2, False), (520, 264, 64, False, False, 3, 2, True)])
def test_x6_presplit_b_operand_gives_identical_results(backend, tile, M, N, K, akc, bkc, sk, nb, shared):
    """segx_gemm_desc.b_planes: the wave-specialised kernels with the B operand split ahead of time (copy-only B loader) against the same kernels
    splitting B in registers -- the LDS image is the same, so the results must agree bit for bit; ragged N (clamped rows), batches with their own
    or a shared B, split-K slabs, both layouts of A and of the fp32 B the planes are made from."""
    L = backend.L
    prev = L.set_engine('x6')
    try:
        assert L.c.segx_tune(9, 8) == 0
        g = torch.Generator(device='cpu').manual_seed(M + N + K)
        A = torch.randn(nb, M, K, generator=g, device='cpu').to(backend.dev)
        B = (torch.randn(1 if shared else nb, N, K, generator=g, device='cpu') * 0.3).to(backend.dev)
        Am = A if akc else A.transpose(1, 2).contiguous()
        Bm = B if bkc else B.transpose(1, 2).contiguous()
        a_str = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
        b_str = (0, 0 if shared else N * K, K, 1) if bkc else (0, 0 if shared else N * K, 1, N)
        out = []
        for pre in (False, True):
            C = torch.full((nb, M, N), float('nan'), device=backend.dev)
            ws = torch.empty(sk * nb * M * N, device=backend.dev) if sk > 1 else None
            planes = L.x6_presplit(Bm, N, K, b_str[2], b_str[3], nb=(1, nb), s_b=(b_str[0], b_str[1])) if pre else None
            L.x6_launches()
            L.gemm(Am, Bm, C, M, N, K, a_str, b_str, (0, M * N, N), nb=(1, nb), splitk=sk, workspace=ws, tile=tile, alpha=0.5, b_planes=planes)
            assert L.x6_launches() == 1
            out.append(C)
    finally:
        L.c.segx_tune(9, 256)
        L.set_engine(prev)
    ref = 0.5 * _ref(A, B.expand(nb, N, K))
    assert (out[0].double() - ref).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize('tile', [segx.TILE_256x128, segx.TILE_WS128x256])
@pytest.mark.parametrize('M,N,K,akc,bkc,sk,nb', [(300, 392, 128, True, True, 1, 2), (264, 520, 96, False, True, 2, 2), (260, 136, 192, True, False, 1, 2),
                                                 (520, 264, 64, False, False, 3, 2)])
def test_x6ws_matches_fp64_with_wide_ranging_rows(backend, tile, M, N, K, akc, bkc, sk, nb):
    """The wave-specialised bf16x6 kernels on rows of very different magnitude (1e-9 .. 1e+6), all four layouts, ragged edges, batches, split-K,
    alpha and bias: error against fp64 at fp32-rounding level, measured per output element against sum |a||b| (a global bound would hide a small
    row computed badly) -- the three-way bf16 split is elementwise fp32-equivalent whatever the operand range."""
    L = backend.L
    prev = L.set_engine('x6')
    try:
        assert L.c.segx_tune(9, 8) == 0
        g = torch.Generator(device='cpu').manual_seed(M + 2 * N + K)
        A = torch.randn(nb, M, K, generator=g, device='cpu') * torch.logspace(-9, 6, M, device='cpu')[None, :, None]
        B = torch.randn(nb, N, K, generator=g, device='cpu') * torch.logspace(3, -6, N, device='cpu')[None, :, None]
        bias = torch.randn(N, generator=g, device='cpu').to(backend.dev)
        Ad, Bd = A.to(backend.dev), B.to(backend.dev)
        Am = Ad if akc else Ad.transpose(1, 2).contiguous()
        Bm = Bd if bkc else Bd.transpose(1, 2).contiguous()
        a_str = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
        b_str = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
        C = torch.full((nb, M, N), float('nan'), device=backend.dev)
        ws = torch.empty(sk * nb * M * N, device=backend.dev) if sk > 1 else None
        L.x6_launches()
        L.gemm(Am, Bm, C, M, N, K, a_str, b_str, (0, M * N, N), nb=(1, nb), splitk=sk, workspace=ws, tile=tile, alpha=0.5, bias=bias, bias_mode=segx.BIAS_N)
        assert L.x6_launches() == 1
        C = C.cpu().double()
    finally:
        L.c.segx_tune(9, 256)
        L.set_engine(prev)
    ref = 0.5 * _ref(A, B) + bias.cpu().double()[None, None, :]
    mag = 0.5 * (A.double().abs() @ B.double().abs().transpose(-1, -2)) + bias.cpu().double().abs()[None, None, :]     # sum |a||b|: the scale of the rounding
    e6 = ((C - ref).abs() / mag).max().item()
    assert e6 < 1e-6, e6                                # rounding noise of the fp32 accumulation (K / 2 * 2^-24 worst case)


def test_x6_wave_specialised_gelu_epilogue(backend):
    L = backend.L
    dev = backend.dev
    prev = L.set_engine('x6')
    try:
        Mo, R, F = 2, 300, 136
        g = torch.Generator(device='cpu').manual_seed(6)
        H = torch.randn(Mo, R, F, generator=g, device='cpu').to(dev); W = (torch.randn(Mo, F, F, generator=g, device='cpu') * 0.3).to(dev)
        b = torch.randn(Mo, F, generator=g, device='cpu').to(dev)
        outs = []
        for tile in (segx.TILE_128x128, segx.TILE_256x128, segx.TILE_WS128x128):
            Y = torch.zeros(Mo, R, F, device=dev); T = torch.zeros(Mo, R, F, device=dev)
            L.gemm(H, W, Y, R, F, F, (0, R * F, F, 1), (0, F * F, F, 1), (0, R * F, F), nb=(1, Mo), bias=b, bias_mode=segx.BIAS_N, bias_b1=F,
                   epilogue=segx.EPI_GELU, aux=T, dropout_p=0.25, seed=3, offset=8, tile=tile)
            outs.append((Y, T))
        Tref = torch.einsum('mrf,mgf->mrg', H.double(), W.double()) + b[:, None, :].double()
        assert (outs[0][1].double() - Tref).abs().max().item() < (1e-5 if backend.name == 'hip' else 4e-5)    # the emulator's bf16 MFMA is a sequential fmaf chain
        for Y, T in outs[1:]:
            assert torch.equal(T, outs[0][1]) and torch.equal(Y, outs[0][0])
    finally:
        L.set_engine(prev)


@pytest.mark.parametrize('default,eng', [('x6', 'f32'), ('f32', 'x6')])
def test_planned_gemm_honours_the_per_call_engine_under_the_opposite_default(backend, default, eng):
    """splitk = 0 (the library plans tile + split factor, what every model-level call does) with desc.engine set against the process default:
    the plan must be made for the engine of THIS call (it was made for the default one, so an 'f32' call under an x6 default could be
    handed a wave-specialised tile and fail, and an 'x6' call under an f32 default silently stayed on the fp32 engine)."""
    L = backend.L
    dev = backend.dev
    prev = L.set_engine(default)
    try:
        g = torch.Generator(device='cpu').manual_seed(11)
        M, N, K = 512, 384, 512                              # large enough for the planner to pick a wave-specialised tile on the bf16x6 engine
        A = torch.randn(M, K, generator=g, device='cpu').to(dev); B = torch.randn(N, K, generator=g, device='cpu').to(dev)
        C = torch.empty(M, N, device=dev)
        L.x6_launches()
        L.gemm(A, B, C, M, N, K, (0, 0, K, 1), (0, 0, K, 1), (0, 0, N), splitk=0, engine=eng)
        assert L.x6_launches() == (1 if eng == 'x6' else 0)
        ref = _ref(A.cpu()[None], B.cpu()[None])[0]
        assert (C.cpu().double() - ref).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())
    finally:
        L.set_engine(prev)


@pytest.mark.gpu
def test_gemm_is_reentrant_across_threads_streams_and_engines():
    """include/segx.h conventions: two host threads, each on its own stream, call segx_gemm_f32 concurrently -- one with
    desc.engine = f32, one with desc.engine = bf16x6 -- while the process default stays untouched; every result equals the serial one."""
    import threading
    from segtran_amd import segx as sx
    L = sx.lib()
    dev = torch.device('cuda', 0)
    default = L.set_engine('f32'); L.set_engine(default)
    g = torch.Generator(device='cpu').manual_seed(3)
    M, N, K = 512, 384, 256
    A = torch.randn(M, K, generator=g).to(dev); B = torch.randn(N, K, generator=g).to(dev)
    ref = {}
    for eng in ('f32', 'x6'):
        C = torch.empty(M, N, device=dev)
        L.gemm(A, B, C, M, N, K, (0, 0, K, 1), (0, 0, K, 1), (0, 0, N), engine=eng)
        ref[eng] = C.clone()
    torch.cuda.synchronize()
    assert not torch.equal(ref['f32'], ref['x6'])           # different engines, different fp32 rounding: the selector is honoured
    errs = []

    def work(eng):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(50):
                    C = torch.empty(M, N, device=dev)
                    L.gemm(A, B, C, M, N, K, (0, 0, K, 1), (0, 0, K, 1), (0, 0, N), engine=eng)
                    if not torch.equal(C, ref[eng]):
                        errs.append(eng)
            st.synchronize()
        except Exception as e:                               # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(e,)) for e in ('f32', 'x6', 'x6', 'f32')]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs[:3]
    prev = L.set_engine(default)
    assert prev == default                                   # no call changed the process default


@pytest.mark.parametrize('engine', ['f32', 'x6'])
@pytest.mark.parametrize('M,N,K,nb,sk,tile', [(130, 72, 64, 2, 1, segx.TILE_AUTO), (96, 200, 40, 1, 1, segx.TILE_AUTO), (64, 260, 96, 3, 2, segx.TILE_128x128),
                                               (300, 392, 128, 2, 1, segx.TILE_256x128), (33, 70, 17, 2, 1, segx.TILE_AUTO)])
def test_gemm_residual_epilogue(backend, engine, M, N, K, nb, sk, tile):
    """segx_gemm_desc.resid: C = alpha A B^T + bias + resid in one launch (with split-K: in the slab reduction) -- what carries the skip connection's
    gradient into the expansion convolution's dX GEMM.  16-byte and 4-byte store paths (N % 4), ragged edges, batches, the wave-specialised tile."""
    L = backend.L
    if tile == segx.TILE_256x128 and engine != 'x6':
        pytest.skip('wave-specialised tiles exist on the bf16x6 engine only')
    prev = L.set_engine(engine)
    try:
        g = torch.Generator(device='cpu').manual_seed(M + N + K)
        A = torch.randn(nb, M, K, generator=g, device='cpu'); B = torch.randn(nb, N, K, generator=g, device='cpu')
        R = torch.randn(nb, M, N, generator=g, device='cpu'); bias = torch.randn(N, generator=g, device='cpu')
        Ad, Bd, Rd, bd = (t.to(backend.dev) for t in (A, B, R, bias))
        C = torch.full((nb, M, N), float('nan'), device=backend.dev)
        ws = torch.empty(sk * nb * M * N, device=backend.dev) if sk > 1 else None
        L.gemm(Ad, Bd, C, M, N, K, (0, M * K, K, 1), (0, N * K, K, 1), (0, M * N, N), nb=(1, nb), splitk=sk, workspace=ws, tile=tile, alpha=0.5, bias=bd,
               bias_mode=segx.BIAS_N, resid=Rd)
        ref = 0.5 * _ref(A, B) + bias.double()[None, None, :] + R.double()
        assert (C.cpu().double() - ref).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())
    finally:
        L.set_engine(prev)


def test_second_consumer_gradient_rides_on_the_dx_gemm(backend):
    """conv1x1(pass_input=True): the alias output carries a second use of the input (the MBConv skip connection); both gradients reach the GEMM node
    and are summed inside its dX launch -- equal to plain autograd accumulation."""
    from segtran_amd import functional as SF
    dev = backend.dev
    g = torch.Generator(device='cpu').manual_seed(9)
    x0 = torch.randn(2, 8, 6, 10, generator=g, device='cpu').to(dev)
    W = torch.randn(12, 8, 1, 1, generator=g, device='cpu').to(dev).requires_grad_(True)
    G1 = torch.randn(2, 12, 6, 10, generator=g, device='cpu').to(dev); G2 = torch.randn(2, 8, 6, 10, generator=g, device='cpu').to(dev)
    outs = []
    for passed in (True, False):
        x = x0.clone().requires_grad_(True)
        h = x * 1.0                                           # a non-leaf input, as inside the network
        if passed:
            y, h2 = SF.conv1x1(h, W, pass_input=True)
            assert h2.data_ptr() == h.data_ptr()
        else:
            y, h2 = SF.conv1x1(h, W), h
        W.grad = None
        ((y * G1).sum() + (h2 * G2).sum()).backward()
        outs.append((x.grad.clone(), W.grad.clone()))
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-5) and torch.allclose(outs[0][1], outs[1][1], atol=1e-5)
    x = x0.clone().requires_grad_(True)                      # alias unused: no residual, plain gradient
    y, _ = SF.conv1x1(x * 1.0, W, pass_input=True)
    (y * G1).sum().backward()
    assert torch.allclose(x.grad, outs[0][0] - G2, atol=1e-5)
