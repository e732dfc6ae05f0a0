"""CPU: segtran_amd/csrc/tokens.hip on the fiber emulator vs plain PyTorch fp32 (autograd) references."""
import pytest
import torch
import torch.nn.functional as F

EPS = 1e-12


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g, device='cpu') * scale).to(torch.get_default_device())


def close(a, b, tol=2e-5):
    s = max(b.abs().max().item(), 1e-20)
    err = (a - b).abs().max().item()
    assert err <= tol * s, 'err %.3e scale %.3e' % (err, s)


@pytest.mark.parametrize('rows,L', [(7, 64), (5, 260), (9, 1024), (2, 4096), (6, 121), (3, 5184), (4, 1)])      # last three: generic-width kernels
@pytest.mark.parametrize('clamped', [False, True])
def test_softmax_fwd_bwd(backend, rows, L, clamped):
    Lb = backend.L
    S = rnd(rows, L, seed=1, scale=3.0)
    clip = 4.0
    gmax = torch.tensor([S.max().item() if clamped else 1.0])
    if clamped and L > 1:
        assert S.max() > clip and S.min() < -clip
    P = torch.empty_like(S)
    Lb.softmax_fwd(S, P, None, rows, L, clip, gmax, 0.0, 0, 0)
    Sr = S.clone().requires_grad_(True)
    Pr = (Sr.clamp(-clip, clip) if clamped else Sr).softmax(-1)
    close(P, Pr.detach(), 1e-5)
    G = rnd(rows, L, seed=2)
    Pr.backward(G)
    dS = torch.empty_like(S)
    Lb.softmax_bwd(P, G, S, dS, rows, L, clip, gmax, 0.0, 0, 0)
    close(dS, Sr.grad, 2e-5)


@pytest.mark.parametrize('L', [256, 121])
def test_softmax_dropout_consistent_fwd_bwd(backend, L):
    Lb = backend.L
    rows, p = 6, 0.25
    S = rnd(rows, L, seed=3)
    P = torch.empty_like(S); Pd = torch.empty_like(S)
    Lb.softmax_fwd(S, P, Pd, rows, L, 500.0, None, p, 77, 1024)
    keep = Pd != 0
    assert 0.15 < 1 - keep.float().mean().item() < 0.35
    assert torch.allclose(Pd[keep], P[keep] / (1 - p), rtol=1e-6)
    mask = keep.float() / (1 - p)
    Sr = S.clone().requires_grad_(True)
    G = rnd(rows, L, seed=4)
    (Sr.softmax(-1) * mask).backward(G)
    dS = torch.empty_like(S)
    Lb.softmax_bwd(P, G, None, dS, rows, L, 500.0, None, p, 77, 1024)
    close(dS, Sr.grad, 2e-5)


@pytest.mark.parametrize('rows,C', [(10, 64), (5, 448), (3, 1792), (4, 1024)])
@pytest.mark.parametrize('affine', [True, False])
def test_layernorm_fwd_bwd_param_grads(backend, rows, C, affine):
    Lb = backend.L
    X = rnd(rows, C, seed=5) * 2 + 0.5
    w = (1 + 0.1 * rnd(C, seed=6)) if affine else None
    b = (0.1 * rnd(C, seed=7)) if affine else None
    Y = torch.empty_like(X); mean = torch.empty(rows); rstd = torch.empty(rows)
    Lb.layernorm_fwd(X, w, b, Y, mean, rstd, rows, C, EPS)
    Xr = X.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True) if affine else None
    br = b.clone().requires_grad_(True) if affine else None
    Yr = F.layer_norm(Xr, (C,), wr, br, EPS)
    close(Y, Yr.detach(), 1e-5)
    G = rnd(rows, C, seed=8)
    Yr.backward(G)
    dX = torch.empty_like(X)
    Lb.layernorm_bwd(G, X, w, mean, rstd, dX, rows, C)
    close(dX, Xr.grad, 3e-5)
    if affine:
        dw = torch.empty(C); db = torch.empty(C); ws = torch.empty(Lb.colreduce_ws(rows, C, 2))
        Lb.ln_param_grad(G, X, mean, rstd, dw, db, ws, rows, C)
        close(dw, wr.grad, 2e-5); close(db, br.grad, 2e-5)


@pytest.mark.parametrize('rows', [40, 300])
def test_wide_column_reductions_take_the_four_column_form(backend, rows):
    """C >= 1024, C % 4 == 0: colreduce_stage1_v4 (16-byte loads, four columns per thread) for the plain column sum (one-pass form at <= 64 rows, two stages
    above) and for the LayerNorm parameter gradients; a column's partial sums are formed in the same order as in the one-column form, so a width just
    below the switch (which takes the one-column kernel) must give bit-identical columns."""
    Lb = backend.L
    C = 1028
    X = rnd(rows, C, seed=70); dY = rnd(rows, C, seed=71)
    out = torch.empty(C); ws = torch.empty(Lb.colreduce_ws(rows, C, 1))
    Lb.colsum(X, out, ws, rows, C)
    close(out, X.double().sum(0).float(), 1e-5)
    Xn = X[:, :1020].contiguous(); outn = torch.empty(1020); wsn = torch.empty(Lb.colreduce_ws(rows, 1020, 1))
    Lb.colsum(Xn, outn, wsn, rows, 1020)                                   # 1020 < 1024: the one-column kernel
    assert torch.equal(out[:1020].cpu(), outn.cpu())
    mean = X.mean(1).contiguous(); rstd = (1.0 / (X.var(1, unbiased=False) + 1e-5).sqrt()).contiguous()
    dw = torch.empty(C); db = torch.empty(C); ws2 = torch.empty(Lb.colreduce_ws(rows, C, 2))
    Lb.ln_param_grad(dY, X, mean, rstd, dw, db, ws2, rows, C)
    xhat = (X - mean[:, None]) * rstd[:, None]
    close(dw, (dY.double() * xhat.double()).sum(0).float(), 1e-5)
    close(db, dY.double().sum(0).float(), 1e-5)


def test_colsum_and_sum(backend):
    Lb = backend.L
    X = rnd(1000, 70, seed=9)
    out = torch.empty(70); ws = torch.empty(Lb.colreduce_ws(1000, 70, 1))
    Lb.colsum(X, out, ws, 1000, 70)
    close(out, X.double().sum(0).float(), 1e-5)
    o = torch.empty(1); ws = torch.empty(1024)
    Lb.sum(X, X.numel(), o, ws, 0.5)
    assert abs(o.item() - 0.5 * X.double().sum().item()) < 1e-2


def _prenorm_ref(X, w1, b1, pos, pw, mask, C, keep=None):
    u = F.layer_norm(X, (C,), w1, b1, EPS) + pw * pos[None, :, :C]
    y = F.layer_norm(u, (C,), None, None, EPS)
    if keep is not None:
        y = y * keep
    return y * mask[:, :, None]


@pytest.mark.parametrize('B,N,C,Cpos', [(2, 12, 64, 64), (2, 9, 448, 896), (1, 5, 1792, 1792)])
def test_prenorm_fwd_bwd(backend, B, N, C, Cpos):
    Lb = backend.L
    X = rnd(B, N, C, seed=10) + 0.3
    w1 = 1 + 0.1 * rnd(C, seed=11); b1 = 0.1 * rnd(C, seed=12)
    pos = rnd(N, Cpos, seed=13); pw = 0.7
    mask = (torch.rand(B, N, generator=torch.Generator(device='cpu').manual_seed(14), device='cpu').to(backend.dev) > 0.3).float()
    Y = torch.empty_like(X); stats = torch.empty(4 * B * N)
    Lb.prenorm_fwd(X, w1, b1, pos, Cpos, pw, mask, Y, stats, B, N, C, EPS, 0.0, 0, 0)
    Xr, w1r, b1r, posr = (t.clone().requires_grad_(True) for t in (X, w1, b1, pos))
    Yr = _prenorm_ref(Xr, w1r, b1r, posr, pw, mask, C)
    close(Y, Yr.detach(), 1e-5)
    G = rnd(B, N, C, seed=15)
    Yr.backward(G)
    dX = torch.empty_like(X); dU = torch.empty_like(X)
    Lb.prenorm_bwd(G, X, w1, b1, pos, Cpos, pw, mask, stats, dX, dU, B, N, C, 0.0, 0, 0)
    close(dX, Xr.grad, 5e-5)
    # parameter grads from dU via the column reductions
    dw = torch.empty(C); db = torch.empty(C); ws = torch.empty(Lb.colreduce_ws(B * N, C, 2))
    Lb.ln_param_grad(dU, X, stats[:B * N], stats[B * N:2 * B * N], dw, db, ws, B * N, C)
    close(dw, w1r.grad, 5e-5); close(db, b1r.grad, 5e-5)
    dpos = torch.empty(N * C); ws = torch.empty(Lb.colreduce_ws(B, N * C, 1))
    Lb.colsum(dU, dpos, ws, B, N * C)
    close(pw * dpos.view(N, C), posr.grad[:, :C], 5e-5)
    # the one-pass form (segx_prenorm_bwd_all): dX bit for bit, the sums that were taken from dU formed inside the kernel
    dX2 = torch.full_like(X, float('nan')); ds2 = torch.full((N, C), float('nan')); dw2 = torch.full((C,), float('nan')); db2 = torch.full((C,), float('nan'))
    Lb.prenorm_bwd_all(G, X, w1, b1, pos, Cpos, pw, mask, stats, dX2, ds2, dw2, db2, torch.full((Lb.prenorm_bwd_all_ws(N, C),), float('nan')), B, N, C, 0.0, 0, 0)
    assert torch.equal(dX2, dX)
    close(dw2, w1r.grad, 5e-5); close(db2, b1r.grad, 5e-5); close(pw * ds2, posr.grad[:, :C], 5e-5)


def test_prenorm_dropout_mask_matches_between_fwd_and_bwd(backend):
    Lb = backend.L
    B, N, C, p = 2, 6, 64, 0.2
    X = rnd(B, N, C, seed=16); w1 = 1 + 0.1 * rnd(C, seed=17); b1 = 0.1 * rnd(C, seed=18)
    pos = rnd(N, C, seed=19); mask = torch.ones(B, N)
    Y = torch.empty_like(X); stats = torch.empty(4 * B * N)
    Lb.prenorm_fwd(X, w1, b1, pos, C, 1.0, mask, Y, stats, B, N, C, EPS, p, 5, 64)
    full = _prenorm_ref(X, w1, b1, pos, 1.0, mask, C)
    keep = (Y != 0).float() / (1 - p)
    close(Y, full * keep, 1e-5)
    Xr = X.clone().requires_grad_(True)
    G = rnd(B, N, C, seed=20)
    _prenorm_ref(Xr, w1, b1, pos, 1.0, mask, C, keep).backward(G)
    dX = torch.empty_like(X); dU = torch.empty_like(X)
    Lb.prenorm_bwd(G, X, w1, b1, pos, C, 1.0, mask, stats, dX, dU, B, N, C, p, 5, 64)
    close(dX, Xr.grad, 5e-5)
    dX2 = torch.full_like(X, float('nan')); ds2 = torch.empty(N, C); dw2 = torch.empty(C); db2 = torch.empty(C)
    Lb.prenorm_bwd_all(G, X, w1, b1, pos, C, 1.0, mask, stats, dX2, ds2, dw2, db2, torch.empty(Lb.prenorm_bwd_all_ws(N, C)), B, N, C, p, 5, 64)
    assert torch.equal(dX2, dX)
    close(ds2, dU.sum(0), 5e-5)


@pytest.mark.parametrize('N,C,pd', [(20, 64, 2), (7, 1792, 2), (11, 1024, 3)])
def test_posembed_fwd_bwd(backend, N, C, pd):
    Lb = backend.L
    posn = torch.rand(N, pd, generator=torch.Generator(device='cpu').manual_seed(21), device='cpu').to(backend.dev)
    Wp = rnd(C, pd, seed=22); bp = 0.1 * rnd(C, seed=23)
    out = torch.empty(N, C); stats = torch.empty(2 * N)
    Lb.posembed_fwd(posn, Wp, bp, out, stats, N, C, pd, EPS)
    Wr = Wp.clone().requires_grad_(True); br = bp.clone().requires_grad_(True)
    z = posn @ Wr.t() + br
    mix = torch.stack((torch.sin(z[:, 0::2]), torch.cos(z[:, 1::2])), dim=2).view(N, C)
    ref = F.layer_norm(mix, (C,), None, None, EPS)
    close(out, ref.detach(), 1e-5)
    G = rnd(N, C, seed=24)
    ref.backward(G)
    dZ = torch.empty(N, C)
    Lb.posembed_bwd(G, posn, Wp, bp, stats, dZ, N, C, pd)
    close(dZ.t() @ posn, Wr.grad, 5e-5)
    close(dZ.sum(0), br.grad, 5e-5)


def _aggr_ref(Z, lnw, lnb, wa, ba, keep=None):
    zd = Z if keep is None else Z * keep
    zn = F.layer_norm(zd, (Z.shape[-1],), lnw, lnb, EPS)
    sc = zn @ wa + ba                                           # [Mo, R]
    return (zn * sc.softmax(0)[..., None]).sum(0)


@pytest.mark.parametrize('Mo,R,Fd', [(4, 9, 64), (4, 5, 448), (4, 3, 1792), (1, 6, 1024), (4, 4, 32)])
@pytest.mark.parametrize('p', [0.0, 0.2])
def test_modes_aggr_fwd_bwd_param_grads(backend, Mo, R, Fd, p):
    Lb = backend.L
    Z = rnd(Mo, R, Fd, seed=25) * 1.5 + 0.2
    lnw = 1 + 0.1 * rnd(Fd, seed=26); lnb = 0.1 * rnd(Fd, seed=27)
    wa = 0.2 * rnd(Fd, seed=28); ba = torch.tensor([0.3])
    Y = torch.empty(R, Fd); stats = torch.empty(3 * Mo * R)
    seed, off = 9, 4096
    Lb.modes_aggr_fwd(Z, lnw, lnb, wa, ba, Y, stats, Mo, R, Fd, EPS, p, seed, off)
    keep = None
    if p > 0:
        # recover the mask through the GELU-bwd kernel, which shares the (seed, offset, flat index) stream
        ones = torch.ones(Mo * R * Fd); big = torch.full((Mo * R * Fd,), 30.0); k = torch.empty(Mo * R * Fd)
        Lb.gelu_bwd(ones, big, k, Mo * R * Fd, p, seed, off)          # gelu'(30) == 1
        keep = k.view(Mo, R, Fd)
        assert 0.1 < (keep == 0).float().mean().item() < 0.3
    Zr, lnwr, lnbr, war, bar = (t.clone().requires_grad_(True) for t in (Z, lnw, lnb, wa, ba))
    Yr = _aggr_ref(Zr, lnwr, lnbr, war, bar, keep)
    close(Y, Yr.detach(), 2e-5)
    G = rnd(R, Fd, seed=29)
    Yr.backward(G)
    dZ = torch.empty_like(Z); dscore = torch.empty(Mo * R)
    Lb.modes_aggr_bwd(G, Z, lnw, lnb, wa, stats, dZ, dscore, Mo, R, Fd, p, seed, off)
    close(dZ, Zr.grad, 1e-4)
    dlnw = torch.empty(Fd); dlnb = torch.empty(Fd); dwa = torch.empty(Fd)
    ws = torch.empty(Lb.colreduce_ws(R, Fd, 3))
    Lb.modes_aggr_param_grad(G, Z, lnw, lnb, wa, stats, dscore, dlnw, dlnb, dwa, ws, Mo, R, Fd, p, seed, off)
    close(dlnw, lnwr.grad, 1e-4); close(dlnb, lnbr.grad, 1e-4)
    if Mo > 1:
        close(dwa, war.grad, 1e-4)
    else:
        assert dwa.abs().max() < 1e-6 and dscore.abs().max() < 1e-6      # single mode: softmax == 1, zero grads
    # the one-pass form (segx_modes_aggr_bwd_all: parameter-gradient terms collected inside the dZ kernel; one mode runs the two passes above)
    dZ2 = torch.full_like(Z, float('nan')); ds2 = torch.full((Mo * R,), float('nan'))
    g3 = [torch.full((Fd,), float('nan')) for _ in range(3)]
    ws2 = torch.full((Lb.modes_aggr_bwd_all_ws(Mo, R, Fd),), float('nan'))
    Lb.modes_aggr_bwd_all(G, Z, lnw, lnb, wa, stats, dZ2, ds2, g3[0], g3[1], g3[2], ws2, Mo, R, Fd, p, seed, off)
    assert torch.equal(dZ2, dZ) and torch.equal(ds2, dscore)
    close(g3[0], lnwr.grad, 1e-4); close(g3[1], lnbr.grad, 1e-4); close(g3[2], dwa, 1e-4 if Mo > 1 else 1.0)


@pytest.mark.parametrize('R,Fd,lnorm', [(4096 + 6, 64, True), (16384 + 3, 32, True), (4099, 1792, False)])
def test_modes_aggr_one_pass_backward_walks_several_tokens_per_wave(backend, R, Fd, lnorm):
    """segx_modes_aggr_bwd_all with 2 / 4 tokens per wave (R >= 4096 / 16384), a ragged last workgroup, with and without the LayerNorm (lnw = NULL: the no-FFN
    branch aggregates the raw mode features) -- against the two-pass form it replaces (same dZ bit for bit, the same column sums in another order)."""
    Lb = backend.L
    if backend.name == 'emu' and R * Fd > 300000:
        pytest.skip('emulator: the long rows are covered by the short token list above')
    Mo, p, seed, off = 4, 0.2, 5, 1024
    Z = rnd(Mo, R, Fd, seed=41) * 1.5 + 0.2
    lnw = (1 + 0.1 * rnd(Fd, seed=42)) if lnorm else None; lnb = 0.1 * rnd(Fd, seed=43) if lnorm else None
    wa = 0.2 * rnd(Fd, seed=44); ba = torch.tensor([0.3])
    Y = torch.empty(R, Fd); stats = torch.empty(3 * Mo * R)
    Lb.modes_aggr_fwd(Z, lnw, lnb, wa, ba, Y, stats, Mo, R, Fd, EPS, p, seed, off)
    G = rnd(R, Fd, seed=45)
    dZ = torch.empty_like(Z); dscore = torch.empty(Mo * R)
    Lb.modes_aggr_bwd(G, Z, lnw, lnb, wa, stats, dZ, dscore, Mo, R, Fd, p, seed, off)
    ref = [torch.empty(Fd) for _ in range(3)]
    Lb.modes_aggr_param_grad(G, Z, lnw, lnb, wa, stats, dscore, ref[0], ref[1], ref[2], torch.empty(Lb.colreduce_ws(R, Fd, 3)), Mo, R, Fd, p, seed, off)
    dZ2 = torch.full_like(Z, float('nan')); ds2 = torch.full((Mo * R,), float('nan'))
    got = [torch.full((Fd,), float('nan')) for _ in range(3)]
    Lb.modes_aggr_bwd_all(G, Z, lnw, lnb, wa, stats, dZ2, ds2, got[0], got[1], got[2], torch.full((Lb.modes_aggr_bwd_all_ws(Mo, R, Fd),), float('nan')), Mo, R, Fd, p, seed, off)
    assert torch.equal(dZ2, dZ) and torch.equal(ds2, dscore)
    for a, b in zip(got, ref):
        close(a, b, 2e-4)


def test_gelu_bwd(backend):
    Lb = backend.L
    T = rnd(50, 36, seed=30) * 2
    G = rnd(50, 36, seed=31)
    Tr = T.clone().requires_grad_(True)
    F.gelu(Tr).backward(G)
    dT = torch.empty_like(T)
    Lb.gelu_bwd(G, T, dT, T.numel(), 0.0, 0, 0)
    close(dT, Tr.grad, 1e-5)


@pytest.mark.parametrize('rows,N,p', [(50, 36, 0.0), (2051, 64, 0.3), (300, 1792, 0.2)])
def test_gelu_bwd_with_the_bias_gradient_from_the_same_pass(backend, rows, N, p):
    """segx_gelu_bwd_colsum: dT bit-identical to segx_gelu_bwd (same mask stream), its column sums = the bias gradient of the linear layer in front of the GELU."""
    Lb = backend.L
    T = rnd(rows, N, seed=34) * 2; G = rnd(rows, N, seed=35)
    ref = torch.empty_like(T)
    Lb.gelu_bwd(G, T, ref, T.numel(), p, 7, 64)
    dT = torch.full_like(T, float('nan')); cs = torch.full((N,), float('nan'))
    Lb.gelu_bwd_colsum(G, T, dT, cs, torch.full((Lb.gelu_bwd_colsum_ws(rows, N),), float('nan')), rows, N, p, 7, 64)
    assert torch.equal(dT, ref)
    close(cs, ref.double().sum(0).float(), 2e-5)


@pytest.mark.parametrize('engine', ['f32', 'x6'])
@pytest.mark.parametrize('Fd,offset', [(36, 8), (36, 12), (35, 8), (64, 4096)])
def test_gemm_gelu_dropout_mask_equals_gelu_bwd_mask(backend, engine, Fd, offset):
    """The GEMM epilogue (scattered MFMA lanes) and gelu_bwd (float4 lanes) must regenerate the same mask: the epilogue's quad form (one Philox
    counter per 4 lanes, stream position and row length multiples of 4) and its per-lane form (row length not a multiple of 4) alike."""
    from segtran_amd import segx
    Lb = backend.L
    R, p = 72, 0.3
    H = rnd(R, 64, seed=32); W = rnd(Fd, 64, seed=33) * 0.3
    Y = torch.zeros(R, Fd); T = torch.zeros(R, Fd)
    Lb.set_engine(engine)
    try:
        Lb.gemm(H, W, Y, R, Fd, 64, (0, 0, 64, 1), (0, 0, 64, 1), (0, 0, Fd), epilogue=segx.EPI_GELU, aux=T, dropout_p=p, seed=3, offset=offset)
    finally:
        Lb.set_engine(segx.DEFAULT_ENGINE)
    k = torch.empty(R * Fd)
    Lb.gelu_bwd(torch.ones(R * Fd), torch.full((R * Fd,), 30.0), k, R * Fd, p, 3, offset)
    close(Y, F.gelu(T) * k.view(R, Fd), 1e-5)
    assert 0.2 < (k == 0).float().mean().item() < 0.4


@pytest.mark.parametrize('shape,R', [((5, 6), 2), ((3, 4, 2), 1), ((4, 4), 7)])
@pytest.mark.parametrize('clamped', [False, True])
def test_sliding_pos_bias_add(backend, shape, R, clamped):
    """K14: scores + w * bias[N,N] computed on the fly from the (2R+1)^d table, vs the oracle's materialised lookup."""
    from oracle import segtran_oracle as O
    from segtran_amd import functional as SF
    N = 1
    for s_ in shape:
        N *= s_
    nd = len(shape)
    table = rnd(*([2 * R + 1] * nd), seed=40)
    S = rnd(3, 2, N, N, seed=41, scale=3.0)
    clip, w = 4.0, 0.7
    gmax = torch.tensor([S.max().item() if clamped else 1.0])
    tr = table.detach().clone().cpu().requires_grad_(True)
    Sr = S.detach().clone().cpu().requires_grad_(True)
    with torch.device('cpu'):                                  # the oracle is CPU code; the hip backend sets a cuda default device
        ref = (Sr.clamp(-clip, clip) if clamped else Sr) + w * O.sliding_pos_biases(tr, shape)
    t2 = table.clone().requires_grad_(True); S2 = S.clone().requires_grad_(True)
    out = SF.pos_bias_add(S2, t2, shape, w, clip, gmax)
    close(out.cpu(), ref.detach(), 1e-6)
    G = rnd(3, 2, N, N, seed=42)
    out.backward(G); ref.backward(G.cpu())
    close(S2.grad.cpu(), Sr.grad, 1e-6)
    close(t2.grad.cpu(), tr.grad, 1e-5)
