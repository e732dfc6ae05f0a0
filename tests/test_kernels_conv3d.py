"""conv3d.hip: implicit-GEMM 3-D convolution (fwd / bwd-data / bwd-weight) and 'same' max-pool vs PyTorch."""
import pytest
import torch
import torch.nn.functional as F
from segtran_amd import functional as SF


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g, device='cpu') * scale).to(torch.get_default_device())


def close(a, b, tol=5e-5):
    s = max(b.abs().max().item(), 1e-20)
    err = (a - b).abs().max().item()
    assert err <= tol * s, 'err %.3e scale %.3e' % (err, s)


def _ref_conv(x, w, stride):
    pads = SF._same_pads(x.shape[2:], w.shape[2:], stride)
    xp = F.pad(x, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    return F.conv3d(xp, w, None, stride)


@pytest.mark.parametrize('B,Cin,Cout,size,k,stride', [(2, 8, 12, (4, 6, 5), (3, 3, 3), (1, 1, 1)), (1, 3, 8, (8, 10, 12), (7, 7, 7), (2, 2, 2)),
                                                      (1, 16, 140, (3, 9, 9), (3, 3, 3), (1, 1, 1)), (2, 4, 6, (5, 7, 7), (1, 3, 3), (1, 1, 1)),
                                                      (2, 24, 16, (4, 5, 6), (3, 3, 3), (1, 1, 1)),     # packed K order forward AND backward-data
                                                      (1, 8, 8, (6, 7, 8), (7, 7, 7), (1, 1, 1))])      # packed + per-axis masks (343 taps)
def test_conv3d_same(backend, B, Cin, Cout, size, k, stride):
    x = rnd(B, Cin, *size, seed=1).requires_grad_(True)
    w = (rnd(Cout, Cin, *k, seed=2) * 0.2).requires_grad_(True)
    y = SF.conv3d_same(x, w, stride)
    xr, wr = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    yr = _ref_conv(xr, wr, stride)
    assert y.shape == yr.shape
    close(y, yr.detach())
    G = rnd(*y.shape, seed=3)
    y.backward(G); yr.backward(G)
    close(x.grad, xr.grad, 1e-4); close(w.grad, wr.grad, 1e-4)


@pytest.mark.parametrize('size,k,stride', [((4, 9, 10), (1, 3, 3), (1, 2, 2)), ((6, 8, 8), (3, 3, 3), (2, 2, 2)), ((4, 7, 7), (2, 2, 2), (2, 2, 2)),
                                           ((3, 5, 6), (3, 3, 3), (1, 1, 1)), ((6, 11, 37), (3, 3, 3), (1, 1, 1)),      # several LDS tiles per plane
                                           ((4, 7, 9), (1, 3, 3), (1, 1, 1)), ((5, 9, 10), (3, 3, 3), (3, 3, 3)),      # generic-stride gather
                                           ((3, 6, 12), (1, 3, 3), (1, 2, 2)), ((4, 8, 8), (2, 2, 2), (2, 2, 2)), ((5, 7, 16), (3, 3, 3), (2, 2, 2)),      # four-cell stride-2 gather
                                           ((3, 5, 8), (3, 3, 3), (1, 1, 1)), ((4, 6, 12), (1, 3, 3), (1, 1, 1)), ((2, 3, 4), (3, 3, 3), (1, 1, 1)),      # four-output stride-1 scan
                                           ((6, 16, 40), (3, 3, 3), (1, 1, 1))])      # ... over several waves (neighbour columns across lanes and at wave edges)
def test_maxpool3d_same(backend, size, k, stride):
    x = torch.relu(rnd(2, 3, *size, seed=4)).requires_grad_(True)          # post-ReLU inputs as in I3D (incl. exact zeros)
    y = SF.maxpool3d_same(x, k, stride)
    xr = x.detach().clone().requires_grad_(True)
    pads = SF._same_pads(size, k, stride)
    yr = F.max_pool3d(F.pad(xr, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1])), k, stride)
    assert torch.equal(y, yr.detach())
    G = rnd(*y.shape, seed=5)
    y.backward(G); yr.backward(G)
    close(x.grad, xr.grad, 1e-6)



# (24, 14, 14) / (12, 7, 7): the Mixed_4 / Mixed_5 planes of cfg4 (rows of 14 / 7 floats: whole plane = one slab); (48, 28, 28): Mixed_3 of cfg4 in slabs of
# 8 slices + halo; (32, 16, 16): Mixed_4 of cfg5 = exactly 8192 floats; (20, 32, 32): slabs of 6; (5, 9, 11) / (3, 3, 5): odd sizes whose slabs start off a
# 16-byte boundary (scalar staging); (9, 90, 91): a slice alone fills the LDS budget (8190 of 8192 floats: no slab form, falls through to the other kernels)
@pytest.mark.parametrize('policy', [0, 1, 2])
@pytest.mark.parametrize('size', [(24, 14, 14), (12, 7, 7), (48, 28, 28), (32, 16, 16), (20, 32, 32), (5, 9, 11), (3, 3, 5), (9, 90, 91)])
def test_maxpool3d_stride1_slab_form(backend, size, policy):
    """segx_tune knob 14: the slab-in-LDS form of the stride-1 3 x 3 x 3 'same' pools (0: where the four-cells-per-thread form does not apply, 1: wherever a
    slab fits, 2: never) -- outputs, arg-max routing and gradients identical to F.max_pool3d under every policy, NaN and all-zero windows included."""
    L = backend.L
    if backend.name == 'emu' and size[0] * size[1] * size[2] > 12000 and policy != 1:
        pytest.skip('large planes on the emulator: the slab policy only (the other kernels are covered at small sizes)')
    assert L.c.segx_tune(14, 3) < 0
    assert L.c.segx_tune(14, policy) == 0
    try:
        x = torch.relu(rnd(2, 2, *size, seed=14))
        x[0, 0, :2] = 0.0                                      # whole windows of zeros: the first tap (or the padding) wins
        x[1, 1, size[0] // 2, size[1] // 2, size[2] // 2] = float('nan')
        x = x.requires_grad_(True)
        y = SF.maxpool3d_same(x, (3, 3, 3), (1, 1, 1))
        xr = x.detach().clone().requires_grad_(True)
        yr = F.max_pool3d(F.pad(xr, (1, 1, 1, 1, 1, 1)), 3, 1)
        assert torch.equal(torch.nan_to_num(y, nan=-7.0), torch.nan_to_num(yr.detach(), nan=-7.0))
        G = rnd(*y.shape, seed=15)
        y.backward(G); yr.backward(G)
        if x.is_cuda:
            close(x.grad, xr.grad, 1e-6)                       # ATen's device backward accumulates with atomics: the order of a cell's <= 27 terms is not fixed there
        else:
            assert torch.equal(x.grad, xr.grad)
    finally:
        assert L.c.segx_tune(14, 0) == 0




@pytest.mark.parametrize('size', [(4, 5, 8), (5, 6, 12), (9, 7, 16), (16, 8, 8), (17, 9, 20), (48, 6, 28), (3, 5, 8), (6, 16, 40)])
def test_maxpool3d_stride1_depth_sliding_form(backend, size):
    """segx_tune knob 15: the stride-1 3 x 3 x 3 pools with W % 4 == 0 slide along the depth (a thread keeps its four cells for 2 / 4 / 8 slices and reduces each
    input slice once) -- outputs, arg-max indices and gradients identical, bit for bit, to the one-slice-per-thread form (knob 15 = 0) and equal to
    F.max_pool3d; chunk boundaries (D not a multiple of the chunk), D < 4 (falls back), zero windows and a NaN included."""
    L = backend.L
    x0 = torch.relu(rnd(2, 2, *size, seed=34))
    x0[0, 0, :2] = 0.0
    x0[1, 1, size[0] // 2, size[1] // 2, size[2] // 2] = float('nan')
    x0[1, 0, -1, -1, -1] = float('nan')
    B, C = 2, 2
    D, H, W = size
    geom = (D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1)
    G = rnd(B, C, *size, seed=35)
    outs = {}
    assert L.c.segx_tune(15, 2) < 0
    for knob in (1, 0):
        assert L.c.segx_tune(15, knob) == 0
        try:
            y, arg, dx = torch.empty_like(x0), torch.empty(x0.shape, dtype=torch.int32), torch.empty_like(x0)
            L.maxpool3d_fwd(x0, y, arg, B * C, geom)
            L.maxpool3d_bwd(G, arg, dx, B * C, geom)
            outs[knob] = (y, arg, dx)
        finally:
            assert L.c.segx_tune(15, 1) == 0
    assert torch.equal(torch.nan_to_num(outs[1][0], nan=-7.0), torch.nan_to_num(outs[0][0], nan=-7.0))
    assert torch.equal(outs[1][1], outs[0][1]) and torch.equal(outs[1][2], outs[0][2])
    xr = x0.clone().requires_grad_(True)
    yr = F.max_pool3d(F.pad(xr, (1, 1, 1, 1, 1, 1)), 3, 1)
    assert torch.equal(torch.nan_to_num(outs[1][0], nan=-7.0), torch.nan_to_num(yr.detach(), nan=-7.0))
    yr.backward(G)
    close(outs[1][2], xr.grad, 1e-6)


@pytest.mark.parametrize('size,k,stride', [((4, 8, 8), (1, 3, 3), (1, 2, 2)), ((6, 8, 16), (3, 3, 3), (2, 2, 2)), ((4, 6, 8), (2, 2, 2), (2, 2, 2)), ((5, 7, 9), (3, 3, 3), (2, 2, 2)),
                                           ((3, 5, 8), (3, 3, 3), (1, 1, 1))])
def test_maxpool3d_with_input_alias_adds_the_other_consumers_gradient_in_its_backward(backend, size, k, stride):
    """r05: maxpool3d_same(x, pass_input=True) -> (y, alias): a second consumer of x reads the alias; its gradient reaches the pool node and is added by the
    backward KERNEL (segx_maxpool3d_bwd addend; four-cell and generic strided gathers) -- or, for a stride-1 pool, by one torch add.  Same numbers as
    two independent consumers of x, bit for bit (the sum has two terms)."""
    x = torch.relu(rnd(2, 3, *size, seed=24)).requires_grad_(True)
    y, xa = SF.maxpool3d_same(x, k, stride, pass_input=True)
    other = (xa * xa).sum() * 0.5 + xa.sum()
    G = rnd(*y.shape, seed=25)
    (y * G).sum().backward(retain_graph=True) if False else ((y * G).sum() + other).backward()
    xr = x.detach().clone().requires_grad_(True)
    yr = SF.maxpool3d_same(xr, k, stride)
    ((yr * G).sum() + (xr * xr).sum() * 0.5 + xr.sum()).backward()
    assert torch.equal(y, yr.detach()) and torch.equal(x.grad, xr.grad)
    # only the alias used / only the pooled output used
    x2 = x.detach().clone().requires_grad_(True)
    y2, xa2 = SF.maxpool3d_same(x2, k, stride, pass_input=True)
    (xa2 * 3.0).sum().backward()
    assert torch.equal(x2.grad, torch.full_like(x2, 3.0))
    x3 = x.detach().clone().requires_grad_(True)
    y3, _ = SF.maxpool3d_same(x3, k, stride, pass_input=True)
    (y3 * G).sum().backward()
    x4 = x.detach().clone().requires_grad_(True)
    (SF.maxpool3d_same(x4, k, stride) * G).sum().backward()
    assert torch.equal(x3.grad, x4.grad)


@pytest.mark.parametrize('size', [(4, 5, 8), (6, 7, 7), (12, 16, 16)])
def test_maxpool3d_bwd_honours_addend_for_stride_1_pools_through_the_abi(backend, size):
    """segx_maxpool3d_bwd(..., addend) with a stride-1 pool used to take a kernel that never reads `addend` and return rc 0; now such a call is served
    by the gather that adds it: dX == dX(without addend) + addend, bit for bit (two terms)."""
    L = backend.L
    B, C = 2, 2
    D, H, W = size
    geom = (D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1)
    x = torch.relu(rnd(B, C, *size, seed=71))
    G, add = rnd(B, C, *size, seed=72), rnd(B, C, *size, seed=73)
    y, arg = torch.empty_like(x), torch.empty(x.shape, dtype=torch.int32)
    L.maxpool3d_fwd(x, y, arg, B * C, geom)
    dx0, dx1 = torch.empty_like(x), torch.empty_like(x)
    L.maxpool3d_bwd(G, arg, dx0, B * C, geom)
    L.maxpool3d_bwd(G, arg, dx1, B * C, geom, add)
    assert torch.equal(dx1, dx0 + add)


@pytest.mark.parametrize('Cout,k', [(40, (3, 3, 3)), (130, (1, 3, 3)), (8, (7, 7, 7))])
def test_conv3d_forward_split_k(backend, Cout, k):
    """Forward with the contraction split over 3 slabs (the low-resolution Inception stages) == un-split result; both the
    whole-window tap mask (<= 32 taps) and the per-axis masks (7x7x7) are exercised, Cout on both tile shapes."""
    L = backend.L
    B, Cin, size = 2, 6, (5, 6, 7)
    x = rnd(B, Cin, *size, seed=21); w = rnd(Cout, Cin, *k, seed=22) * 0.2
    pads = SF._same_pads(size, k, (1, 1, 1))
    geom = (Cin,) + size + size + k + (1, 1, 1) + tuple(p[0] for p in pads)
    y1 = torch.full((B, Cout) + size, float('nan')); y3 = torch.full((B, Cout) + size, float('nan'))
    L.conv3d_fwd(x, w, y1, B, Cout, geom)
    L.conv3d_fwd(x, w, y3, B, Cout, geom, 3, torch.empty(3 * y3.numel()))
    close(y1, _ref_conv(x, w, (1, 1, 1)))
    close(y3, y1, 1e-5)
    assert L.conv3d_splitk(B, Cout, geom, False) >= 1 and L.conv3d_splitk(4, 192, (832, 3, 14, 14, 3, 14, 14, 3, 3, 3, 1, 1, 1, 1, 1, 1), False) > 1


@pytest.mark.parametrize('B,H,W,stride,pad,Cout,xgrad', [(2, 20, 24, 1, (1, 1, 1, 1), 48, False), (2, 32, 32, 2, (0, 1, 0, 1), 48, False), (3, 17, 19, 2, (1, 1, 1, 1), 24, False),
                                                         (1, 64, 128, 2, (0, 1, 0, 1), 48, False), (2, 20, 24, 1, (1, 1, 1, 1), 48, True)])
def test_stem_conv2d_direct_and_on_implicit_gemm(backend, B, H, W, stride, pad, Cout, xgrad):
    """EfficientNet stem (efficientnet/model.py:128, 163): dense 3 x 3, 3 -> c0 channels, stride 1 / 2, static TF-'same' pads.  An input without gradient takes the direct
    kernels of stem2d.hip (forward; dW through the im2col matrix and the batch-reduced skinny GEMM), one with gradient the implicit-GEMM path (xgrad)."""
    x = rnd(B, 3, H, W, seed=6).requires_grad_(xgrad)
    w = (rnd(Cout, 3, 3, 3, seed=7) * 0.3).requires_grad_(True)
    y = SF.conv2d_dense(x, w, stride, pad)
    assert (type(y.grad_fn).__name__ == '_ConvStem2dBackward') == (not xgrad)
    wr = w.detach().clone().requires_grad_(True)
    xr = x.detach().clone().requires_grad_(xgrad)
    yr = F.conv2d(F.pad(xr, pad), wr, stride=stride)
    assert y.shape == yr.shape
    close(y, yr.detach())
    G = rnd(*y.shape, seed=8)
    y.backward(G); yr.backward(G)
    close(w.grad, wr.grad, 1e-4)
    if xgrad:
        close(x.grad, xr.grad, 1e-4)


@pytest.mark.parametrize('size,k,stride,cin', [((8, 10, 12), (7, 7, 7), (2, 2, 2), 3), ((7, 9, 11), (3, 3, 3), (2, 2, 2), 2),
                                                ((6, 9, 8), (3, 5, 3), (1, 2, 3), 4), ((6, 8, 9), (3, 3, 5), (2, 2, 1), 2)])    # last: 5 w-taps -> scalar kernel
def test_strided_backward_data_direct(backend, size, k, stride, cin):
    """The stem's transposed convolution (7x7x7, stride 2) through the residue-class gather kernel; odd sizes and mixed strides
    exercise classes of different population."""
    x = rnd(1, cin, *size, seed=9).requires_grad_(True)
    w = (rnd(6, cin, *k, seed=10) * 0.1).requires_grad_(True)
    y = SF.conv3d_same(x, w, stride)
    xr, wr = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    yr = _ref_conv(xr, wr, stride)
    G = rnd(*y.shape, seed=11)
    y.backward(G); yr.backward(G)
    close(x.grad, xr.grad, 1e-4)


def test_nonzero_mask_and_label_maps(backend):
    from oracle import segtran_oracle as O
    x = rnd(2, 3, 16, 24, seed=12); x[:, :, :8, :8] = 0; x[1, :, 8:, 16:] = 0
    m = SF.nonzero_mask(x, (8, 8))
    ref = (F.avg_pool2d(x.abs(), 8).sum(dim=1) > 0).float()
    assert torch.equal(m, ref)
    v = rnd(1, 4, 8, 16, 16, seed=13); v[:, :, :4] = 0; v[:, :, :, :8, 8:] = 0
    m3 = SF.nonzero_mask(v, (4, 8, 8))
    assert torch.equal(m3, (F.avg_pool3d(v.abs(), (4, 8, 8)).sum(dim=1) > 0).float())
    g = torch.Generator(device='cpu').manual_seed(14)
    fm = (torch.randint(0, 2, (2, 3, 9, 7), generator=g, device='cpu') * 255).to(torch.uint8)
    assert torch.equal(SF.label_nhot(fm.to(backend.dev), 'fundus').cpu(), O.fundus_map_mask(fm).cpu())
    assert torch.equal(SF.label_nhot(fm.to(backend.dev), 'polyp').cpu(), O.polyp_map_mask(fm).cpu())
    lab = torch.randint(0, 4, (2, 5, 6, 4), generator=g, device='cpu')
    assert torch.equal(SF.label_nhot(lab.to(backend.dev), 'brats').cpu(), O.brats_map_label(lab).cpu())


@pytest.mark.parametrize('Cb,D', [(4, 8), (4, 72), (4, 260), (3, 8)], ids=['rows', 'rows-two-loads', 'cells-deep', 'cells-3-modalities'])
@pytest.mark.parametrize('bias', [False, True], ids=['no-bias', 'bias'])
def test_bridge_mask_equals_mask_of_bridged_image(backend, bias, Cb, D):
    """r05: get_mask(in_bridge_to3(batch)) (segtran3d.py:420-425) straight from the raw batch [B, Cb, H, W, D] equals the mask of the materialised, permuted bridge
    output: cells the brain does not reach are exactly 0 in every modality, a cell alive only in a modality the bridge gives no weight is background, and a NaN voxel makes its cell background (NaN > 0 is false)."""
    B, H, W, pool = 2, 16, 24, (4, 8, 8)                                                # D rows of <= 256 floats with 4 modalities: the row form; else one wave per cell
    x = rnd(B, Cb, H, W, D, seed=61)
    x[:, :, :8, :8] = 0; x[1, :, 8:, 16:, 4:] = 0
    w = rnd(3, Cb, 1, 1, 1, seed=62)
    w[:, Cb - 1] = 0
    x[0, :Cb - 1, 8:, 8:16, :4] = 0                                                           # only the modality no bridge channel reads is alive in this cell
    x[1, 0, 3, 12, 1] = float('nan')
    b = rnd(3, seed=63) if bias else None
    if bias: b[1:] = 0
    m = SF.bridge_mask(x, w, b, pool)
    y = SF.conv1x1(x, w.reshape(3, Cb), b).permute(0, 1, 4, 2, 3)
    ref = SF.nonzero_mask(y, pool)
    assert m.shape == ref.shape == (B, D // 4, H // 8, W // 8)
    assert torch.equal(m, ref)
    assert torch.equal(ref, (F.avg_pool3d(y.abs(), pool).sum(dim=1) > 0).float())
    if not bias:
        assert m[0, 0, 1, 1] == 0 and m[:, :, 0, 0].sum() == 0 and m[1, 0, 0, 1] == 0 and m.sum() > 0
    else:
        assert m[0, 0, 1, 1] == 1                                                          # the bias alone makes the cancelled cell foreground, as in the reference


@pytest.mark.parametrize('B,Cin,Cout,size,k,stride', [(2, 8, 12, (4, 6, 5), (3, 3, 3), (1, 1, 1)), (1, 24, 140, (3, 9, 9), (3, 3, 3), (1, 1, 1)),
                                                      (2, 16, 16, (5, 7, 7), (1, 3, 3), (1, 1, 1)), (1, 8, 8, (6, 8, 8), (3, 3, 3), (2, 2, 2)),
                                                      (1, 8, 72, (4, 6, 10), (3, 3, 3), (1, 1, 1)), (2, 16, 24, (3, 5, 8), (3, 3, 3), (1, 1, 1)),
                                                      (1, 8, 16, (2, 3, 16), (1, 3, 3), (1, 1, 1)), (1, 8, 96, (2, 3, 8), (3, 3, 3), (1, 1, 1)),
                                                      (2, 8, 16, (3, 6, 32), (3, 3, 3), (2, 2, 2)), (1, 8, 24, (2, 4, 32), (1, 5, 5), (1, 2, 2)),
                                                      (1, 8, 16, (2, 3, 12), (3, 3, 3), (1, 1, 1)), (2, 16, 72, (2, 5, 20), (3, 3, 3), (1, 1, 1)), (1, 8, 136, (3, 2, 28), (1, 3, 3), (1, 1, 1))])
@pytest.mark.parametrize('wgrad_all', [0, 1], ids=['wgrad-x6-fast-rows', 'wgrad-x6-everywhere'])
def test_conv3d_on_the_bf16x6_engine(backend, B, Cin, Cout, size, k, stride, wgrad_all):
    """Forward, backward-data and backward-weight convolutions (packed contraction order) through the implicit GEMM on the bf16x6 engine:
    same loaders on the global side, operands split into bf16 planes on their way into LDS.  Position counts that are not multiples of
    8 / 32 and rows that wrap inside a thread's position octet (OW = 5, 7, 9, 10) exercise the incremental decode of the weight-gradient loader;
    OW = 8 / 16 its row-of-eight fast path (the one the product uses: rows read with 16-byte loads at stride 1 and 2 -- the last two cases --, windows
    sticking out of the row on both sides, the first / last floats of a sample through the scalar form; OW = 12 / 20 / 28: the two-quad form whose
    second quad may lie on the next output row)."""
    L = backend.L
    prev = L.set_engine('x6')
    L.c.segx_tune(7, wgrad_all)          # 1: also the general (per-position decode) weight-gradient gather; 0: only rows of 8 consecutive floats (OW % 8 == 0)
    try:
        L.x6_launches()
        dgrad = stride == (1, 1, 1)                      # the product differentiates strided convolutions w.r.t. x only for the 3-channel stem
        x = rnd(B, Cin, *size, seed=61).requires_grad_(dgrad)
        w = (rnd(Cout, Cin, *k, seed=62) * 0.2).requires_grad_(True)
        y = SF.conv3d_same(x, w, stride)
        xr, wr = x.detach().clone().requires_grad_(dgrad), w.detach().clone().requires_grad_(True)
        yr = _ref_conv(xr, wr, stride)
        close(y, yr.detach(), 1e-5)
        G = rnd(*y.shape, seed=63)
        y.backward(G); yr.backward(G)
        if dgrad:
            close(x.grad, xr.grad, 1e-4)
        close(w.grad, wr.grad, 1e-4)
        assert L.x6_launches() >= 1                      # the engine really ran (forward; the backward passes where their operands are float4-legal)
    finally:
        L.c.segx_tune(7, 0)
        L.set_engine(prev)


# (B, Cin, Cout, size): W % 8 == 0 -> 4 x 4 x 8 tiles, W % 4 == 0 and D % 8 == 0 -> 8 x 4 x 4 tiles; Cout 24 / 72 / 136 / 200: the 64-, 128- and 192-row tiles with a
# partial last tile; Cin 8 / 16 / 24: one, two, three channel blocks; several tiles along every axis (halo rows of neighbours, zero padding at all six faces)
@pytest.mark.parametrize('B,Cin,Cout,size,mtile', [(2, 8, 24, (4, 4, 8), 0), (1, 16, 72, (8, 8, 16), 0), (1, 24, 136, (4, 8, 8), 0), (1, 8, 200, (4, 4, 8), 0),
                                                   (2, 16, 40, (8, 4, 4), 0), (1, 8, 72, (16, 8, 12), 0), (1, 24, 136, (8, 4, 4), 0), (1, 8, 200, (8, 8, 4), 192),
                                                   (1, 8, 72, (4, 4, 8), 64), (1, 16, 200, (8, 4, 4), 128),
                                                   (1, 8, 24, (8, 7, 14), 0), (2, 16, 72, (11, 8, 8), 0), (1, 8, 40, (7, 11, 15), 0), (1, 8, 24, (24, 14, 14), 0)])      # edge tiles beyond the extent (masked)
def test_conv3d_halo_kernel_forward_and_data_gradient(backend, B, Cin, Cout, size, mtile):
    """r06: the LDS-resident-halo form of the 3 x 3 x 3 stride-1 'same' convolutions (conv3d_halo.hip) through the C ABI -- forward with the pre-split filter bank
    of pack mode 0 and the data gradient (the same kernel on pack mode 1) against F.conv3d and its autograd; channel-slice operands (sample strides) included."""
    L = backend.L
    prev = L.set_engine('x6')
    assert L.c.segx_tune(17, 1) == 0
    try:
        D, H, W = size
        geom = (Cin, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1)
        assert L.conv3d_halo_ok(B, Cout, geom)
        assert not L.conv3d_halo_ok(B, Cout, (Cin, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 0)) and not L.conv3d_halo_ok(B, Cout, (Cin + 4,) + geom[1:])
        assert not L.conv3d_halo_ok(B, Cout, (Cin, 5, 10, 11, 5, 10, 11) + geom[7:])          # tiles would be > 1.5 x the extent
        x = rnd(B, Cin + 8, *size, seed=81)                       # the convolution reads channels 8.. of a wider tensor (x_bs = its sample stride)
        w = rnd(Cout, Cin, 3, 3, 3, seed=82) * 0.2
        y = torch.full((B, Cout + 3, D, H, W), 7.0)               # ... and writes channels 3.. of a wider one (y_bs)
        L.x6_launches()
        L.conv3d_halo_fwd(x[:, 8:], L.conv3d_halo_pack(w, Cout, Cin, 0), y[:, 3:], B, Cout, geom, x_bs=(Cin + 8) * D * H * W, y_bs=(Cout + 3) * D * H * W, mtile=mtile)
        assert L.x6_launches() == 1
        xr = x[:, 8:].clone().requires_grad_(True)
        yr = F.conv3d(xr, w, None, 1, 1)
        close(y[:, 3:], yr.detach(), 1e-5)
        assert torch.equal(y[:, :3], torch.full((B, 3, D, H, W), 7.0))      # nothing written outside the slice
        G = rnd(B, Cout, *size, seed=83)
        yr.backward(G)
        g2 = (Cout, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1)
        dx = torch.empty(B, Cin, D, H, W)
        L.conv3d_halo_fwd(G, L.conv3d_halo_pack(w, Cin, Cout, 1), dx, B, Cin, g2, mtile=0)
        close(dx, xr.grad, 1e-5)
    finally:
        assert L.c.segx_tune(17, 256) == 0
        L.set_engine(prev)


# Cout 40 / 136: the 4-wave (128-row) form with a partial tile / two tiles; Cout 192 / 200: the 6-wave (192-row) form (+ a second, mostly empty tile); Cin 8 / 16 / 24;
# W = 8 / 16 / 24: whole octets; W = 12 and ragged D / H: masked edge octets and blocks (W % 4 == 0 throughout: dY is read in 16-byte pieces)
@pytest.mark.parametrize('B,Cin,Cout,size', [(2, 8, 40, (4, 4, 8)), (1, 16, 136, (8, 8, 16)), (1, 8, 192, (4, 8, 8)), (2, 24, 200, (4, 4, 8)), (1, 16, 40, (8, 4, 24)),
                                             (1, 8, 40, (8, 8, 12)), (2, 16, 72, (7, 8, 16)), (1, 8, 192, (7, 7, 8)), (3, 8, 24, (12, 12, 8))])
def test_conv3d_halo_weight_gradient(backend, B, Cin, Cout, size):
    """r06: dW of the 3 x 3 x 3 stride-1 'same' convolutions with the halo resident as three x-shifted windows (conv3d_halo.hip), K-split over workgroups + deterministic
    slab reduction, against autograd's weight gradient of F.conv3d; X read as a channel slice of a wider tensor (x_bs)."""
    L = backend.L
    prev = L.set_engine('x6')
    assert L.c.segx_tune(17, 1) == 0
    try:
        D, H, W = size
        geom = (Cin, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1)
        assert L.conv3d_halo_wgrad_ok(B, Cout, geom)
        xw = rnd(B, Cin + 8, *size, seed=101)
        dy = rnd(B, Cout, *size, seed=102)
        w = (rnd(Cout, Cin, 3, 3, 3, seed=103) * 0.2).requires_grad_(True)
        F.conv3d(xw[:, 8:], w, None, 1, 1).backward(dy)
        dw = torch.full_like(w, 7.0)
        L.x6_launches()
        L.conv3d_halo_wgrad(dy, xw[:, 8:], dw, B, Cout, geom, x_bs=(Cin + 8) * D * H * W)
        assert L.x6_launches() == 1
        close(dw, w.grad, 2e-5)
        dw2 = torch.empty_like(w)
        L.conv3d_halo_wgrad(dy, xw[:, 8:], dw2, B, Cout, geom, x_bs=(Cin + 8) * D * H * W)
        assert torch.equal(dw, dw2)                                   # deterministic: fixed split order
    finally:
        assert L.c.segx_tune(17, 256) == 0
        L.set_engine(prev)


def test_conv3d_same_takes_the_halo_kernel_where_it_applies(backend):
    """SF.conv3d_same / SF.conv3d_slices route eligible layers (knob 16 on, >= knob 17 tiles) through the halo kernel -- same results as with the knob off."""
    L = backend.L
    prev = L.set_engine('x6')
    outs = {}
    try:
        for halo in (1, 0):
            assert L.c.segx_tune(16, halo) == 0 and L.c.segx_tune(17, 1) == 0
            x = rnd(1, 24, 4, 8, 8, seed=91).requires_grad_(True)
            w1, w2 = (rnd(16, 8, 3, 3, 3, seed=92) * 0.2).requires_grad_(True), (rnd(24, 16, 3, 3, 3, seed=93) * 0.2).requires_grad_(True)
            calls = []
            from segtran_amd import segx
            Lf = segx.lib()                                           # the instance the autograd layer calls (on the device not the fixture's object)
            orig = Lf.conv3d_halo_fwd
            Lf.conv3d_halo_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            try:
                y1, y2 = SF.conv3d_slices(x, w1, w2)
                y3 = SF.conv3d_same(y2, w2[:, :8].repeat(1, 3, 1, 1, 1).contiguous())
                (y1.sum() * 0.5 + (y3 * y3).sum()).backward()
            finally:
                del Lf.conv3d_halo_fwd
            assert len(calls) == (6 if halo else 0)                  # three forward + three data-gradient launches (the weight gradients: conv3d_halo_wgrad)
            outs[halo] = (y1.detach(), y3.detach(), x.grad.clone(), w1.grad.clone(), w2.grad.clone())
        for a, b in zip(outs[1], outs[0]):
            close(a, b, 2e-5)
    finally:
        assert L.c.segx_tune(16, 1) == 0 and L.c.segx_tune(17, 256) == 0
        L.set_engine(prev)


@pytest.mark.parametrize('engine', ['f32', 'x6'])
@pytest.mark.parametrize('B,H,W,D', [(2, 8, 12, 6), (1, 6, 16, 8)])
def test_stem_and_bridge_as_one_space_to_depth_convolution(backend, engine, B, H, W, D):
    """r06: Conv3d_1a_7x7(in_bridge_to3(batch)) as ONE stride-(2, 2, 1) 7 x 7 x 4 convolution over the space-to-depth image along W + the bias map through the zero
    padding (SF.stem_bridge_conv_s2d) == the two-step computation in PyTorch: output (border voxels included: the bias must not leak into the padding) and the
    gradients of the stem filters, the bridge weight and the bridge bias."""
    L = backend.L
    prev = L.set_engine(engine)
    try:
        batch = rnd(B, 4, H, W, D, seed=111)
        ws = (rnd(8, 3, 7, 7, 7, seed=112) * 0.1).requires_grad_(True)
        wb = (rnd(3, 4, 1, 1, 1, seed=113) * 0.5).requires_grad_(True)
        bb = (rnd(3, seed=114) * 0.5).requires_grad_(True)
        y = SF.stem_bridge_conv_s2d(batch, ws, wb, bb)
        G = rnd(*y.shape, seed=115)
        y.backward(G)
        got = [t.grad.clone() for t in (ws, wb, bb)]
        for t in (ws, wb, bb):
            t.grad = None
        rgb = F.conv3d(batch.permute(0, 1, 4, 2, 3), wb, bb)
        yr = F.conv3d(F.pad(rgb, (2, 3, 2, 3, 2, 3)), ws, None, 2)
        yr.backward(G)
        close(y, yr.detach(), 2e-5)
        for a, t in zip(got, (ws, wb, bb)):
            close(a, t.grad, 1e-4)
    finally:
        L.set_engine(prev)


def test_input_bridge_composed_into_the_stem(backend):
    """in_bridge_to3 (Conv3d 4 -> 3, 1x1x1, bias) followed by the stride-2 'same' stem convolution == ONE convolution of [x, 1, 0, 0, 0] with the
    composed 8-channel filters (segx_stem_compose_fwd / segx_bridge_input): forward, and the gradients of the stem filters, the bridge weight and
    the bridge bias through the composition's chain rule -- against the two-step computation in PyTorch (border voxels included: the bias must
    NOT leak into the zero padding)."""
    B, Cb, H, W, D, O, k = 2, 4, 8, 16, 6, 16, (3, 5, 5)
    x = rnd(B, Cb, H, W, D, seed=71)
    ws = (rnd(O, 3, *k, seed=72) * 0.2).requires_grad_(True)
    wb = (rnd(3, Cb, 1, 1, 1, seed=73) * 0.5).requires_grad_(True)
    bb = (rnd(3, seed=74) * 0.5).requires_grad_(True)
    x8 = SF.bridge_input(x, 8)
    assert x8.shape == (B, 8, D, H, W)
    assert torch.equal(x8[:, :Cb], x.permute(0, 1, 4, 2, 3)) and bool((x8[:, Cb] == 1).all()) and bool((x8[:, Cb + 1:] == 0).all())
    y = SF.conv3d_same(x8, SF.stem_compose(ws, wb, bb, 8), (2, 2, 2))
    wsr, wbr, bbr = (t.detach().clone().requires_grad_(True) for t in (ws, wb, bb))
    rgb = F.conv3d(x, wbr, bbr).permute(0, 1, 4, 2, 3)
    yr = _ref_conv(rgb, wsr, (2, 2, 2))
    close(y, yr.detach(), 2e-5)
    G = rnd(*y.shape, seed=75)
    y.backward(G); yr.backward(G)
    close(ws.grad, wsr.grad, 1e-4); close(wb.grad, wbr.grad, 1e-4); close(bb.grad, bbr.grad, 1e-4)


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('size', [(3, 5, 8), (3, 5, 7)], ids=['quad-planes', 'odd-planes'])
def test_inception_module_branches_write_the_concatenation_in_place(backend, training, size, monkeypatch):
    """r05: the module's output assembled by the branches' BatchNorm kernels (SF.bn_act_cat, cat_in_place) against torch.cat of the same branch tensors: output,
    input gradient, every parameter gradient and every buffer bit for bit (planes of 3 x 5 x 7 floats cannot be written in quads and keep the copy)."""
    from segtran_amd.networks.aj_i3d.aj_i3d import InceptionModule
    import copy
    a = InceptionModule(24, [16, 16, 24, 8, 16, 8], 'm')
    with torch.no_grad():
        for p in a.parameters():
            p.copy_(rnd(*p.shape, seed=int(p.numel()) % 97) * (0.2 if p.dim() > 1 else 0.5) + (1.0 if p.dim() == 1 else 0.0))
    b = copy.deepcopy(a)
    dev = torch.get_default_device()
    a.to(dev).train(training); b.to(dev).train(training)
    x = rnd(2, 24, *size, seed=83).requires_grad_(True); xr = x.detach().clone().requires_grad_(True)
    monkeypatch.setattr(InceptionModule, 'fuse_reductions', True)
    monkeypatch.setattr(InceptionModule, 'cat_in_place', True)
    y = a(x)
    monkeypatch.setattr(InceptionModule, 'cat_in_place', False)
    yr = b(xr)
    assert torch.equal(y, yr)
    G = rnd(*y.shape, seed=84)
    y.backward(G); yr.backward(G)
    assert torch.equal(x.grad, xr.grad)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p.grad, q.grad), k
    for (k, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.equal(u, v), k


@pytest.mark.parametrize('training', [True, False])
def test_inception_module_fused_reductions_equal_the_four_branches(backend, training):
    """InceptionModule with the b1a | b2a reductions run as one pointwise convolution + one BatchNorm and the 3x3x3 convolutions reading channel
    slices in place (segx_conv3d_*_bs) == the reference's four independent branches: output, input gradient, every parameter gradient and the
    running statistics of all six BatchNorm layers."""
    from segtran_amd.networks.aj_i3d.aj_i3d import InceptionModule
    import copy
    torch.manual_seed(5)
    a = InceptionModule(24, [16, 16, 24, 8, 16, 8], 'm')
    with torch.no_grad():
        for p in a.parameters():
            p.copy_(rnd(*p.shape, seed=int(p.numel()) % 97) * (0.2 if p.dim() > 1 else 0.5) + (1.0 if p.dim() == 1 else 0.0))
    b = copy.deepcopy(a)
    dev = torch.get_default_device()
    a.to(dev).train(training); b.to(dev).train(training)
    x = rnd(2, 24, 3, 5, 8, seed=81).requires_grad_(True); xr = x.detach().clone().requires_grad_(True)
    prev = InceptionModule.fuse_reductions
    try:
        InceptionModule.fuse_reductions = True
        y = a(x)
        InceptionModule.fuse_reductions = False
        yr = b(xr)
    finally:
        InceptionModule.fuse_reductions = prev
    close(y, yr.detach(), 1e-5)
    G = rnd(*y.shape, seed=82)
    y.backward(G); yr.backward(G)
    close(x.grad, xr.grad, 1e-4)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        close(p.grad, q.grad, 2e-4)
    for (k, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
        if u.dtype.is_floating_point:
            close(u, v, 1e-5)
        else:
            assert int(u) == int(v), k


def test_inception_module_block_node_equals_per_op_nodes(backend, monkeypatch):
    """SF.block_node over a whole Inception module (aj_i3d.py:121-126 -- the alias outputs of the fused reduction GEMM and of conv3d_slices included) against the per-op
    autograd nodes: output, input gradient, parameter gradients and buffers bit for bit."""
    from segtran_amd.networks.aj_i3d.aj_i3d import InceptionModule
    from segtran_amd import functional as SF
    import copy
    a = InceptionModule(24, [16, 16, 24, 8, 16, 8], 'm')
    with torch.no_grad():
        for p in a.parameters():
            p.copy_(rnd(*p.shape, seed=int(p.numel()) % 97) * (0.2 if p.dim() > 1 else 0.5) + (1.0 if p.dim() == 1 else 0.0))
    b = copy.deepcopy(a)
    dev = torch.get_default_device()
    a.to(dev).train(); b.to(dev).train()
    x = rnd(2, 24, 3, 5, 8, seed=83).requires_grad_(True); xr = x.detach().clone().requires_grad_(True)
    monkeypatch.setattr(InceptionModule, 'fuse_reductions', True)
    monkeypatch.setattr(InceptionModule, 'cat_in_place', True)
    monkeypatch.setattr(SF, 'block_nodes', True)
    y = a(x * 1.0)
    assert type(y.grad_fn).__name__ == '_BlockBackward'
    monkeypatch.setattr(SF, 'block_nodes', False)
    yr = b(xr * 1.0)
    assert type(yr.grad_fn).__name__ != '_BlockBackward'
    assert torch.equal(y, yr)
    G = rnd(*y.shape, seed=84)
    y.backward(G); yr.backward(G)
    assert torch.equal(x.grad, xr.grad)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p.grad, q.grad), k
    for (k, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.equal(u, v), k
