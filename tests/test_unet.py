"""U-Net host of the Polyformer layer (reference code/networks/unet2d/): the new host-side ops against plain PyTorch, and
the whole network against a fixture produced by the reference's own UNet (tests/golden/make_golden.py case_unet).  The ops run on the fiber
emulator here; the whole network (2.4 GMAC forward, 7 with backward) runs on the HIP build, and on the emulator only with SEGX_SLOW_TESTS=1
(evaluation-mode forward, ~4 minutes)."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from test_modules import golden_on, assert_close
from segtran_amd import functional as SF
from segtran_amd.synth import synth_state_dict, sample


def _rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator(device='cpu').manual_seed(seed), device='cpu')


@pytest.mark.parametrize('shape,size', [((2, 3, 4, 6), (8, 12)), ((1, 2, 3, 5), (6, 10)), ((2, 2, 5, 4), (9, 7)), ((1, 3, 1, 4), (1, 8))])
def test_bilinear_align_corners_vs_torch(backend, shape, size):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (unet_parts.py:48) and general sizes: forward and adjoint."""
    x = _rnd(*shape, seed=1).to(backend.dev).requires_grad_(True)
    xr = x.detach().cpu().clone().requires_grad_(True)
    y = SF.interp_linear(x, size, align_corners=True)
    yr = F.interpolate(xr, size=size, mode='bilinear', align_corners=True)
    assert_close(y, yr.detach(), 2e-6, 'fwd')
    G = _rnd(*yr.shape, seed=2)
    y.backward(G.to(backend.dev)); yr.backward(G)
    assert_close(x.grad, xr.grad, 2e-6, 'bwd')


def test_conv3x3_bias_and_maxpool2_vs_torch(backend):
    x = _rnd(2, 5, 9, 11, seed=3).to(backend.dev).requires_grad_(True)
    w = (0.2 * _rnd(7, 5, 3, 3, seed=4)).to(backend.dev).requires_grad_(True)
    b = _rnd(7, seed=5).to(backend.dev).requires_grad_(True)
    xr, wr, br = (t.detach().cpu().clone().requires_grad_(True) for t in (x, w, b))
    y = SF.maxpool2d(SF.conv2d_bias(x, w, b, pad=1), 2)
    yr = F.max_pool2d(F.conv2d(xr, wr, br, padding=1), 2)
    assert y.shape == yr.shape == (2, 7, 4, 5)
    assert_close(y, yr.detach(), 1e-5, 'fwd')
    G = _rnd(*yr.shape, seed=6)
    y.backward(G.to(backend.dev)); yr.backward(G)
    for a, r, n in ((x, xr, 'dx'), (w, wr, 'dw'), (b, br, 'db')):
        assert_close(a.grad, r.grad, 2e-5, n)


def test_conv_transpose2x2_vs_torch(backend):
    """nn.ConvTranspose2d(k 2, s 2) (unet_parts.py:53) as pointwise GEMM + re-arrangement: forward, dX, dW, db against F.conv_transpose2d."""
    x = _rnd(2, 8, 5, 7, seed=11).to(backend.dev).requires_grad_(True)
    w = (0.3 * _rnd(8, 4, 2, 2, seed=12)).to(backend.dev).requires_grad_(True)
    b = _rnd(4, seed=13).to(backend.dev).requires_grad_(True)
    xr, wr, br = (t.detach().cpu().clone().requires_grad_(True) for t in (x, w, b))
    y = SF.conv_transpose2x2(x, w, b)
    yr = F.conv_transpose2d(xr, wr, br, stride=2)
    assert y.shape == yr.shape == (2, 4, 10, 14)
    assert_close(y, yr.detach(), 1e-5, 'fwd')
    G = _rnd(*yr.shape, seed=14)
    y.backward(G.to(backend.dev)); yr.backward(G)
    for a, r, n in ((x, xr, 'dx'), (w, wr, 'dw'), (b, br, 'db')):
        assert_close(a.grad, r.grad, 2e-5, n)


@pytest.mark.gpu
def test_unet_transposed_conv_decoder_vs_reference():
    """UNet(bilinear=False) (unet_model.py:17-27 with factor 1, unet_parts.py:52-54) against the reference's own network in evaluation mode:
    logits, input gradient, every parameter gradient (tests/golden/unet_deconv.npz, make_golden.py case_unet_deconv)."""
    from segtran_amd.networks.unet2d import UNet
    dev = torch.device('cuda', 0)
    g = golden_on('unet_deconv', dev)
    net = UNet(3, 2, False, None)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    X = g['X'].clone().requires_grad_(True)
    Y = net(X)
    assert_close(Y, g['Y'], 5e-5, 'logits')
    (Y * g['G']).sum().backward()
    assert_close(X.grad, g['dX'], 3e-4, 'dX')
    grads = dict(net.named_parameters())
    gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith('grad:'))
    n = 0
    for k, v in g.items():
        if k.startswith('grad:'):
            assert_close(sample(grads[k[5:]].grad, v.numel()), v, 5e-4, k, scale=gscale); n += 1
    assert n == len(grads)


def test_discriminator_and_gradient_reversal_vs_reference(backend):
    """networks/discriminator.py + revgrad.py (the adversarial branch of the few-shot recipe, train2d.py:876-926, 1259-1284) against the
    reference's own module in training mode: scores, the REVERSED input gradient, parameter gradients, BatchNorm running statistics; same
    state_dict keys (the reversal layer shifts the Sequential indices by one, as in the reference)."""
    from segtran_amd.networks.discriminator import Discriminator
    g = golden_on('discriminator', backend.dev)
    net = Discriminator(8, num_classes=1, do_revgrad=True, num_base_chan=8)
    assert list(net.state_dict().keys()) == [str(k) for k in np.load(os.path.join(os.path.dirname(__file__), 'golden', 'discriminator.npz'))['keys']]
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}))
    net = net.to(backend.dev).train()
    X = g['X'].clone().requires_grad_(True)
    Y = net(X)
    assert_close(Y, g['Y'], 5e-5, 'scores')
    (Y * g['G']).sum().backward()
    assert_close(X.grad, g['dX'], 5e-4, 'dX (reversed)')
    grads = dict(net.named_parameters())
    gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith('grad:'))
    for k, v in g.items():
        if k.startswith('grad:'):
            assert_close(grads[k[5:]].grad, v, 5e-4, k, scale=gscale)
        if k.startswith('stat:'):
            assert_close(net.state_dict()[k[5:]], v, 2e-5, k)


def test_polyformer_mode_selects_the_reference_parameter_sets():
    """train2d.py:463-503 (host logic): --polyformer source|target optimises only the chosen pieces of the Squeeze-and-Expansion layers."""
    from argparse import Namespace
    from segtran_amd import engine
    from segtran_amd.networks.unet2d import UNet
    pargs = Namespace(polyformer_mode='source', num_attractors=16, num_modes=4, tie_qk_scheme='loose', qk_have_bias=True, pos_code_type='lsinu')
    net = UNet(3, 2, True, pargs)
    net.discriminator, net.recon = None, None
    names = lambda mode: sorted(n for n, _ in engine.polyformer_optimized_params(net, mode, is_segtran=False))      # noqa: E731
    layer = net.polyformer.polyformer_layers[0]
    assert names('k') == sorted(n for n, _ in layer.in_ator_trans.key.named_parameters())
    assert names('k,v') == sorted([n for n, _ in layer.in_ator_trans.key.named_parameters()] + [n for n, _ in layer.in_ator_trans.out_trans.first_linear.named_parameters()])
    assert len(names('allpoly')) == len(list(net.polyformer.polyformer_layers.named_parameters())) and len(names('allnet')) == len(list(net.parameters()))
    assert names('h') == ['conv.bias', 'conv.weight']
    opt = engine.init_optimizer(net, 'fundus', polyformer_mode='target', poly_opt_mode='k')
    assert sum(len(gp['params']) for gp in opt.param_groups) == len(names('k')) and all(gp['weight_decay'] == 0 for gp in opt.param_groups)


def _build(dev):
    from argparse import Namespace
    from segtran_amd.networks.unet2d import UNet
    pargs = Namespace(polyformer_mode='source', num_attractors=16, num_modes=4, tie_qk_scheme='loose', qk_have_bias=True, pos_code_type='lsinu')
    net = UNet(3, 2, True, pargs)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd)
    return net.to(dev), sd


def test_unet_state_dict_keys_match_reference_fixture():
    """the fixture's gradient / statistics names ARE the reference's parameter / buffer names"""
    g = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'unet_poly.npz'))
    net, _ = _build('cpu')
    params = {k for k, _ in net.named_parameters()}
    ref_params = {k[len('train:grad:'):] for k in g.files if k.startswith('train:grad:')} | {str(k) for k in g['unused']}
    assert params == ref_params
    bufs = {k for k, _ in net.named_buffers() if 'num_batches' not in k}
    assert {k[len('train:stat:'):] for k in g.files if k.startswith('train:stat:')} == bufs


@pytest.mark.parametrize('mode', ['eval', 'train'])
@pytest.mark.parametrize('tile_engine', ['x6', 'f32'])
def test_unet_polyformer_vs_reference(backend, mode, tile_engine):
    """Logits, input gradient, parameter gradients (and, after a training-mode pass, the BatchNorm running statistics) against the reference's
    own UNet, on both tile engines.

    One combination is forward-only: TRAINING-mode gradients on the bf16x6 engine.  With batch statistics over as few as n = 48 values (the
    2 x 3 and 4 x 6 maps) a single ReLU input that lies within fp32 rounding of zero decides 1/48 of a channel's mean gradient, and this network
    has ~1.3 M ReLU inputs of scale 0.03 (smallest |input| over 300 input seeds: 5e-10 .. 1e-7, fp64 evaluation) -- so two fp32-accurate
    implementations agree on every mask only if their roundings are correlated.  The fp32-MFMA engine accumulates like the reference does and
    agrees (dX to 2e-6); the bf16x6 engine is equally accurate against fp64 (every convolution of this very step replayed: 2e-7 .. 1e-6, same as
    the fp32 engine) but rounds differently: on the device ONE of 24 576 inputs of `up1`'s first BatchNorm (channel 456, |input| < 1e-7) took the
    other side of the kink and moved dX by 1.7 % (device session r02_g).  Its gradients are therefore checked in evaluation mode (no batch
    coupling), its training-mode forward (logits, statistics) strictly."""
    if backend.name == 'emu' and (mode == 'train' or tile_engine == 'f32' or not os.environ.get('SEGX_SLOW_TESTS')):
        pytest.skip('the 36x48 fixture is 2.4 GMAC forward, ~7 with backward: device only (SEGX_SLOW_TESTS=1 runs one forward on the emulator)')
    g = golden_on('unet_poly', backend.dev)
    from segtran_amd import segx
    L = segx.lib() if backend.name == 'hip' else backend.L        # the handle the autograd layer uses (the fixture resets the product's)
    prev = L.set_engine(tile_engine)
    try:
        net, sd = _build(torch.get_default_device())
        net.train(mode == 'train'); net.polyformer.eval()                     # the layer's attention dropout off, as in the fixture
        X = g['X'].clone().requires_grad_(backend.name != 'emu')
        Y = net(X)
        assert_close(Y, g[mode + ':Y'], 5e-5, 'logits')
        if backend.name == 'emu':
            return
        (Y * g['G']).sum().backward()
        if mode == 'train':
            for k, v in g.items():
                if k.startswith('train:stat:'):
                    assert_close(net.state_dict()[k[len('train:stat:'):]], v, 2e-5, k)
            assert all(int(m.num_batches_tracked) == 1 for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
            if tile_engine == 'x6':
                return                                                      # see the docstring
        assert_close(X.grad, g[mode + ':dX'], 3e-4, 'dX')
        grads = dict(net.named_parameters())
        pre = mode + ':grad:'
        gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith(pre))
        n = 0
        for k, v in g.items():
            if k.startswith(pre):
                p = grads[k[len(pre):]]
                assert p.grad is not None, k
                assert_close(sample(p.grad, v.numel()), v, 5e-4, k, scale=gscale)
                n += 1
        assert n > 90
        for k in g['unused']:
            assert grads[str(k)].grad is None, k
    finally:
        L.set_engine(prev)
