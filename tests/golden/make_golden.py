"""Generate golden fixtures from the REAL reference (askerlee/segtran @ /root/reference).

Runs ONLY in the build container (the reference does not exist on the GPU box).  It
  1. imports the reference on CPU (tools/refimport.py, recipe of SURVEY.md 8(c)),
  2. loads name-hashed synthetic weights (segtran_amd/synth.py) into the reference modules,
  3. runs them on seeded inputs and stores inputs + expected outputs (+ selected gradients)
     as small .npz files next to this script,
  4. asserts, for every case, that oracle/segtran_oracle.py reproduces the reference
     (this is what "pins" the oracle).

Fixtures are DATA (inputs / expected outputs); no reference source is stored.
Usage:  python tests/golden/make_golden.py [case ...]
"""
import os, re, sys, json, hashlib, copy
import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refimport as R                                     # noqa: E402
from segtran_amd.synth import (synth_state_dict, load_synth, sample, synth_fundus_mask,   # noqa: E402
                               synth_brats, synth_image2d)
from oracle import segtran_oracle as O                    # noqa: E402

torch.set_num_threads(8)
SAMPLE = 4096


OUT_DIR = HERE                                           # tests/test_golden_regen.py points this at a temp dir


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.ndim > 0 and not k.endswith('_digest'):
            # the fp64-REFEREE arrays (logits64, grad64:<name> ...): fp32 storage (6e-8 relative) is far below the 1e-6 differences they referee.
            # The *_digest arrays of case_init keep fp64 -- they are compared at rtol 1e-12 (the downcast had caught them too).
            v = v.astype(np.float32)
        out[k] = v
    path = os.path.join(OUT_DIR, name + '.npz')
    np.savez_compressed(path, **out)
    print('  wrote %s (%.0f KB)' % (name, os.path.getsize(path) / 1024))


def close(a, b, tol, what):
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item()
    assert err <= tol * scale + 1e-30, '%s: oracle vs reference err %.3e (scale %.3e)' % (what, err, scale)
    return err / scale


def mk_shared_config(ss, in_dim, dims, A, pos_dim=2):
    cfg = ss.SegtranConfig()
    cfg.num_translayers = len(dims) - 1
    cfg.translayer_dims = list(dims)
    cfg.translayer_compress_ratios = [1] * len(dims)
    cfg.trans_in_dim = dims[0]
    cfg.min_feat_dim = min(dims)
    cfg.in_feat_dim, cfg.feat_dim = dims[0], dims[1]
    cfg.num_attractors = A
    cfg.pos_dim = pos_dim
    cfg.hidden_dropout_prob = 0.0
    cfg.attention_probs_dropout_prob = 0.0
    return cfg


def load_prefixed(mod, prefix):
    shapes = {prefix + k: tuple(v.shape) for k, v in mod.state_dict().items()}
    sd = synth_state_dict(shapes)
    mod.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    R.quiet(lambda: [m.tie_qk('shared') for m in mod.modules() if hasattr(m, 'tie_qk') and hasattr(m, 'query')])
    return sd


def ref_param_grads(mod, prefix):
    return {prefix + k: p.grad for k, p in mod.named_parameters()}


def oracle_grads(sdg):
    g = {}
    for k, v in sdg.items():
        if v.grad is None:
            continue
        if '.key.' in k:
            continue
        gg = v.grad
        if '.query.' in k:
            kk = k.replace('.query.', '.key.')
            if kk in sdg and sdg[kk].grad is not None:
                gg = gg + sdg[kk].grad
        g[k] = gg
    return g


def req(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
            for k, v in sd.items()}


# --------------------------------------------------------------------------------------
def case_squeeze():
    ss = R.ref_shared()
    for tag, (C, Fd) in {'c64f64': (64, 64), 'c64f32': (64, 32), 'ffn': (64, 32)}.items():
        cfg = mk_shared_config(ss, C, [C, Fd], 16)
        cfg.has_FFN_in_squeeze = tag == 'ffn'              # --squeezeuseffn: the in-squeeze layer keeps its (1-mode) FFN
        mod = R.quiet(ss.SqueezedAttFeatTrans, cfg, 'L')
        prefix = 'voxel_fusion.translayers.0.'
        sd = load_prefixed(mod, prefix)
        mod.eval()
        g = torch.Generator().manual_seed(11)
        X = torch.randn(2, 48, C, generator=g).requires_grad_(True)
        G = torch.randn(2, 48, Fd, generator=g)
        Y = mod(X); (Y * G).sum().backward()
        sdg = req(sd); Xo = X.detach().clone().requires_grad_(True)
        Yo = O.squeezed_att_feat_trans(sdg, prefix[:-1], Xo, 4, ffn_in_squeeze=tag == 'ffn'); (Yo * G).sum().backward()
        close(Yo, Y, 1e-5, 'squeeze Y'); close(Xo.grad, X.grad, 1e-4, 'squeeze dX')
        rg, og = ref_param_grads(mod, prefix), oracle_grads(sdg)
        arrs = dict(X=X, G=G, Y=Y, dX=X.grad)
        gscale = max(v.abs().max().item() for v in rg.values() if v is not None)
        for k, v in rg.items():
            if v is None:
                assert k not in og or og[k].abs().max() == 0, k
                continue
            assert (og[k] - v).abs().max().item() <= 2e-4 * gscale, k
            arrs['grad:' + k] = v
        save('squeeze_' + tag, **arrs)


def case_fusion():
    ss = R.ref_shared()
    dims = [64, 64, 32]
    cfg = mk_shared_config(ss, 64, dims, 16)
    mod = R.quiet(ss.SegtranFusionEncoder, cfg, 'Fusion')
    prefix = 'voxel_fusion.'
    sd = load_prefixed(mod, prefix)
    mod.eval()
    g = torch.Generator().manual_seed(12)
    H2, W2 = 6, 8
    X = torch.randn(2, H2 * W2, 64, generator=g).requires_grad_(True)
    G = torch.randn(2, H2 * W2, 32, generator=g)
    vmask = (torch.rand(2, H2 * W2, 1, generator=g) > 0.25)
    pos = (O.gen_all_indices((H2, W2)).view(-1, 2).float() * 8).unsqueeze(0).repeat(2, 1, 1)
    Y = mod(X, pos, vmask, torch.Size((H2, W2))); (Y * G).sum().backward()
    sdg = req(sd); Xo = X.detach().clone().requires_grad_(True)
    Yo = O.fusion_encoder(sdg, 'voxel_fusion', Xo, pos, vmask, dims); (Yo * G).sum().backward()
    close(Yo, Y, 1e-5, 'fusion Y'); close(Xo.grad, X.grad, 1e-4, 'fusion dX')
    rg, og = ref_param_grads(mod, prefix), oracle_grads(sdg)
    arrs = dict(X=X, G=G, vmask=vmask, pos=pos, Y=Y, dX=X.grad, dims=np.array(dims))
    gscale = max(v.abs().max().item() for v in rg.values() if v is not None)
    for k, v in rg.items():
        if v is None:
            continue
        assert (og[k] - v).abs().max().item() <= 2e-4 * gscale, k
        arrs['grad:' + k] = v
    save('fusion_small', **arrs)


def case_fusion_nosqueeze():
    """--nosqueeze (plain multi-mode self-attention over all tokens), with 'lsinu' codes and with '--pos bias'
    sliding positional biases (2-D and 3-D), small radius so that in-radius and out-of-radius pairs both occur."""
    ss = R.ref_shared()
    dims = [64, 64, 32]
    for tag, (pos_type, shape) in {'lsinu': ('lsinu', (6, 8)), 'bias2d': ('bias', (6, 8)), 'bias3d': ('bias', (3, 4, 5))}.items():
        pd = len(shape)
        cfg = mk_shared_config(ss, 64, dims, 16, pos_dim=pd)
        cfg.use_squeezed_transformer = False
        cfg.pos_code_type, cfg.pos_bias_radius, cfg.pos_code_weight = pos_type, 2, 0.8
        cfg.max_pos_size = (8,) * pd
        mod = R.quiet(ss.SegtranFusionEncoder, cfg, 'Fusion')
        prefix = 'voxel_fusion.'
        shapes = {prefix + k: tuple(v.shape) for k, v in mod.state_dict().items() if '.all_' not in k}   # not the index buffers
        sd = synth_state_dict(shapes)
        mod.load_state_dict({k[len(prefix):]: v for k, v in sd.items()}, strict=False)
        R.quiet(lambda: [m.tie_qk('shared') for m in mod.modules() if hasattr(m, 'tie_qk') and hasattr(m, 'query')])
        mod.eval()
        g = torch.Generator().manual_seed(21)
        N = int(np.prod(shape))
        X = torch.randn(2, N, 64, generator=g).requires_grad_(True)
        G = torch.randn(2, N, 32, generator=g)
        vmask = (torch.rand(2, N, 1, generator=g) > 0.25)
        pos = (O.gen_all_indices(shape).view(-1, pd).float() * 8).unsqueeze(0).repeat(2, 1, 1)
        Y = mod(X, pos, vmask, torch.Size(shape)); (Y * G).sum().backward()
        sdg = req(sd); Xo = X.detach().clone().requires_grad_(True)
        Yo = O.fusion_encoder(sdg, 'voxel_fusion', Xo, pos, vmask, dims, pos_code_weight=0.8, squeezed=False,
                              pos_code_type=pos_type, feat_shape=shape)
        (Yo * G).sum().backward()
        close(Yo, Y, 1e-5, 'nosqueeze Y'); close(Xo.grad, X.grad, 1e-4, 'nosqueeze dX')
        rg, og = ref_param_grads(mod, prefix), oracle_grads(sdg)
        arrs = dict(X=X, G=G, vmask=vmask, pos=pos, Y=Y, dX=X.grad, dims=np.array(dims), shape=np.array(shape))
        gscale = max(v.abs().max().item() for v in rg.values() if v is not None)
        for k, v in rg.items():
            if v is None:
                continue
            assert (og[k] - v).abs().max().item() <= 2e-4 * gscale, k
            arrs['grad:' + k] = v
        save('fusion_nosqueeze_' + tag, **arrs)


def case_fusion_mince():
    """--mince (multi-scale self-attention, CrossMinceAttFeatTrans): 2-D with per-scale sliding biases, 2-D with 'lsinu' codes on
    a grid the scales do not divide (scale_factor coordinate rule), 3-D with biases.  Query/key stay untied, as in the reference
    (SegtranInitWeights.tie_qk does not match CrossMinceAttFeatTrans)."""
    ss = R.ref_shared()
    dims = [64, 64, 32]
    cases = {'bias2d': ('bias', (8, 12), [4, 2, 1], [1, 1, 2]), 'lsinu2d': ('lsinu', (7, 10), [3, 2, 1], [1, 1, 1]),
             'bias3d': ('bias', (4, 6, 5), [2, 1], [1, 3])}
    for tag, (pos_type, shape, scales, props) in cases.items():
        pd = len(shape)
        cfg = mk_shared_config(ss, 64, dims, 16, pos_dim=pd)
        cfg.use_squeezed_transformer, cfg.use_mince_transformer = False, True
        cfg.mince_scales, cfg.mince_channel_props = scales, props
        cfg.pos_code_type, cfg.pos_bias_radius, cfg.pos_code_weight = pos_type, 2, 0.8
        cfg.max_pos_size = (12,) * pd
        mod = R.quiet(ss.SegtranFusionEncoder, cfg, 'Fusion')
        prefix = 'voxel_fusion.'
        shapes = {prefix + k: tuple(v.shape) for k, v in mod.state_dict().items() if '.all_' not in k}
        sd = synth_state_dict(shapes)
        mod.load_state_dict({k[len(prefix):]: v for k, v in sd.items()}, strict=False)
        mod.eval()
        g = torch.Generator().manual_seed(23)
        N = int(np.prod(shape))
        X = torch.randn(2, N, 64, generator=g).requires_grad_(True)
        G = torch.randn(2, N, 32, generator=g)
        vmask = (torch.rand(2, N, 1, generator=g) > 0.2)
        pos = (O.gen_all_indices(shape).view(-1, pd).float() * 8).unsqueeze(0).repeat(2, 1, 1)
        Y = mod(X, pos, vmask, torch.Size(shape)); (Y * G).sum().backward()
        sdg = req(sd); Xo = X.detach().clone().requires_grad_(True)
        Yo = O.fusion_encoder(sdg, 'voxel_fusion', Xo, pos, vmask, dims, pos_code_weight=0.8, squeezed=False, pos_code_type=pos_type,
                              feat_shape=shape, mince_scales=scales, mince_channel_props=props)
        (Yo * G).sum().backward()
        close(Yo, Y, 1e-5, 'mince Y'); close(Xo.grad, X.grad, 1e-4, 'mince dX')
        rg = ref_param_grads(mod, prefix)
        og = {k: v.grad for k, v in sdg.items() if v.grad is not None}               # untied: every key has its own gradient
        arrs = dict(X=X, G=G, vmask=vmask, pos=pos, Y=Y, dX=X.grad, dims=np.array(dims), shape=np.array(shape),
                    scales=np.array(scales), props=np.array(props, dtype=np.float64))
        gscale = max(v.abs().max().item() for v in rg.values() if v is not None)
        for k, v in rg.items():
            if v is None:
                continue
            assert (og[k] - v).abs().max().item() <= 2e-4 * gscale, k
            arrs['grad:' + k] = v
        save('fusion_mince_' + tag, **arrs)


def case_polyformer():
    """PolyformerLayer (no-FFN 4-mode attention pair on a 2x-pooled map + residual), square and non-square maps
    (the reference maps the w-major token order back as (row, col): quirk N10)."""
    R._install_stubs()
    from networks.polyformer import Polyformer
    for tag, (C, H, W) in {'sq': (64, 12, 12), 'rect': (32, 8, 12)}.items():
        mod = R.quiet(Polyformer, C)
        layer = mod.polyformer_layers[0]
        layer.attractors.data = layer.attractors.data[:, :16]                 # 16 attractors keep the fixture small
        prefix = 'polyformer.polyformer_layers.0.'
        shapes = {prefix + k: tuple(v.shape) for k, v in layer.state_dict().items()}
        sd = synth_state_dict(shapes)
        layer.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
        mod.eval()                                                            # attention dropout (default 0.2) off
        g = torch.Generator().manual_seed(41)
        X = torch.randn(2, C, H, W, generator=g).abs().requires_grad_(True)   # post-ReLU like
        G = torch.randn(2, C, H, W, generator=g)
        Y = mod(X); (Y * G).sum().backward()
        sdg = req(sd); Xo = X.detach().clone().requires_grad_(True)
        Yo = O.polyformer_layer(sdg, prefix[:-1], Xo, 4); (Yo * G).sum().backward()
        close(Yo, Y, 1e-5, 'polyformer Y'); close(Xo.grad, X.grad, 1e-4, 'polyformer dX')
        rg = {prefix + k: p.grad for k, p in layer.named_parameters()}
        og = {k: v.grad for k, v in sdg.items() if v.grad is not None}
        arrs = dict(X=X, G=G, Y=Y, dX=X.grad)
        gscale = max(v.abs().max().item() for v in rg.values() if v is not None)
        for k, v in rg.items():
            if v is None:
                continue
            assert (og[k] - v).abs().max().item() <= 2e-4 * gscale, k
            arrs['grad:' + k] = v
        arrs['unused'] = np.array(sorted(k for k, v in rg.items() if v is None))
        save('polyformer_' + tag, **arrs)


def case_unet():
    """SURVEY 8(f) rank 4, host side: the U-Net the Polyformer layer is inserted into (networks/unet2d/unet_model.py, unet_parts.py), bilinear
    decoder, in training mode (batch statistics, running-statistics updates; the layer's attention dropout off) and in evaluation mode, on a
    map whose decoder needs the zero-pad branch of `Up` (H = 36 -> 18 -> 9 -> 4 -> 2: odd sizes on the way down)."""
    R._install_stubs()
    from networks.unet2d.unet_model import UNet
    from argparse import Namespace
    pargs = Namespace(polyformer_mode='source', num_attractors=16, num_modes=4, tie_qk_scheme='loose', qk_have_bias=True, pos_code_type='lsinu')
    net = R.quiet(UNet, 3, 2, True, pargs)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(77)
    X = torch.randn(2, 3, 36, 48, generator=g)
    G = torch.randn(2, 2, 36, 48, generator=g)
    arrs = dict(X=X, G=G)
    for mode in ('train', 'eval'):
        net.load_state_dict(sd)
        net.train(mode == 'train'); net.polyformer.eval()
        net.zero_grad()
        Xr = X.clone().requires_grad_(True)
        Y = net(Xr); (Y * G).sum().backward()
        sdg, running = req(sd), {}
        Xo = X.clone().requires_grad_(True)
        Yo = O.unet_forward(sdg, Xo, mode == 'train', True, 4, running); (Yo * G).sum().backward()
        close(Yo, Y, 2e-5, 'unet %s logits' % mode); close(Xo.grad, Xr.grad, 2e-4, 'unet %s dX' % mode)
        arrs.update({mode + ':Y': Y, mode + ':dX': Xr.grad})
        rg = {k: p.grad for k, p in net.named_parameters()}
        gscale = max(v.abs().max().item() for v in rg.values() if v is not None)
        for k, v in rg.items():
            if v is None:
                continue
            og = sdg[k].grad if sdg[k].grad is not None else torch.zeros_like(v)
            assert (og - v).abs().max().item() <= 3e-4 * gscale, (mode, k, (og - v).abs().max().item(), gscale)
            arrs['%s:grad:%s' % (mode, k)] = sample(v, 512 if mode == 'train' else 256)
        if mode == 'train':
            new = net.state_dict()
            for k, v in running.items():
                close(v, new[k], 1e-5, 'unet ' + k)
                arrs['train:stat:' + k] = new[k].clone()          # state_dict() hands out the live buffers
            arrs['unused'] = np.array(sorted(k for k, v in rg.items() if v is None))
    save('unet_poly', **arrs)


def case_unet_deconv():
    """UNet(bilinear=False): the transposed-convolution decoder (unet_parts.py:52-54), evaluation mode, no Polyformer layer."""
    R._install_stubs()
    from networks.unet2d.unet_model import UNet
    net = R.quiet(UNet, 3, 2, False, None)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd)
    net.eval()
    g = torch.Generator().manual_seed(78)
    X = torch.randn(2, 3, 32, 48, generator=g).requires_grad_(True)
    G = torch.randn(2, 2, 32, 48, generator=g)
    Y = net(X); (Y * G).sum().backward()
    arrs = dict(X=X.detach(), G=G, Y=Y.detach(), dX=X.grad)
    for k, p_ in net.named_parameters():
        arrs['grad:' + k] = sample(p_.grad, 256)
    save('unet_deconv', **arrs)


def case_discriminator():
    """Domain discriminator of the adversarial few-shot recipe (networks/discriminator.py + revgrad.py; train2d.py:876-926): training mode
    (batch statistics in the four BatchNorm layers), gradient reversal at the input."""
    R._install_stubs()
    from networks.discriminator import Discriminator
    net = Discriminator(8, num_classes=1, do_revgrad=True, num_base_chan=8)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd)
    net.train()
    g = torch.Generator().manual_seed(79)
    X = torch.randn(3, 8, 32, 32, generator=g).requires_grad_(True)
    G = torch.randn(3, 1, generator=g)
    Y = net(X); (Y * G).sum().backward()
    arrs = dict(X=X.detach(), G=G, Y=Y.detach(), dX=X.grad, keys=np.array(list(net.state_dict().keys())))
    for k, p_ in net.named_parameters():
        arrs['grad:' + k] = p_.grad.clone()
    for k, v in net.state_dict().items():
        if 'running' in k:
            arrs['stat:' + k] = v.clone()
    save('discriminator', **arrs)


class _ToyNet(torch.nn.Module):
    """stand-in for the segmentation network in the adversarial-loop fixture: 3 -> Cf feature map (kept in feature_maps[-1], as the networks of the
    recipe do) -> class scores; plain torch, so the reference's loop statements and the product's mirror can both drive it"""

    def __init__(self, Cf=8, nc=3):
        super().__init__()
        self.body = torch.nn.Conv2d(3, Cf, 3, padding=1)
        self.head = torch.nn.Conv2d(Cf, nc, 1)
        self.feature_maps, self.discriminator, self.recon = [], None, None

    def forward(self, x):
        f = torch.tanh(self.body(x))
        self.feature_maps = [f]
        return self.head(f)


def _ref_loop_block(start_marker, end_marker):
    """the statements of the reference's training loop between two marker lines (train2d.py is a script: its loop cannot be imported), dedented;
    compiled and executed at generation time only -- nothing of it is stored"""
    import textwrap
    src = open('/root/reference/code/train2d.py').read()
    a = src.index(start_marker)
    b = src.index(end_marker, a) + len(end_marker)
    return textwrap.dedent(src[src.rfind('\n', 0, a) + 1:b])


# The ADDA branch steps the discriminator IN PLACE between two backward passes over one graph: autograd saved parameter STORAGE, so the second pass
# mixes the old activations with the new weights.  The mirror (whose fused BatchNorm + LeakyReLU backward recomputes the activation slope from the
# live affine parameters) reproduces that artifact only to first order in the learning rate -- and BatchNorm's backward projection amplifies it.  So
# the branch is pinned in two runs: learning rate 1e-3 -> the discriminator's weights after its own step, the (pre-step) domain loss and the total;
# learning rate 0 -> every gradient of the second pass (control flow: zero_grad, backward with the graph retained, step, inverted labels).
ADDA_LRS = {'adda': 1e-3, 'adda0': 0.0}


def case_adversarial():
    """the few-shot / adversarial step of train2d.py.  The reference's OWN statements -- :1259-1286 (target and source
    feature maps, mixed batch, domain labels, discriminator, BCE, the ADDA discriminator step and label inversion) and :1314-1318 (the weighted
    total) -- are executed on a toy network + the reference Discriminator / BertAdam for --adv feat|mask x --adda on|off; stored: inputs, the toy
    network's weights, the domain loss, the total loss, every gradient after `loss.backward()` and the ADDA-updated discriminator weights."""
    import argparse
    R._install_stubs()
    from networks.discriminator import Discriminator
    BertAdam = R.ref_bertadam()
    dom_block = _ref_loop_block('            if args.adversarial_mode:\n                target_feat', '                domain_loss = 0\n')
    tot_block = _ref_loop_block('            supervised_loss = (1 - DICE_W) * total_ce_loss', '            loss = args.SUPERVISED_W * supervised_loss + unsup_loss\n')
    g = torch.Generator().manual_seed(97)
    image = torch.randn(2, 3, 32, 32, generator=g)                      # (supervised +) unsupervised target rows
    source = torch.randn(3, 3, 32, 32, generator=g)
    arrs = dict(image=image, source=source)
    torch.manual_seed(5)
    toy0 = _ToyNet()
    for k, v in toy0.state_dict().items():
        arrs['toy:' + k] = v.clone()
    for mode in ('feat', 'mask'):
        for variant in ('revgrad', 'adda', 'adda0'):
            adda = variant != 'revgrad'
            tag = '%s_%s' % (mode, variant)
            net = _ToyNet(); net.load_state_dict(toy0.state_dict())
            dis = Discriminator(8 if mode == 'feat' else 3, num_classes=1, do_revgrad=not adda, num_base_chan=8)
            dis.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in dis.state_dict().items()}))
            dis.train()
            net.discriminator = dis
            args = argparse.Namespace(adversarial_mode=mode, adda=adda, device='cpu', VCDR_W=0.0, ATTNCONSIST_W=0.0, DOMAIN_LOSS_W=0.002, RECON_W=0.0,
                                      SUPERVISED_W=1.0)
            mask_batch = torch.zeros(2, 3, 64, 64)                     # only its size is used here (:1271)
            outputs = torch.nn.functional.interpolate(net(image), size=mask_batch.shape[2:], mode='bilinear', align_corners=False)
            ns = dict(args=args, net=net, image_batch=image, source_image_batch=source, mask_batch=mask_batch, outputs_soft=torch.sigmoid(outputs),
                      unweighted_bce_loss_func=torch.nn.BCEWithLogitsLoss(), torch=torch, F=torch.nn.functional,
                      discriminator_optim=BertAdam(dis.parameters(), lr=ADDA_LRS[variant], warmup=-1, t_total=-1, weight_decay=0.0) if adda else None)
            exec(compile(dom_block, 'train2d.py:1259-1286', 'exec'), ns)
            ns.update(DICE_W=0.5, total_ce_loss=outputs.square().mean(), total_dice_loss=outputs.abs().mean(), vcdr_loss=0, attn_consist_loss=0, recon_loss=0)
            exec(compile(tot_block, 'train2d.py:1314-1318', 'exec'), ns)
            net.zero_grad()
            if not adda:
                dis.zero_grad()
            ns['loss'].backward()
            arrs[tag + ':domain_loss'] = ns['domain_loss'].detach(); arrs[tag + ':loss'] = ns['loss'].detach()
            if variant != 'adda':
                for k, p_ in net.named_parameters():
                    if not k.startswith('discriminator.'):
                        arrs[tag + ':toygrad:' + k] = p_.grad.clone()
                for k, p_ in dis.named_parameters():
                    arrs[tag + ':disgrad:' + k] = p_.grad.clone()
            else:
                for k, p_ in dis.named_parameters():
                    arrs[tag + ':disparam:' + k] = p_.detach().clone()   # after the discriminator's optimizer step
    save('adversarial', **arrs)


def case_posbias():
    ss = R.ref_shared()
    g = torch.Generator().manual_seed(13)
    m2 = R.quiet(ss.SlidingPosBiases2D, 2, 2, (8, 8))
    m2.biases.data = torch.randn(5, 5, generator=g)
    b2 = m2(torch.Size((5, 6)), 'cpu')
    G2 = torch.randn(30, 30, generator=g); (b2 * G2).sum().backward()
    t2 = m2.biases.detach().clone().requires_grad_(True)
    o2 = O.sliding_pos_biases(t2, (5, 6)); (o2 * G2).sum().backward()
    assert torch.equal(o2, b2) and torch.allclose(t2.grad, m2.biases.grad, atol=1e-6)
    m3 = R.quiet(ss.SlidingPosBiases3D, 3, 1, (4, 4, 4))
    m3.biases.data = torch.randn(3, 3, 3, generator=g)
    b3 = m3(torch.Size((3, 4, 2)), 'cpu')
    G3 = torch.randn(24, 24, generator=g); (b3 * G3).sum().backward()
    t3 = m3.biases.detach().clone().requires_grad_(True)
    o3 = O.sliding_pos_biases(t3, (3, 4, 2)); (o3 * G3).sum().backward()
    assert torch.equal(o3, b3) and torch.allclose(t3.grad, m3.biases.grad, atol=1e-6)
    save('posbias', table2=m2.biases, bias2=b2, G2=G2, dtable2=m2.biases.grad,
         table3=m3.biases, bias3=b3, G3=G3, dtable3=m3.biases.grad)


def case_effnet():
    net = R.ref_efficientnet_b4()
    prefix = 'backbone.'
    shapes = {prefix + k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth_state_dict(shapes)
    net.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    net.eval()
    g = torch.Generator().manual_seed(14)
    arrs = {}
    for tag, shp in {'a': (1, 3, 64, 64), 'b': (1, 3, 96, 64)}.items():
        x = torch.randn(*shp, generator=g)
        with torch.no_grad():
            ep = net.extract_endpoints(x)
            fo = O.effnet_b4_endpoints(sd, 'backbone', x)
        arrs['x_' + tag] = x
        for i in range(5):
            r = ep['reduction_%d' % (i + 1)]
            close(fo[i], r, 2e-5, 'effnet ep%d' % i)
            arrs['%s_shape%d' % (tag, i)] = np.array(r.shape)
            arrs['%s_ep%d' % (tag, i)] = r if tag == 'a' and i > 0 else sample(r, 16384)
    blocks, ep_idx = O.effnet_b4_blocks()
    arrs['pads'] = np.array([b['pad'] for b in blocks]); arrs['endpoint_blk'] = np.array(ep_idx)
    assert ep_idx == net.endpoint_blk_indices
    save('effnet_b4', **arrs)


def case_i3d():
    net = R.ref_i3d()
    prefix = 'backbone.'
    shapes = {prefix + k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth_state_dict(shapes)
    net.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    net.eval()
    g = torch.Generator().manual_seed(15)
    x = synth_image2d(1, 16 * 112, 1501, 112).view(1, 3, 16, 112, 112)
    with torch.no_grad():
        fd = net.extract_features(x)
        fo = O.i3d_features(sd, 'backbone', x)
    names = ['MaxPool3d_2a_3x3', 'Conv3d_2c_3x3', 'Mixed_3c', 'Mixed_4f', 'Mixed_5c']
    arrs = dict(x_sample=sample(x), x_seed=np.array(1501))
    for i, n in enumerate(names):
        close(fo[i], fd[n], 2e-5, 'i3d ' + n)
        arrs['shape%d' % i] = np.array(fd[n].shape)
        arrs['ep%d' % i] = sample(fd[n], 32768)
    save('i3d', **arrs)


GRAD_KEYS_2D = ['in_bn4b.weight', 'in_bn4b.bias', 'out_conv.weight', 'out_fpn_bridgeconv.weight', 'out_fpn12_conv.bias', 'out_gn3b.weight',
                'in_fpn34_conv.weight', 'in_gn4b.bias',
                'voxel_fusion.pos_code_layer.pos_coder.pos_fc.weight',
                'voxel_fusion.vfeat_norm_layers.0.weight',
                'voxel_fusion.pos_code_layer.pos_coder.biases',
                'voxel_fusion.pos_code_layers.0.pos_coder.biases', 'voxel_fusion.pos_code_layers.2.pos_coder.biases',
                'voxel_fusion.translayers.0.key.weight',
                'voxel_fusion.translayers.0.query.weight', 'voxel_fusion.translayers.0.out_trans.first_linear.weight',
                'voxel_fusion.translayers.0.out_trans.output.group_linear.weight',
                'voxel_fusion.translayers.0.attractors',
                'voxel_fusion.translayers.0.in_ator_trans.query.weight',
                'voxel_fusion.translayers.0.in_ator_trans.out_trans.first_linear.weight',
                'voxel_fusion.translayers.0.in_ator_trans.out_trans.first_norm_layer.weight',
                'voxel_fusion.translayers.0.ator_out_trans.query.weight',
                'voxel_fusion.translayers.0.ator_out_trans.query.bias',
                'voxel_fusion.translayers.0.ator_out_trans.out_trans.first_linear.weight',
                'voxel_fusion.translayers.0.ator_out_trans.out_trans.intermediate.shared_linear.weight',
                'voxel_fusion.translayers.0.ator_out_trans.out_trans.intermediate.shared_linear.bias',
                'voxel_fusion.translayers.0.ator_out_trans.out_trans.output.group_linear.weight',
                'voxel_fusion.translayers.0.ator_out_trans.out_trans.output.group_linear.bias',
                'voxel_fusion.translayers.0.ator_out_trans.out_trans.output.resout_norm_layer.weight',
                'voxel_fusion.translayers.0.ator_out_trans.out_trans.feat_softaggr.feat2score.weight',
                'backbone._conv_stem.weight', 'backbone._bn0.weight',
                'backbone._blocks.0._depthwise_conv.weight', 'backbone._blocks.5._se_reduce.weight',
                'backbone._blocks.9._expand_conv.weight', 'backbone._blocks.21._project_conv.weight',
                'backbone._blocks.31._bn1.bias', 'backbone._conv_head.weight']


def run_seg2d(tag, tl, compress, dims, A, B, S, train, fusion_kw=None, task='fundus', tied=True, oracle_kw=None, **over):
    net = R.ref_segtran2d(num_classes=3 if task == 'fundus' else 2, num_attractors=A, num_translayers=tl, compress=compress, dropout_prob=0, **over)
    sd = load_synth(net)
    if train:
        net.train()
        net.backbone._global_params = net.backbone._global_params._replace(drop_connect_rate=0.0)
    else:
        net.eval()
    g = torch.Generator().manual_seed(16)
    x = torch.randn(B, 3, S, S, generator=g)
    mask = synth_fundus_mask(B, S, 1338)
    if task == 'polyp':                                   # single 0/255 channel tiled x3 (datasets2d.py:313-327), mapped by the REFERENCE function
        mask = mask[:, :1].repeat(1, 3, 1, 1)
        ref_map = _ref_functions('dataloaders/datasets2d.py', ['polyp_map_mask'])['polyp_map_mask']
        nhot = O.polyp_map_mask(mask)
        assert torch.equal(nhot, ref_map(mask.float()).float()) or torch.equal(nhot, ref_map(mask).float())
        pw = O.bce_pos_weight([0., 1.])
    else:
        nhot = O.fundus_map_mask(mask)
        pw = O.bce_pos_weight([0., 1., 2.])
    y = R.quiet(net, x)
    loss, ce, dice, _ = O.seg_loss(y, nhot, pw)          # composition restated; pinned separately by case_loss
    loss.backward()
    sdg = req(sd)
    yo = O.segtran2d_forward(sdg, x, dims, training=train, fusion_kw=fusion_kw, **(oracle_kw or {}))
    lo = O.seg_loss(yo, nhot, pw)[0]; lo.backward()
    close(yo, y, 2e-5, tag + ' logits')
    og = oracle_grads(sdg) if tied else {k: v.grad for k, v in sdg.items() if v.grad is not None}
    rg = dict(net.named_parameters())
    gscale = max(p.grad.abs().max().item() for p in rg.values() if p.grad is not None)
    arrs = dict(x=x, mask=mask, logits=y, labels=(y > 0), loss=loss.detach(), margin=y.abs().min().detach(),
                dims=np.array(dims), A=np.array(A), train=np.array(int(train)), nhot=nhot.to(torch.uint8))
    for k in GRAD_KEYS_2D:
        if k not in rg:
            continue
        gr = rg[k].grad
        assert (og[k] - gr).abs().max().item() <= 3e-4 * gscale, (k, (og[k] - gr).abs().max().item(), gscale)
        arrs['grad:' + k] = sample(gr)
    arrs['gscale'] = np.array(gscale)
    unused = sorted(k for k, p in rg.items() if p.grad is None)
    arrs['unused'] = np.array(unused)
    zero = sorted(k for k, p in rg.items() if p.grad is not None and p.grad.abs().max() == 0)
    arrs['zero_grad'] = np.array(zero)
    save(tag, **arrs)


def case_seg2d():
    run_seg2d('seg2d_cfg2_eval', 3, (1, 1, 2, 2), [1792, 1792, 896, 448], 32, 2, 64, False)
    run_seg2d('seg2d_cfg1_eval', 1, (1, 1), [1792, 1792], 32, 2, 64, False)
    run_seg2d('seg2d_cfg2_train', 3, (1, 1, 2, 2), [1792, 1792, 896, 448], 32, 2, 64, True)
    # --nosqueeze --pos bias --posr 3 (8x8 tokens: pairs inside and outside the radius)
    run_seg2d('seg2d_cfg1_nosq_bias_train', 1, (1, 1), [1792, 1792], 32, 2, 64, True,
              fusion_kw=dict(squeezed=False, pos_code_type='bias', pos_code_weight=1.0),
              use_squeezed_transformer=False, pos_code_type='bias', pos_bias_radius=3)


GRAD_KEYS_3D = ['in_bridge_to3.weight', 'out_conv3d.weight', 'out_fpn_bridgeconv3d.weight', 'out_fpn12_conv3d.weight',
                'out_gn2b.weight', 'in_fpn34_conv.weight', 'in_gn4b.weight',
                'voxel_fusion.pos_code_layer.pos_coder.pos_fc.weight',
                'voxel_fusion.translayers.0.attractors',
                'voxel_fusion.translayers.0.ator_out_trans.query.weight',
                'voxel_fusion.translayers.0.ator_out_trans.out_trans.output.group_linear.weight',
                'backbone.Conv3d_1a_7x7.conv3d.weight', 'backbone.Conv3d_1a_7x7.bn.weight',
                'backbone.Conv3d_2c_3x3.conv3d.weight', 'backbone.Mixed_3b.b1b.conv3d.weight',
                'backbone.Mixed_4d.b2b.conv3d.weight', 'backbone.Mixed_4f.b3b.bn.bias',
                'backbone.Mixed_5c.b0.conv3d.weight']


def case_seg2d_mince():
    """--nosqueeze --mince --mincescales 4,2,1 --minceprops 1,1,2 --pos bias --posr 2 (untied query/key) on a 96 x 96 input ->
    12 x 12 tokens -> 3 x 3 / 6 x 6 / 12 x 12 grids"""
    run_seg2d('seg2d_cfg1_mince_train', 1, (1, 1), [1792, 1792], 32, 2, 96, True, tied=False,
              fusion_kw=dict(squeezed=False, pos_code_type='bias', pos_code_weight=1.0, mince_scales=[4, 2, 1], mince_channel_props=[1, 1, 2]),
              use_squeezed_transformer=False, use_mince_transformer=True, mince_scales=[4, 2, 1], mince_channel_props=[1, 1, 2],
              pos_code_type='bias', pos_bias_radius=2)


def case_seg2d_inbn():
    """--inbn: BatchNorm2d (batch statistics, train mode) instead of GroupNorm in the in-FPN"""
    run_seg2d('seg2d_cfg1_inbn_train', 1, (1, 1), [1792, 1792], 32, 2, 64, True, oracle_kw=dict(in_fpn_use_bn=True), in_fpn_use_bn=True)


def case_seg2d_polyp():
    """cfg3 flags (polyp: 2 classes, 3 layers with compression) at 88 x 88 -> an 11 x 11 token grid (odd, not a power of two)"""
    run_seg2d('seg2d_cfg3_polyp_train', 3, (1, 1, 2, 2), [1792, 1792, 896, 448], 32, 1, 88, True, task='polyp')


def case_seg3d():
    for tag, train, tl, comp in (('seg3d_cfg4_eval', False, 1, (1, 1)), ('seg3d_cfg4_train', True, 1, (1, 1)), ('seg3d_cfg5_eval', False, 2, (1, 1, 1))):
        if len(sys.argv) > 2 and tag not in sys.argv[2:]:
            continue
        A = 64
        net = R.ref_segtran3d(num_attractors=A, num_translayers=tl, compress=comp, dropout_prob=0)
        sd = load_synth(net)
        net.train() if train else net.eval()
        x, lab = synth_brats(1, 112, 112, 16, 1337)
        nhot = O.brats_map_label(lab)
        pw = O.bce_pos_weight([0., 3., 1., 1.75])
        y = R.quiet(net, x)
        loss = O.seg_loss(y, nhot, pw)[0]; loss.backward()
        sdg = req(sd)
        yo = O.segtran3d_forward(sdg, x, [1024] * (tl + 1), training=train)
        lo = O.seg_loss(yo, nhot, pw)[0]; lo.backward()
        close(yo, y, 3e-5, tag + ' logits')
        og = oracle_grads(sdg); rg = dict(net.named_parameters())
        gscale = max(p.grad.abs().max().item() for p in rg.values() if p.grad is not None)
        arrs = dict(x_sample=sample(x), logits=sample(y, 65536), labels=np.packbits((y > 0).numpy()),
                    loss=loss.detach(), margin=y.abs().min().detach(), A=np.array(A), train=np.array(int(train)), tl=np.array(tl))
        for k in GRAD_KEYS_3D:
            gr = rg[k].grad
            assert (og[k] - gr).abs().max().item() <= 3e-4 * gscale, (k, (og[k] - gr).abs().max().item(), gscale)
            arrs['grad:' + k] = sample(gr)
        arrs['gscale'] = np.array(gscale)
        arrs['unused'] = np.array(sorted(k for k, p in rg.items() if p.grad is None))
        # fp64 referee: the same reference modules in double precision.  Batch-1 train-mode BatchNorm over 49
        # samples per channel amplifies fp32 summation-order noise; the referee says how far the fp32 CPU reference itself is from the
        # exact result, which bounds what can be asked of any other fp32 implementation.
        net64 = R.ref_segtran3d(num_attractors=A, num_translayers=tl, compress=comp, dropout_prob=0)
        net64.load_state_dict(sd); net64 = net64.double(); _force_double_inputs(net64)
        net64.train() if train else net64.eval()
        y64 = R.quiet(net64, x.double())
        l64 = O.seg_loss(y64, nhot.double(), pw.double())[0]; l64.backward()
        rg64 = dict(net64.named_parameters())
        arrs['logits64'] = sample(y64, 65536); arrs['loss64'] = l64.detach()
        arrs['ref32_vs_64_logit_err'] = (y.double() - y64).abs().max().detach()
        worst = 0.0
        for k in GRAD_KEYS_3D:
            arrs['grad64:' + k] = sample(rg64[k].grad)
            worst = max(worst, (rg64[k].grad - rg[k].grad.double()).abs().max().item() / gscale)
        arrs['ref32_vs_64_grad_err'] = np.array(worst)
        print('    %s: fp32 reference vs fp64 referee: logits %.2e, gradients %.2e of gscale' % (tag, arrs['ref32_vs_64_logit_err'].item(), worst))
        save(tag, **arrs)


def _ref_functions(relpath, names, extra=None):
    """exec only the named top-level functions of a reference file (its module imports cv2 / imgaug / medpy, which this image
    lacks) in a namespace with torch on the CPU; the reference file itself is untouched."""
    import ast, math
    src = open(os.path.join('/root/reference/code', relpath)).read()
    tree = ast.parse(src)
    ns = dict(torch=R.cpu_torch(), np=np, F=torch.nn.functional, math=math)
    ns.update(extra or {})
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), relpath, 'exec'), ns)
    return ns


def case_eval():
    """sliding-window evaluation.  2-D: reference test_single_batch driving the reference Segtran2d (eval mode,
    64x64 patches) over (a) an image larger than the window with overlapping strides, (b) an image smaller than the window (zero
    padding) whose windows are resized 96 -> 64 for the network and back.  3-D: reference test_single_case on Segtran3d."""
    d2 = _ref_functions('dataloaders/datasets2d.py', ['harden_segmap2d'])
    t2 = _ref_functions('test_util2d.py', ['test_single_batch', 'calc_dice'], dict(harden_segmap2d=d2['harden_segmap2d']))
    net = R.ref_segtran2d(num_attractors=32, num_translayers=1, compress=(1, 1), dropout_prob=0)
    sd = load_synth(net); net.eval()
    dims = [1792, 1792]
    g = torch.Generator().manual_seed(31)
    arrs = {}
    for tag, (shape, orig, patch, stride) in {'a': ((2, 3, 80, 96), (64, 64), (64, 64), (32, 48)),
                                              'b': ((1, 3, 80, 100), (96, 96), (64, 64), (96, 96))}.items():
        x = torch.randn(*shape, generator=g)
        hard, soft = R.quiet(t2['test_single_batch'], net, x, orig, patch, stride, 'fundus', 3, 'segtran')
        oh, osf = O.test_single_batch(lambda p: O.segtran2d_forward(sd, p, dims), x, orig, patch, stride, 3)
        close(osf, soft, 1e-5, 'eval2d soft ' + tag)
        safe = (soft - 0.5).abs() > 1e-5
        assert torch.equal(oh[:, 1:][safe[:, 1:]], hard[:, 1:][safe[:, 1:]])
        arrs.update({'x_' + tag: x, 'soft_' + tag: soft, 'hard_' + tag: hard.to(torch.uint8),
                     'cfg_' + tag: np.array(list(orig) + list(patch) + list(stride))})
    gt = (torch.rand(2, 3, 80, 96, generator=g) > 0.5).float()
    arrs['gt'] = gt.to(torch.uint8)
    arrs['dice'] = torch.stack([t2['calc_dice'](arrs['hard_a'][:, c].float(), gt[:, c]) for c in range(3)], dim=1)
    assert torch.allclose(arrs['dice'], torch.stack([O.calc_dice(arrs['hard_a'][:, c].float(), gt[:, c]) for c in range(3)], dim=1))
    arrs['A'] = np.array(32)
    save('eval2d', **arrs)

    d3 = _ref_functions('dataloaders/datasets3d.py', ['make_brats_pred_consistent', 'brats_inv_map_label', 'harden_segmap3d'])
    t3 = _ref_functions('test_util3d.py', ['test_single_case'], d3)
    net3 = R.ref_segtran3d(num_attractors=32, dropout_prob=0)
    sd3 = load_synth(net3); net3.eval()
    vol = synth_brats(1, 112, 168, 16, 1441)[0][0]                   # [4, 112, 168, 16]: two overlapping windows along W
    hard3, soft3 = R.quiet(t3['test_single_case'], net3, vol, (112, 112, 16), (112, 112, 16), 2, 56, 16, 'brats', 'segtran', 4)
    oh3, os3 = O.test_single_case(lambda p: O.segtran3d_forward(sd3, p, [1024, 1024]), vol, (112, 112, 16), (112, 112, 16), 2, 56, 16)
    close(os3, soft3, 1e-5, 'eval3d soft')
    safe = (soft3 - 0.5).abs() > 1e-5
    assert torch.equal(oh3[1:][safe[1:]], hard3[1:][safe[1:]])
    p = torch.rand(4, 5, 6, 7, generator=g)
    cons = {k: d3['make_brats_pred_consistent'](p, k) for k in (True, False)}
    assert all(torch.equal(cons[k], O.make_brats_pred_consistent(p, k)) for k in cons)
    pc = cons[False].clone(); pc[1] = torch.minimum(pc[1], pc[3]); pc[3] = torch.minimum(pc[3], pc[2])     # ET <= TC <= WT: no negative
    inv = d3['brats_inv_map_label'](pc)
    assert torch.equal(inv, O.brats_inv_map_label(pc))
    save('eval3d', soft=sample(soft3, 65536), hard=torch.from_numpy(np.packbits(hard3.numpy().astype(bool))), seed=np.array(1441),
         probs=p, cons_true=cons[True], cons_false=cons[False], inv_in=pc, inv=inv, harden3d=d3['harden_segmap3d'](p).to(torch.uint8),
         A=np.array(32))


def case_loss():
    """train2d.py:1219-1242,1314-1318 composed from the reference's own dice_loss_indiv + torch BCE."""
    dice_ref = R.ref_dice()
    g = torch.Generator().manual_seed(17)
    arrs = {}
    for tag, shp, bw in (('2d', (2, 3, 24, 20), [0., 1., 2.]), ('3d', (2, 4, 10, 12, 6), [0., 3., 1., 1.75])):
        logits = (torch.randn(*shp, generator=g) * 2).requires_grad_(True)
        nc = shp[1]
        mask = (torch.rand(*shp, generator=g) > 0.6).float()
        pw = torch.tensor(bw); pw = pw * (nc - 1) / pw.sum()
        perm = (0,) + tuple(range(2, len(shp))) + (1,)
        ce = torch.nn.BCEWithLogitsLoss(pos_weight=pw)(logits.permute(*perm), mask.permute(*perm))
        soft = torch.sigmoid(logits)
        cw = torch.ones(nc); cw[0] = 0; cw /= cw.sum()
        dice = 0
        for c in range(1, nc):
            dice = dice + dice_ref(soft[:, c], mask[:, c]) * cw[c]
        loss = 0.5 * ce + 0.5 * dice
        loss.backward()
        lo = logits.detach().clone().requires_grad_(True)
        l2, ce2, d2, _ = O.seg_loss(lo, mask, pw); l2.backward()
        assert torch.allclose(l2, loss, atol=1e-6) and torch.allclose(lo.grad, logits.grad, atol=1e-7)
        arrs.update({'logits' + tag: logits, 'mask' + tag: mask, 'pw' + tag: pw, 'loss' + tag: loss,
                     'ce' + tag: ce, 'dice' + tag: dice, 'dlogits' + tag: logits.grad})
    save('loss', **arrs)


def case_bertadam():
    BertAdam = R.ref_bertadam()
    g = torch.Generator().manual_seed(18)
    shapes = [(7, 5), (33,), (4, 3, 2), (6,)]
    p0 = [torch.randn(*s, generator=g) for s in shapes]
    grads = [[torch.randn(*s, generator=g) * sc for s in shapes] for sc in (1.0, 0.01, 5.0, 0.1)]
    params = [torch.nn.Parameter(p.clone()) for p in p0]
    wds = [1e-4, 1e-5, 0.0, 1e-4]
    groups = [dict(params=[params[0], params[3]], weight_decay=1e-4, lr=2e-4),
              dict(params=[params[1]], weight_decay=1e-5, lr=2e-4),
              dict(params=[params[2]], weight_decay=0.0, lr=2e-4)]
    opt = BertAdam(groups, warmup=0.25, t_total=8, weight_decay=1e-4)
    po = [p.clone() for p in p0]; state = [dict() for _ in po]
    arrs = {'p0_%d' % i: p for i, p in enumerate(p0)}
    for step, gs in enumerate(grads):
        gs = [x.clone() for x in gs]
        for p, gr in zip(params, gs):
            p.grad = gr.clone()
        params[3].grad = None                                   # N3: a parameter that never gets a grad
        torch.nn.utils.clip_grad_norm_([p for p in params], 0.1)
        opt.step()
        og = [x.clone() for x in gs]; og[3] = None
        O.global_clip_([x for x in og if x is not None], 0.1)
        O.bertadam_step(po, og, state, 2e-4, wds, 0.25, 8)
        for i in range(4):
            assert torch.allclose(po[i], params[i].data, atol=1e-7), (step, i)
            arrs['g%d_%d' % (step, i)] = gs[i]
            arrs['p%d_%d' % (step + 1, i)] = params[i].data.clone()
    save('bertadam', **arrs)


FULL_GRAD_KEYS_2D = ['out_conv.weight', 'out_conv.bias', 'out_fpn_bridgeconv.weight', 'out_fpn12_conv.weight', 'out_fpn23_conv.weight', 'out_gn2b.weight',
                     'out_gn3b.bias', 'in_fpn34_conv.weight', 'in_gn4b.weight',
                     'voxel_fusion.pos_code_layer.pos_coder.pos_fc.weight', 'voxel_fusion.vfeat_norm_layers.0.weight', 'voxel_fusion.vfeat_norm_layers.2.bias',
                     'voxel_fusion.translayers.0.attractors', 'voxel_fusion.translayers.0.in_ator_trans.query.weight',
                     'voxel_fusion.translayers.0.in_ator_trans.query.bias',
                     'voxel_fusion.translayers.0.in_ator_trans.out_trans.first_linear.weight',
                     'voxel_fusion.translayers.0.in_ator_trans.out_trans.first_norm_layer.weight',
                     'voxel_fusion.translayers.0.ator_out_trans.query.weight', 'voxel_fusion.translayers.0.ator_out_trans.query.bias',
                     'voxel_fusion.translayers.0.ator_out_trans.out_trans.first_linear.weight',
                     'voxel_fusion.translayers.0.ator_out_trans.out_trans.intermediate.shared_linear.weight',
                     'voxel_fusion.translayers.0.ator_out_trans.out_trans.intermediate.shared_linear.bias',
                     'voxel_fusion.translayers.0.ator_out_trans.out_trans.output.group_linear.weight',
                     'voxel_fusion.translayers.0.ator_out_trans.out_trans.output.resout_norm_layer.weight',
                     'voxel_fusion.translayers.0.ator_out_trans.out_trans.feat_softaggr.feat2score.weight',
                     'voxel_fusion.translayers.1.attractors', 'voxel_fusion.translayers.1.ator_out_trans.query.weight',
                     'voxel_fusion.translayers.1.ator_out_trans.out_trans.output.group_linear.weight',
                     'voxel_fusion.translayers.2.in_ator_trans.query.weight',
                     'voxel_fusion.translayers.2.ator_out_trans.out_trans.first_linear.weight',
                     'voxel_fusion.translayers.2.ator_out_trans.out_trans.output.group_linear.weight',
                     'backbone._conv_stem.weight', 'backbone._bn0.weight', 'backbone._blocks.0._depthwise_conv.weight',
                     'backbone._blocks.2._expand_conv.weight', 'backbone._blocks.6._se_reduce.weight', 'backbone._blocks.10._bn1.bias',
                     'backbone._blocks.22._project_conv.weight', 'backbone._blocks.31._depthwise_conv.weight', 'backbone._conv_head.weight']
FULL_GRAD_KEYS_3D = GRAD_KEYS_3D + ['out_conv3d.bias', 'out_fpn23_conv3d.weight', 'out_gn3b.weight',
                                    'voxel_fusion.vfeat_norm_layers.0.weight', 'voxel_fusion.translayers.0.in_ator_trans.query.weight',
                                    'voxel_fusion.translayers.0.in_ator_trans.out_trans.first_linear.weight',
                                    'voxel_fusion.translayers.0.ator_out_trans.out_trans.first_linear.weight',
                                    'voxel_fusion.translayers.0.ator_out_trans.out_trans.intermediate.shared_linear.weight',
                                    'voxel_fusion.translayers.0.ator_out_trans.out_trans.feat_softaggr.feat2score.weight',
                                    'voxel_fusion.translayers.1.attractors',
                                    'voxel_fusion.translayers.1.ator_out_trans.out_trans.output.group_linear.weight',
                                    'backbone.Mixed_3c.b2b.conv3d.weight', 'backbone.Mixed_4b.b1a.conv3d.weight', 'backbone.Mixed_5b.b2b.bn.weight']
NEAR0 = 1e-3          # reference logits with |y| < NEAR0 are stored (index + fp32 value + fp64 value): any label margin <= NEAR0 can be applied later


def _force_double_inputs(net):
    """fp64 referee runs: the reference creates a few fp32 tensors on the fly (`.float()` coordinates, scale factors); cast every
    fp32 tensor argument of every sub-module to fp64 on entry (forward pre-hooks; the reference code itself is untouched)."""
    def hook(mod, args):
        return tuple(a.double() if isinstance(a, torch.Tensor) and a.dtype == torch.float32 else a for a in args)
    for m in net.modules():
        m.register_forward_pre_hook(hook)


def _full_one(tag, dim, build, x, nhot, pw, dims, grad_keys, referee64=True, train_only=False):
    """One BASELINE shape at batch 1: (a) eval forward -> packed label bits of EVERY logit + the near-zero logits, (b) one
    dropout-free train-mode step (batch statistics in every BatchNorm, drop_connect off) -> loss + sampled parameter gradients,
    both from the real reference in fp32, plus the same two runs of the same reference modules in fp64 as referee."""
    import time
    t0 = time.time()
    net = build(); sd = load_synth(net)
    fwd_o = O.segtran2d_forward if dim == 2 else O.segtran3d_forward
    arrs = {}
    if not train_only:
        net.eval()
        with torch.no_grad():
            y = R.quiet(net, x)
            yo = fwd_o(sd, x, dims)
        close(yo, y, 5e-5, tag + ' eval logits (oracle)')
        arrs = dict(shape=np.array(y.shape), absmax=y.abs().max(), logits=sample(y, 65536)[::4], labels=np.packbits((y > 0).numpy().reshape(-1)))
        flat = y.reshape(-1)
        near = torch.nonzero(flat.abs() < NEAR0).reshape(-1)
        arrs['near_idx'] = near.to(torch.int32); arrs['near_val'] = flat[near]
        print('    %s eval fwd done %.0fs, %d logits with |y| < %g, min |y| %.2e' % (tag, time.time() - t0, near.numel(), NEAR0, flat.abs().min().item()))
    # train step
    net.train()
    if dim == 2:
        net.backbone._global_params = net.backbone._global_params._replace(drop_connect_rate=0.0)
    yt = R.quiet(net, x)
    loss = O.seg_loss(yt, nhot, pw)[0]; loss.backward()
    rg = dict(net.named_parameters())
    gscale = max(p.grad.abs().max().item() for p in rg.values() if p.grad is not None)
    loss32 = loss.item()
    arrs.update(train_logits=sample(yt, 65536)[::4], loss=loss.detach(), gscale=np.array(gscale))
    keys = [k for k in grad_keys if k in rg and rg[k].grad is not None]
    for k in keys:
        arrs['grad:' + k] = sample(rg[k].grad)
    arrs['unused'] = np.array(sorted(k for k, p in rg.items() if p.grad is None))
    print('    %s train step done %.0fs (loss %.5f, gscale %.3e, %d gradients stored)' % (tag, time.time() - t0, loss.item(), gscale, len(keys)))
    save(tag, **arrs)                 # the fp32 part is complete: keep it even if the (slow) referee below is interrupted
    if train_only:
        arrs.update(shape=np.array(yt.shape), absmax=yt.detach().abs().max())
        # The plain fp64 train step does not fit the 64-GiB build container at these batches (cfg2 at batch 6 and cfg4 at batch 4 were stopped by the OOM killer / the
        # watchdog).  Here the referee runs with ACTIVATION RECOMPUTATION: the forward of every backbone block (MBConv block / I3D end point) and of the transformer
        # layers is wrapped in torch.utils.checkpoint on the module INSTANCES (the reference source is untouched; dropout and drop_connect are off, so the recomputed
        # forward is the same computation; BatchNorm's running statistics are updated twice, which no output depends on).  Same numbers as the plain fp64 step.
        del yt, loss
        import gc; gc.collect()
        from torch.utils.checkpoint import checkpoint
        net64 = build(); net64.load_state_dict(sd); net64 = net64.double()
        _force_double_inputs(net64)
        if dim == 2:
            net64.backbone._global_params = net64.backbone._global_params._replace(drop_connect_rate=0.0)
            blocks = list(net64.backbone._blocks)
        else:
            blocks = [m for n, m in net64.backbone._modules.items() if n in getattr(net64.backbone, 'VALID_ENDPOINTS', ()) or n.startswith(('Conv3d', 'Mixed', 'MaxPool'))]
        assert len(blocks) >= 10, len(blocks)
        for m in blocks:
            m.forward = (lambda f: (lambda *a, **k: checkpoint(f, *a, use_reentrant=False, **k)))(m.forward)
        net64.train()
        yt64 = R.quiet(net64, x.double())
        l64 = O.seg_loss(yt64, nhot.double(), pw.double())[0]; l64.backward()
        rg64 = dict(net64.named_parameters())
        arrs['loss64'] = l64.detach(); arrs['train_logits64'] = sample(yt64, 65536)[::4]
        worst = 0.0
        for k in keys:
            arrs['grad64:' + k] = sample(rg64[k].grad)
            worst = max(worst, (arrs['grad64:' + k].double() - arrs['grad:' + k].double()).abs().max().item() / gscale)
        arrs['ref32_vs_64_grad_err'] = np.array(worst)
        print('    %s fp64 referee (recomputing block activations) done %.0fs: fp32 reference vs fp64: loss %.6f vs %.6f, gradients %.2e of gscale' % (tag, time.time() - t0, loss32, l64.item(), worst))
        save(tag, **arrs)
        return
    if referee64:
        # the SAME reference modules in double precision: tells which side is off when fp32 results disagree
        del net, yt, loss
        import gc; gc.collect()
        net64 = build(); net64.load_state_dict(sd); net64 = net64.double()
        _force_double_inputs(net64)
        if dim == 2:
            net64.backbone._global_params = net64.backbone._global_params._replace(drop_connect_rate=0.0)
        if not train_only:
            net64.eval()
            with torch.no_grad():
                y64 = R.quiet(net64, x.double())
            f64 = y64.reshape(-1)
            arrs['near_val64'] = f64[near]
            arrs['logits64'] = sample(y64, 65536)[::4]
            arrs['ref32_vs_64_logit_err'] = (y.double() - y64).abs().max()
            arrs['ref32_label_flips_vs_64'] = np.array(int(((y > 0) != (y64 > 0)).sum()))
        net64.train()
        yt64 = R.quiet(net64, x.double())
        l64 = O.seg_loss(yt64, nhot.double(), pw.double())[0]; l64.backward()
        rg64 = dict(net64.named_parameters())
        arrs['loss64'] = l64.detach(); arrs['train_logits64'] = sample(yt64, 65536)[::4]
        worst = 0.0
        for k in keys:
            arrs['grad64:' + k] = sample(rg64[k].grad)
            worst = max(worst, (rg64[k].grad - rg[k].grad.double()).abs().max().item() / gscale)
        arrs['ref32_vs_64_grad_err'] = np.array(worst)
        print('    %s fp64 referee done %.0fs: fp32 reference vs fp64: logits %.2e, %d label flips, gradients %.2e of gscale'
              % (tag, time.time() - t0, float(arrs.get('ref32_vs_64_logit_err', float('nan'))), int(arrs.get('ref32_label_flips_vs_64', -1)), worst))
    save(tag, **arrs)


def case_fullshape():
    """BASELINE.json shapes at batch 1: label bits of the whole map, full-size gradients, fp64 referee.
    Sub-cases: full_cfg2 full_cfg3 full_cfg4 full_cfg5 (python make_golden.py fullshape full_cfg4 ...).
    r06: full_cfg2_b6 / full_cfg4_b4 = the exact batches bench.py times (reference train2d.py:1147-1245 at --bs 6, train3d.py at --bs 4).
    At those batches only the dropout-free TRAIN step of the fp32 reference fits the 64-GiB build container (the eval pass + oracle pass of cfg2 at batch 6 grew
    past 58 GiB and was stopped; the fp64 referee needs twice the fp32 step): `train_only` fixtures hold loss, sampled train-mode logits and the sampled gradients
    (fp32 reference and, where the container's memory allows the fp64 train step, the fp64 referee columns), no label bits.  They are generated
    explicitly (`python make_golden.py fullshape full_cfg2_b6 full_cfg4_b4`, ~1 h of 8 cores), not by the default case list."""
    want = [a for a in sys.argv[2:] if a.startswith('full_')] or ['full_cfg2', 'full_cfg3', 'full_cfg4', 'full_cfg5', 'full_cfg2_b2', 'full_cfg4_b2']
    for tag in want:
        # *_b2: the same shapes at BATCH 2 -- train-mode BatchNorm statistics over two samples, (B, M) batch strides of
        # the attention GEMMs, and B x N token rows above the product's re-association gate at cfg4 (2 x 2352 rows)
        mb = re.search(r'_b(\d+)$', tag)               # *_b6 / *_b4: the batches bench.py times (cfg2 bs 6, cfg4 bs 4)
        B = int(mb.group(1)) if mb else 1
        base = tag[:mb.start()] if mb else tag
        if base in ('full_cfg2', 'full_cfg3'):
            S, task = (512, 'fundus') if base == 'full_cfg2' else (352, 'polyp')
            x = synth_image2d(B, S, 1337)
            mask = synth_fundus_mask(B, S, 1338)
            if task == 'polyp':
                mask = mask[:, :1].repeat(1, 3, 1, 1)
                nhot, pw = O.polyp_map_mask(mask), O.bce_pos_weight([0., 1.])
            else:
                nhot, pw = O.fundus_map_mask(mask), O.bce_pos_weight([0., 1., 2.])
            nc = 3 if task == 'fundus' else 2
            _full_one(tag, 2, lambda: R.ref_segtran2d(num_classes=nc, dropout_prob=0), x, nhot.float(), pw, [1792, 1792, 896, 448], FULL_GRAD_KEYS_2D, train_only=B >= 4)
        else:
            size, tl = ((112, 112, 96), 1) if base == 'full_cfg4' else ((128, 128, 128), 2)
            x, lab = synth_brats(B, *size, 1337)
            nhot, pw = O.brats_map_label(lab), O.bce_pos_weight([0., 3., 1., 1.75])
            _full_one(tag, 3, lambda: R.ref_segtran3d(num_translayers=tl, compress=(1,) * (tl + 1), dropout_prob=0), x, nhot.float(), pw,
                      [1024] * (tl + 1), FULL_GRAD_KEYS_3D, train_only=B >= 4)


def case_augment():
    """In-step label maps (all three tasks, fundus with and without --exclusive) and the 3-D RandomResizedCrop (--randscale), produced by
    the REFERENCE's own functions (datasets2d.py:90-139, 200-223; datasets3d.py:16-40, 611-665); the oracle restatements are asserted equal."""
    f2 = _ref_functions('dataloaders/datasets2d.py', ['fundus_map_mask', 'polyp_map_mask'])
    f3 = _ref_functions('dataloaders/datasets3d.py', ['brats_map_label', 'RandomResizedCrop'])
    g = torch.Generator().manual_seed(31)
    fm = synth_fundus_mask(2, 40, 77)
    fm[:, 1] = torch.where(torch.rand(2, 40, 40, generator=g) < 0.1, torch.tensor(255, dtype=torch.uint8), fm[:, 1])   # cup pixels outside the disc too
    lab = torch.randint(0, 4, (2, 6, 7, 5), generator=g)
    arrs = dict(fundus_in=fm, brats_in=lab.to(torch.int8))
    for ex in (False, True):
        ref = f2['fundus_map_mask'](fm, ex).float()
        assert torch.equal(O.fundus_map_mask(fm, ex), ref)
        arrs['fundus_excl%d' % ex] = ref.to(torch.uint8)
    ref = f2['polyp_map_mask'](fm[:, :1].repeat(1, 3, 1, 1)).float()
    assert torch.equal(O.polyp_map_mask(fm[:, :1].repeat(1, 3, 1, 1)), ref)
    arrs['polyp'] = ref.to(torch.uint8)
    ref = f3['brats_map_label'](lab, False).float()
    assert torch.equal(O.brats_map_label(lab), ref)
    arrs['brats'] = ref.to(torch.uint8)
    # RandomResizedCrop: seeds chosen below cover scale < 1 (pad) and scale > 1 (crop); the RNG is torch's global CPU generator
    vol = torch.randn(1, 2, 14, 16, 10, generator=g)
    mask = f3['brats_map_label'](torch.randint(0, 4, (1, 14, 16, 10), generator=g), False).float()
    arrs.update(rrc_vol=vol, rrc_mask=mask.to(torch.uint8))
    seen = set()
    for i, seed in enumerate((3, 4, 11)):
        for iso in (True, False):
            torch.manual_seed(seed)
            v3, m3 = f3['RandomResizedCrop'](vol, mask, (14, 16, 10), (-0.3, 0.3), iso)
            torch.manual_seed(seed)
            vo, mo = O.random_resized_crop(vol, mask, (14, 16, 10), (-0.3, 0.3), iso)
            assert torch.equal(vo, v3) and torch.equal(mo, m3), 'oracle RandomResizedCrop differs from the reference'
            torch.manual_seed(seed)
            sc = float(torch.rand(1) * 0.6 + 0.7)
            seen.add(sc > 1)
            arrs['rrc_v_%d_%d' % (seed, iso)] = v3; arrs['rrc_m_%d_%d' % (seed, iso)] = m3
    assert seen == {True, False}, 'seeds must cover both the padding and the cropping branch'
    arrs['rrc_seeds'] = np.array([3, 4, 11])
    save('augment', **arrs)


def case_augment3d():
    """The 3-D trainer's per-sample transforms (train3d.py:571-578) produced by the REFERENCE's own classes (datasets3d.py:491-597, pure
    numpy, exec-ed from the reference file): RandomRotFlip -> RandomCrop for seeds that cover every rotation count, every flip axis, the
    padding branch and a volume without a modality axis; RandomNoise with the standard-normal field it drew stored beside the result."""
    f3 = _ref_functions('dataloaders/datasets3d.py', ['RandomCrop', 'RandomRotFlip', 'RandomNoise'], extra=dict(pdb=None))
    rs = np.random.RandomState(5)
    img = rs.randn(3, 13, 11, 9).astype(np.float32)
    img[:, :2] = 0                                               # exact zeros: RandomNoise(nonzero_only) leaves them alone
    lab = rs.randint(0, 4, (13, 11, 9)).astype(np.float32)
    arrs = dict(image=img, mask=lab)
    seen_k, seen_ax = set(), set()
    cases = []
    for seed in range(12):
        for tag, out in (('crop', (8, 7, 6)), ('pad', (8, 12, 6))):          # 'pad': W = 11 <= 12 -> the zero-padding branch on every axis
            np.random.seed(seed)
            k, ax = np.random.randint(0, 4), np.random.randint(0, 3)
            seen_k.add(k); seen_ax.add(ax)
            np.random.seed(seed)
            smp = f3['RandomCrop'](out)(f3['RandomRotFlip']()({'image': img, 'mask': lab}))
            assert smp['image'].shape == (3,) + out and smp['mask'].shape == out
            arrs['rfc_img_%d_%s' % (seed, tag)] = np.ascontiguousarray(smp['image']); arrs['rfc_msk_%d_%s' % (seed, tag)] = np.ascontiguousarray(smp['mask'])
            cases.append((seed, tag))
    assert seen_k == {0, 1, 2, 3} and seen_ax == {0, 1, 2}
    arrs['rfc_seeds'] = np.array(sorted({c[0] for c in cases}))
    # single-modality volume ([H, W, D] image): rot90 / flip act on axes (0, 1) / axis directly
    np.random.seed(3)
    smp = f3['RandomCrop']((8, 7, 6))(f3['RandomRotFlip']()({'image': img[0], 'mask': lab}))
    arrs['rfc1_img'] = np.ascontiguousarray(smp['image']); arrs['rfc1_msk'] = np.ascontiguousarray(smp['mask'])
    # RandomNoise: the reference draws np.random.randn(*image.shape) first
    for nz in (1, 0):
        np.random.seed(21)
        z = np.random.randn(*img.shape)
        np.random.seed(21)
        out = f3['RandomNoise'](mu=0.05, sigma=0.1, nonzero_only=bool(nz))({'image': img, 'mask': lab})['image']
        arrs['noise_z'] = z.astype(np.float32); arrs['noise_out_%d' % nz] = out.astype(np.float32)
    save('augment3d', **arrs)


def _tensor_digest(t):
    f = t.detach().double().reshape(-1)
    return np.concatenate([[f.sum().item(), (f * f).sum().item()], sample(t, 8).double().numpy()[:8], np.zeros(max(0, 8 - min(8, f.numel())))])[:10]


def case_init():
    """SegtranInitWeights.init_weights / tie_qk / add_identity_bias (segtran_shared.py:392-402, 522-546, 1241-1264) as the
    model constructors apply them (segtran2d.py:210-213, segtran3d.py:246-249).  The random part is made comparable by re-running the
    three passes on an already built model under a fixed seed: `Module.apply` visits the modules in registration order, which the
    state_dict-order test pins, so the product's mirror must draw the same normal_() streams.  Stored: a digest (sum, sum of squares,
    8 strided samples) of EVERY state_dict entry after the passes."""
    out = {}
    for tag, build in (('cfg2', lambda: R.ref_segtran2d(num_attractors=64)),
                       ('cfg4', lambda: R.ref_segtran3d(num_attractors=64)),
                       ('cfg1_nosq', lambda: R.ref_segtran2d(num_attractors=64, num_translayers=1, compress=(1, 1), use_squeezed_transformer=False))):
        net = build(); load_synth(net)
        torch.manual_seed(4242)
        R.quiet(lambda: (net.apply(net.init_weights), net.apply(net.tie_qk), net.apply(net.add_identity_bias)))
        sd = net.state_dict()
        keys = [k for k in sd if '.pos_coder.all_' not in k and sd[k].is_floating_point()]
        out[tag + '_keys'] = np.array(keys)
        out[tag + '_digest'] = np.stack([_tensor_digest(sd[k]) for k in keys])
        tl = net.voxel_fusion.translayers[0]
        att = tl.ator_out_trans if hasattr(tl, 'ator_out_trans') else tl
        assert att.key.weight is att.query.weight                                   # N2 after tie_qk
    save('init', **out)


def case_keys():
    """state_dict key -> shape lists (checkpoint wire format)."""
    n2 = R.ref_segtran2d()
    n3 = R.ref_segtran3d()
    n3b = R.ref_segtran3d(num_translayers=2, compress=(1, 1, 1))
    n1 = R.ref_segtran2d(num_translayers=1, compress=(1, 1))
    json.dump({'cfg2': {k: list(v.shape) for k, v in n2.state_dict().items()},
               'cfg1': {k: list(v.shape) for k, v in n1.state_dict().items()},
               'cfg4': {k: list(v.shape) for k, v in n3.state_dict().items()},
               'cfg5': {k: list(v.shape) for k, v in n3b.state_dict().items()},
               'cfg2_params': [k for k, _ in n2.named_parameters()],
               'cfg4_params': [k for k, _ in n3.named_parameters()]},
              open(os.path.join(HERE, 'state_dict_keys.json'), 'w'))
    print('  wrote state_dict_keys.json')


CASES = dict(adversarial=case_adversarial, squeeze=case_squeeze, fusion=case_fusion, fusion_nosqueeze=case_fusion_nosqueeze, fusion_mince=case_fusion_mince, posbias=case_posbias, eval=case_eval, polyformer=case_polyformer, unet=case_unet, unet_deconv=case_unet_deconv, discriminator=case_discriminator, effnet=case_effnet, i3d=case_i3d,
             seg2d=case_seg2d, seg2d_polyp=case_seg2d_polyp, seg2d_mince=case_seg2d_mince, seg2d_inbn=case_seg2d_inbn, seg3d=case_seg3d, loss=case_loss, bertadam=case_bertadam, keys=case_keys,
             fullshape=case_fullshape, augment=case_augment, augment3d=case_augment3d, init=case_init)

if __name__ == '__main__':
    todo = [a for a in sys.argv[1:] if a in CASES] or list(CASES)        # further arguments select sub-cases (see case_seg3d)
    for c in todo:
        print('[golden] ' + c)
        CASES[c]()
    print('done')
