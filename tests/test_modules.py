"""CPU: the product modules (segtran_amd.networks.segtran_shared) driven through the fiber-emulated
kernels, checked against the golden fixtures generated from the real reference."""
import pytest
import torch

from segtran_amd import segx, functional as SF
from segtran_amd.networks import segtran_shared as ss
from segtran_amd.synth import synth_state_dict
from util import golden, assert_close


def golden_on(name, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in golden(name).items()}


def mk_config(dims, A, pos_dim=2):
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


def load(mod, prefix):
    shapes = {prefix + k: tuple(v.shape) for k, v in mod.state_dict().items()}
    sd = synth_state_dict(shapes)
    mod.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    mod.to(torch.get_default_device())
    for m in mod.modules():
        if isinstance(m, ss.CrossAttFeatTrans):
            m.tie_qk('shared')


def check_grads(mod, prefix, g, tol=3e-4):
    grads = {prefix + k: p.grad for k, p in mod.named_parameters()}
    gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith('grad:'))
    n = 0
    for k, v in g.items():
        if not k.startswith('grad:'):
            continue
        got = grads[k[5:]]
        assert got is not None, k
        assert_close(got, v, tol, k[5:], scale=gscale)
        n += 1
    assert n >= 10
    return grads


@pytest.mark.parametrize('reassoc', [True, False], ids=['projections-after-contraction', 'reference-op-order'])
@pytest.mark.parametrize('tag,C,Fd', [('c64f64', 64, 64), ('c64f32', 64, 32), ('ffn', 64, 32)])
def test_squeezed_att_feat_trans_vs_reference(backend, monkeypatch, tag, C, Fd, reassoc):
    """reassoc=True (default): the in-squeeze key/value projections are applied after the contraction with the attractor-side
    operands ((q Wk) X^T, (P X) Wv^T); False: projections over all tokens first, as the reference orders them."""
    monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_projections', reassoc)
    monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_min_rows', 0)      # fixtures are tiny; the product only re-associates at >= 4096 rows
    calls = []
    orig = ss.ExpandedFeatTrans.forward
    monkeypatch.setattr(ss.ExpandedFeatTrans, 'forward', lambda self, *a, **kw: (calls.append(kw.get('value_last', False)), orig(self, *a, **kw))[1])
    g = golden_on('squeeze_' + tag, backend.dev)
    cfg = mk_config([C, Fd], 16)
    cfg.has_FFN_in_squeeze = tag == 'ffn'                      # --squeezeuseffn: the in-squeeze layer keeps its one-mode FFN
    mod = ss.SqueezedAttFeatTrans(cfg, 'L').to('cpu')
    prefix = 'voxel_fusion.translayers.0.'
    load(mod, prefix)
    mod.eval()
    X = g['X'].clone().requires_grad_(True)
    Y = mod(X)
    assert calls == [reassoc, False]                           # in-squeeze (one mode, 16 queries, 48 tokens), then squeeze-out
    assert_close(Y, g['Y'], 2e-5, 'Y')
    (Y * g['G']).sum().backward()
    assert_close(X.grad, g['dX'], 1e-4, 'dX')
    grads = check_grads(mod, prefix, g)
    # N3: in-squeeze FFN / output parameters (unless --squeezeuseffn) and the first_norm_layer of every layer WITH an FFN
    # never receive gradients
    for k, v in grads.items():
        if ((tag != 'ffn' and ('in_ator_trans.out_trans.intermediate' in k or 'in_ator_trans.out_trans.output' in k))
                or 'ator_out_trans.out_trans.first_norm_layer' in k or (tag == 'ffn' and 'in_ator_trans.out_trans.first_norm_layer' in k)):
            assert v is None, k
    # in-squeeze soft-aggregate: exact-zero gradients (softmax over a single mode)
    z = grads[prefix + 'in_ator_trans.out_trans.feat_softaggr.feat2score.weight']
    assert z is not None and z.abs().max() == 0


def test_squeeze_out_query_reassociation_equals_reference_op_order(backend, monkeypatch):
    """Squeeze-out layer with many tokens and few attractors (40 tokens, 8 attractors, C = 128, 4 modes: the cost test selects the
    key-side fold  X (Wq_m^T k_m^T) + 1 (k_m bq_m)^T ): output and every gradient equal the reference op order (which the
    fixtures pin) to fp32 rounding."""
    cfg = mk_config([128, 64], 8)
    res = []
    for reassoc in (True, False):
        monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_projections', reassoc)
        monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_min_rows', 0)
        taken = []
        orig = ss.SF.bgemm
        monkeypatch.setattr(ss.SF, 'bgemm', lambda A, B, spec, **kw: (taken.append(spec.bias_b0), orig(A, B, spec, **kw))[1])
        mod = ss.SqueezedAttFeatTrans(cfg, 'L')
        load(mod, 'voxel_fusion.translayers.0.')
        mod.eval()
        X = torch.randn(2, 40, 128, generator=torch.Generator(device='cpu').manual_seed(5), device='cpu').to(backend.dev).requires_grad_(True)
        G = torch.randn(2, 40, 64, generator=torch.Generator(device='cpu').manual_seed(6), device='cpu').to(backend.dev)
        Y = mod(X)
        (Y * G).sum().backward()
        monkeypatch.setattr(ss.SF, 'bgemm', orig)
        assert any(b != 0 for b in taken) == reassoc            # the per-(sample, mode) key-side bias only exists in the folded form
        res.append((Y.detach(), X.grad, {k: p.grad for k, p in mod.named_parameters()}))
    (Y1, dX1, g1), (Y0, dX0, g0) = res
    assert_close(Y1, Y0, 2e-5, 'Y')
    assert_close(dX1, dX0, 1e-4, 'dX')
    gscale = max(v.abs().max().item() for v in g0.values() if v is not None)
    for k, v in g0.items():
        if v is None:
            assert g1[k] is None, k
        else:
            assert_close(g1[k], v, 3e-4, k, scale=gscale)


def test_reassociation_is_gated_by_the_row_threshold(backend, monkeypatch):
    """Below `reassociate_min_rows` token rows (launch-bound configurations such as cfg1) the layer keeps the reference op order."""
    if backend.name != 'emu':
        pytest.skip('host-side dispatch logic: covered once, on the emulator')
    calls = []
    orig = ss.ExpandedFeatTrans.forward
    monkeypatch.setattr(ss.ExpandedFeatTrans, 'forward', lambda self, *a, **kw: (calls.append(kw.get('value_last', False)), orig(self, *a, **kw))[1])
    mod = ss.SqueezedAttFeatTrans(mk_config([64, 32], 16), 'L')
    load(mod, 'voxel_fusion.translayers.0.')
    mod.eval()
    X = torch.randn(2, 48, 64, generator=torch.Generator(device='cpu').manual_seed(2), device='cpu')
    assert ss.CrossAttFeatTrans.reassociate_projections and ss.CrossAttFeatTrans.reassociate_min_rows == 4096
    y_ref_order = mod(X)                                     # 96 token rows < 4096: reference order
    assert calls == [False, False]
    monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_min_rows', 0)
    y_reassoc = mod(X)
    assert calls[2:] == [True, False]
    assert_close(y_reassoc, y_ref_order, 2e-5, 'both orders')


@pytest.mark.parametrize('tag,dims', [('squeeze_c64f64', [64, 64]), ('squeeze_c64f32', [64, 32])])
def test_squeeze_layer_on_the_bf16x6_engine(backend, tag, dims):
    """The squeeze-and-expansion layer with every eligible GEMM (forward and backward, all layouts, split-K, fused epilogues) on the
    bf16x6 tile engine reproduces the reference fixture at the tolerances of the fp32-MFMA engine."""
    L = backend.L
    prev = L.set_engine('x6')
    try:
        L.x6_launches()
        g = golden_on(tag, backend.dev)
        mod = ss.SqueezedAttFeatTrans(mk_config(dims, 16), 'L').to('cpu')
        prefix = 'voxel_fusion.translayers.0.'
        load(mod, prefix)
        mod.eval()
        X = g['X'].clone().requires_grad_(True)
        Y = mod(X)
        assert_close(Y, g['Y'], 2e-5, 'Y')
        (Y * g['G']).sum().backward()
        assert_close(X.grad, g['dX'], 1e-4, 'dX')
        check_grads(mod, prefix, g)
        assert L.x6_launches() >= 6                          # the engine really ran (projections + FFN, forward and backward)
    finally:
        L.set_engine(prev)


def test_fusion_encoder_vs_reference(backend):
    g = golden_on('fusion_small', backend.dev)
    dims = [int(d) for d in g['dims']]
    mod = ss.SegtranFusionEncoder(mk_config(dims, 16), 'Fusion')
    prefix = 'voxel_fusion.'
    load(mod, prefix)
    mod.eval()
    X = g['X'].clone().requires_grad_(True)
    Y = mod(X, g['pos'], g['vmask'], torch.Size((6, 8)))
    assert_close(Y, g['Y'], 2e-5, 'Y')
    (Y * g['G']).sum().backward()
    assert_close(X.grad, g['dX'], 1e-4, 'dX')
    check_grads(mod, prefix, g)


@pytest.mark.parametrize('tag', ['lsinu', 'bias2d', 'bias3d'])
def test_fusion_encoder_nosqueeze_vs_reference(backend, tag):
    """--nosqueeze self-attention, with lsinu codes and with --pos bias sliding positional biases."""
    g = golden_on('fusion_nosqueeze_' + tag, backend.dev)
    dims = [int(d) for d in g['dims']]
    shape = tuple(int(v) for v in g['shape'])
    cfg = mk_config(dims, 16, pos_dim=len(shape))
    cfg.use_squeezed_transformer = False
    cfg.pos_code_type = 'lsinu' if tag == 'lsinu' else 'bias'
    cfg.pos_bias_radius, cfg.pos_code_weight, cfg.max_pos_size = 2, 0.8, (8,) * len(shape)
    mod = ss.SegtranFusionEncoder(cfg, 'Fusion')
    prefix = 'voxel_fusion.'
    load(mod, prefix)
    mod.eval()
    X = g['X'].clone().requires_grad_(True)
    Y = mod(X, g['pos'], g['vmask'], torch.Size(shape))
    assert_close(Y, g['Y'], 2e-5, 'Y')
    (Y * g['G']).sum().backward()
    assert_close(X.grad, g['dX'], 1e-4, 'dX')
    check_grads(mod, prefix, g)


@pytest.mark.parametrize('tag', ['bias2d', 'lsinu2d', 'bias3d'])
def test_fusion_encoder_mince_vs_reference(backend, tag):
    """--mince: multi-scale self-attention (CrossMinceAttFeatTrans) with per-scale sliding biases (2-D / 3-D) and with lsinu
    codes on a 7x10 grid that the scales 3 and 2 do not divide (scale_factor coordinate rule of F.interpolate)."""
    g = golden_on('fusion_mince_' + tag, backend.dev)
    dims = [int(d) for d in g['dims']]
    shape = tuple(int(v) for v in g['shape'])
    cfg = mk_config(dims, 16, pos_dim=len(shape))
    cfg.use_squeezed_transformer, cfg.use_mince_transformer = False, True
    cfg.mince_scales, cfg.mince_channel_props = [int(v) for v in g['scales']], [float(v) for v in g['props']]
    cfg.pos_code_type = 'lsinu' if tag.startswith('lsinu') else 'bias'
    cfg.pos_bias_radius, cfg.pos_code_weight, cfg.max_pos_size = 2, 0.8, (12,) * len(shape)
    mod = ss.SegtranFusionEncoder(cfg, 'Fusion')
    prefix = 'voxel_fusion.'
    load(mod, prefix)                                         # query/key stay untied: not CrossAttFeatTrans instances
    assert all(m.key.weight is not m.query.weight for m in mod.modules() if isinstance(m, ss.CrossMinceAttFeatTrans))
    mod.eval()
    X = g['X'].clone().requires_grad_(True)
    Y = mod(X, g['pos'], g['vmask'], torch.Size(shape))
    assert_close(Y, g['Y'], 2e-5, 'Y')
    (Y * g['G']).sum().backward()
    assert_close(X.grad, g['dX'], 1e-4, 'dX')
    check_grads(mod, prefix, g)


def test_mince_rejected_with_squeeze(backend):
    cfg = mk_config([64, 32], 16)
    cfg.use_mince_transformer, cfg.mince_scales, cfg.mince_channel_props = True, [2, 1], [1, 1]
    with pytest.raises(ValueError):
        ss.SegtranFusionEncoder(cfg, 'Fusion')


def test_pos_bias_rejected_with_squeeze_and_reference_buffers_ignored(backend):
    cfg = mk_config([64, 32], 16)
    cfg.pos_code_type = 'bias'
    with pytest.raises(ValueError):
        ss.SegtranFusionEncoder(cfg, 'Fusion')
    m = ss.SlidingPosBiases2D(2, 2, (8, 8))
    sd = {'biases': torch.ones(5, 5), 'all_h1s': torch.zeros(8, 8, 5, 5, dtype=torch.long), 'all_w2s': torch.zeros(1, dtype=torch.long)}
    m.load_state_dict(sd)                                    # strict: reference index buffers are dropped
    assert m.biases.detach().sum() == 25


def test_training_dropout_runs_and_is_reproducible(backend):
    from segtran_amd import functional as SF
    cfg = mk_config([64, 32], 16)
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.2
    mod = ss.SqueezedAttFeatTrans(cfg, 'L')
    load(mod, 'voxel_fusion.translayers.0.')
    mod.train()
    X = torch.randn(2, 20, 64, generator=torch.Generator(device='cpu').manual_seed(3), device='cpu').to(backend.dev)
    SF.manual_seed(11); y1 = mod(X)
    SF.manual_seed(11); y2 = mod(X)
    SF.manual_seed(12); y3 = mod(X)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    mod.eval()
    assert not torch.allclose(mod(X), y1)
    y1.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in mod.parameters() if p.grad is not None)


@pytest.mark.parametrize('k,s,e,cin,cout,pad,train', [(3, 1, 6, 8, 8, (1, 1), True), (5, 2, 6, 8, 12, (1, 2), True),
                                                      (3, 1, 1, 8, 8, (1, 1), False), (5, 1, 6, 8, 8, (2, 2), False)])
def test_mbconv_block_vs_oracle(backend, k, s, e, cin, cout, pad, train):
    """Product MBConvBlock (libsegx GEMM + depthwise + BN/swish + SE + skip kernels) vs the oracle's restatement."""
    from oracle import segtran_oracle as O
    from segtran_amd.efficientnet.model import MBConvBlock
    blk = MBConvBlock(k, s, e, cin, cout, 0.25, 16)
    blk._depthwise_conv.static_pad = (pad[0], pad[1], pad[0], pad[1])
    prefix = 'backbone._blocks.3.'
    sd = synth_state_dict({prefix + n: tuple(v.shape) for n, v in blk.state_dict().items()})
    blk.load_state_dict({n[len(prefix):]: v for n, v in sd.items()})
    blk.to(backend.dev)
    blk.train(train)
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn(2, cin, 12, 10, generator=g, device='cpu')
    xo = x.clone().requires_grad_(True)
    sdo = {n: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in n else v.clone()) for n, v in sd.items()}
    yo = O.mbconv(sdo, prefix[:-1], xo, dict(k=k, s=s, e=e, cin=cin, cout=cout, pad=pad), train)
    xp = x.to(backend.dev).requires_grad_(True)
    y = blk(xp)
    assert_close(y, yo, 3e-5, 'mbconv y')
    G = torch.randn(*yo.shape, generator=g, device='cpu')
    yo.backward(G); y.backward(G.to(backend.dev))
    assert_close(xp.grad, xo.grad, 3e-4, 'mbconv dx')
    gs = max(v.grad.abs().max().item() for v in sdo.values() if v.is_floating_point() and v.grad is not None)
    for n, p in blk.named_parameters():
        assert_close(p.grad, sdo[prefix + n].grad, 3e-4, n, scale=gs)


def test_i3d_unit_vs_oracle(backend):
    from oracle import segtran_oracle as O
    from segtran_amd.networks.aj_i3d.aj_i3d import Unit3D
    for kshape, stride in (((1, 1, 1), (1, 1, 1)), ((3, 3, 3), (1, 1, 1))):
        u = Unit3D(6, 10, kshape, stride)
        prefix = 'backbone.Mixed_3b.b1b.'
        sd = synth_state_dict({prefix + n: tuple(v.shape) for n, v in u.state_dict().items()})
        u.load_state_dict({n[len(prefix):]: v for n, v in sd.items()})
        u.to(backend.dev).train()
        g = torch.Generator(device='cpu').manual_seed(6)
        x = torch.randn(2, 6, 4, 6, 5, generator=g, device='cpu')
        xo = x.clone().requires_grad_(True)
        yo = O.unit3d(sd, prefix[:-1], xo, kshape, stride, training=True)
        xp = x.to(backend.dev).requires_grad_(True)
        y = u(xp)
        assert_close(y, yo, 3e-5, 'unit3d y')
        G = torch.randn(*yo.shape, generator=g, device='cpu')
        yo.backward(G); y.backward(G.to(backend.dev))
        assert_close(xp.grad, xo.grad, 3e-4, 'unit3d dx')


@pytest.mark.parametrize('tag', ['sq', 'rect'])
def test_polyformer_layer_vs_reference(backend, tag):
    """PolyformerLayer (no-FFN multi-mode attention pair on a pooled map + residual) incl. the reference's
    w-major token order that is mapped back as (row, col) (N10)."""
    from segtran_amd.networks.polyformer import Polyformer
    g = golden_on('polyformer_' + tag, backend.dev)
    B, C, H, W = g['X'].shape
    mod = Polyformer(C)
    layer = mod.polyformer_layers[0]
    layer.attractors.data = layer.attractors.data[:, :16]
    prefix = 'polyformer.polyformer_layers.0.'
    shapes = {prefix + k: tuple(v.shape) for k, v in layer.state_dict().items()}
    layer.load_state_dict({k[len(prefix):]: v for k, v in synth_state_dict(shapes).items()})
    mod.to(torch.get_default_device()); mod.eval()
    X = g['X'].clone().requires_grad_(True)
    Y = mod(X)
    assert_close(Y, g['Y'], 2e-5, 'Y')
    (Y * g['G']).sum().backward()
    assert_close(X.grad, g['dX'], 1e-4, 'dX')
    grads = {prefix + k: p.grad for k, p in layer.named_parameters()}
    gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith('grad:'))
    for k, v in g.items():
        if k.startswith('grad:'):
            assert grads[k[5:]] is not None, k
            assert_close(grads[k[5:]], v, 3e-4, k[5:], scale=gscale)
    for k in g['unused']:
        assert grads[str(k)] is None, k


def test_avgpool2_and_transpose(backend):
    from segtran_amd import functional as SF
    import torch.nn.functional as F
    x = torch.randn(2, 3, 7, 10, generator=torch.Generator(device='cpu').manual_seed(8), device='cpu').to(backend.dev).requires_grad_(True)
    y = SF.avg_pool2(x)
    xr = x.detach().cpu().clone().requires_grad_(True)
    yr = F.avg_pool2d(xr, 2)
    assert_close(y, yr.detach(), 1e-6, 'avgpool')
    G = torch.randn(*yr.shape, generator=torch.Generator(device='cpu').manual_seed(9), device='cpu')
    y.backward(G.to(backend.dev)); yr.backward(G)
    assert_close(x.grad, xr.grad, 1e-6, 'avgpool bwd')
    t = torch.randn(3, 37, 70, generator=torch.Generator(device='cpu').manual_seed(10), device='cpu').to(backend.dev).requires_grad_(True)
    tt = SF.transpose12(t)
    assert torch.equal(tt.cpu(), t.detach().cpu().transpose(1, 2))
    tt.backward(tt.detach())
    assert torch.equal(t.grad.cpu(), t.detach().cpu())


@pytest.mark.parametrize('k,s,e,cin,cout', [(3, 1, 6, 8, 8), (5, 2, 6, 8, 12), (3, 1, 1, 8, 8)])
def test_mbconv_block_node_equals_per_op_nodes(backend, monkeypatch, k, s, e, cin, cout):
    """SF.block_node (one autograd node per MBConv block, its ops recorded on a tape: efficientnet/model.py:82-126) runs the same kernels in the same order as the
    per-op autograd nodes of rounds 1-5: output, input gradient and every parameter gradient bit for bit, drop_connect and the skip connection included."""
    from segtran_amd.efficientnet.model import MBConvBlock
    blk = MBConvBlock(k, s, e, cin, cout, 0.25, 16)
    prefix = 'backbone._blocks.3.'
    sd = synth_state_dict({prefix + n: tuple(v.shape) for n, v in blk.state_dict().items()})
    blk.load_state_dict({n[len(prefix):]: v for n, v in sd.items()})
    blk.to(backend.dev).train()
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn(3, cin, 12, 10, generator=g, device='cpu').to(backend.dev)
    G = torch.randn(3, cout, 12 // s, 10 // s, generator=g, device='cpu').to(backend.dev)
    res = []
    for on in (True, False):
        monkeypatch.setattr(SF, 'block_nodes', on)
        blk.load_state_dict({n[len(prefix):]: v for n, v in sd.items()})              # running statistics back to the start
        blk.zero_grad(set_to_none=True)
        SF.manual_seed(21)
        xp = x.clone().requires_grad_(True)
        xin = xp * 1.0                                                                  # a non-leaf input, as inside the backbone
        y = blk(xin, drop_connect_rate=0.3)
        assert (type(y.grad_fn).__name__ == '_BlockBackward') == on
        y.backward(G)
        res.append([y.detach().clone(), xp.grad.clone()] + [p.grad.clone() for p in blk.parameters()] + [b.clone() for b in blk.buffers()])
    assert len(res[0]) == len(res[1])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_block_node_fails_loudly_on_untracked_use(backend, monkeypatch):
    """A block input that requires a gradient and is used OUTSIDE the ops of functional.py inside a block node gets no gradient: that must raise, not train on zeros."""
    monkeypatch.setattr(SF, 'block_nodes', True)                                        # opt-in (off by default: measured slower on the host)
    w = torch.randn(4, 4, device=backend.dev, requires_grad=True)
    x = torch.randn(2, 4, device=backend.dev, requires_grad=True)

    def fn(x_):
        return SF.linear(x_ + w.sum(), torch.eye(4, device=backend.dev))              # `+` is not a libsegx op: neither x_ nor w is seen by the tape

    y = SF.block_node(fn, x, (), [w])
    with pytest.raises(RuntimeError, match='received none'):
        y.sum().backward()
    with torch.no_grad():
        assert SF.block_node(fn, x, (), [w]).grad_fn is None                            # gradients off: plain ops, no node
    monkeypatch.setattr(SF, 'block_nodes', False)
    y = SF.block_node(fn, x, (), [w])                                                   # switched off: per-op autograd nodes, ATen work is differentiated as usual
    assert type(y.grad_fn).__name__ != '_BlockBackward'
    y.sum().backward()
    assert w.grad is not None and x.grad is not None
