"""GPU (-m gpu): whole-model parity of the HIP path against the golden fixtures generated from the real reference,
full-size checks at the BASELINE.json shapes, and train-step sanity.  Tolerances: logits within 1e-3 absolute
(north_star) AND 1e-4 of the tensor's scale; hardened label maps bit-exact wherever |logit| > 1e-5."""
import numpy as np
import pytest
import torch

from segtran_amd import engine, functional as SF
from segtran_amd.efficientnet.model import MBConvBlock
from segtran_amd.networks.aj_i3d.aj_i3d import InceptionModule
from segtran_amd.synth import sample, synth_brats, synth_image2d
from util import golden, golden_json, assert_close

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
# fp64 referee: where the fp32 reference itself is further than the tolerance from the fp64 truth, the HIP result may be at most REFEREE x as far from
# the truth as the fp32 reference is (r04: 3; r05: 2.25 -- session r05_g logged all 62 gradients that needed the referee on both engines and op orders,
# profiles/r05_g_referee.txt: the largest ratio is 2.06, the stem BatchNorm weight of the small cfg4 fixture on the fp32 engine; 1.5 fails 14 of them)
import os as _os
REFEREE = float(_os.environ.get('SEGX_REFEREE_FACTOR', '2.25'))


def _referee_log(name, e32, e64, r64):
    """SEGX_REFEREE_LOG=<file>: every gradient that needed the fp64 referee (|hip - ref32| above the tolerance), with its three distances"""
    path = _os.environ.get('SEGX_REFEREE_LOG')
    if path and e32 > 1e-3:
        with open(path, 'a') as f:
            f.write('%s %s e32=%.3e e64=%.3e r64=%.3e ratio=%.2f\n' % (_os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0], name, e32, e64, r64, e64 / max(r64, 1e-30)))


@pytest.fixture(params=['x6', 'f32'], autouse=True)
def tile_engine(request):
    """Every whole-model parity test runs on BOTH tile engines: bf16x6 (the product default) and the fp32 MFMA (segx_tune knob 4)."""
    from segtran_amd import segx
    L = segx.lib()
    prev = L.set_engine(request.param)
    yield request.param
    L.set_engine(prev)


def _grads_vs_golden(net, g, tol=1e-3, referee=False):
    named = dict(net.named_parameters())
    gscale = float(g['gscale'])
    n = 0
    for k, v in g.items():
        if not k.startswith('grad:'):
            continue
        got = named[k[5:]].grad
        assert got is not None, k
        got = got if got.numel() == v.numel() else sample(got)
        if referee and 'grad64:' + k[5:] in g:
            got, v, v64 = got.detach().cpu().reshape(-1), v.reshape(-1), g['grad64:' + k[5:]].reshape(-1)
            e32, e64, r64 = [(a - b).abs().max().item() / gscale for a, b in ((got, v), (got, v64), (v, v64))]
            _referee_log(k[5:], e32, e64, r64)
            assert e32 <= tol or e64 <= REFEREE * r64, '%s: |hip - ref32| %.2e, |hip - fp64| %.2e, |ref32 - fp64| %.2e of the gradient scale' % (k[5:], e32, e64, r64)
        else:
            assert_close(got.reshape(-1), v.reshape(-1), tol, k[5:], scale=gscale)
        n += 1
    assert n >= 10
    for k in g['unused']:                                  # N3
        assert named[str(k)].grad is None, k


@pytest.mark.parametrize('fused_tail', [True, False], ids=['reassociated', 'reference-op-order'])
@pytest.mark.parametrize('tag,cfg,train', [('seg2d_cfg2_eval', 'cfg2', False), ('seg2d_cfg1_eval', 'cfg1', False),
                                           ('seg2d_cfg2_train', 'cfg2', True)])
def test_segtran2d_vs_reference(tag, cfg, train, fused_tail, monkeypatch):
    from segtran_amd.networks import segtran_shared as ss
    monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_projections', fused_tail)      # both re-associations on, or neither
    monkeypatch.setattr(MBConvBlock, 'gate_in_weights', fused_tail)                        # ... and the squeeze-excite gate in the projection weights
    monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_min_rows', 0)                  # the fixtures are small; the product's size threshold is 4096 rows
    g = golden(tag)
    c = dict(engine.CONFIGS[cfg], size=(64, 64))
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=int(g['A']))
    net.fuse_output_tail = fused_tail                      # default True: class projection composed into the bridge weights
    net.backbone.drop_connect_rate = 0.0                   # fixture: drop_connect off, dropout 0 (H3)
    net.train() if train else net.eval()
    x = g['x'].to(DEV)
    y = net(x)
    assert_close(y, g['logits'], 1e-4, 'logits')
    assert (y.cpu() - g['logits']).abs().max().item() < 1e-3
    safe = g['logits'].abs() > 1e-5
    assert torch.equal((y.cpu() > 0)[safe], g['labels'][safe]), 'hardened label map differs'
    pw, cw = engine.loss_weights('fundus', DEV)
    loss, _ = SF.seg_loss(y, engine.map_mask('fundus', g['mask'].to(DEV)), pw, cw)
    assert abs(loss.item() - float(g['loss'])) < 2e-5
    loss.backward()
    _grads_vs_golden(net, g)


def test_segtran2d_nosqueeze_pos_bias_vs_reference():
    """--nosqueeze --pos bias --posr 3, train mode, logits + loss + gradients (incl. the bias table)."""
    g = golden('seg2d_cfg1_nosq_bias_train')
    c = dict(engine.CONFIGS['cfg1'], size=(64, 64))
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=int(g['A']), use_squeezed_transformer=False,
                             pos_code_type='bias', pos_bias_radius=3)
    net.backbone.drop_connect_rate = 0.0
    net.train()
    y = net(g['x'].to(DEV))
    assert_close(y, g['logits'], 1e-4, 'logits')
    assert (y.cpu() - g['logits']).abs().max().item() < 1e-3
    safe = g['logits'].abs() > 1e-5
    assert torch.equal((y.cpu() > 0)[safe], g['labels'][safe])
    pw, cw = engine.loss_weights('fundus', DEV)
    loss, _ = SF.seg_loss(y, engine.map_mask('fundus', g['mask'].to(DEV)), pw, cw)
    assert abs(loss.item() - float(g['loss'])) < 2e-5
    loss.backward()
    assert 'grad:voxel_fusion.pos_code_layer.pos_coder.biases' in g
    _grads_vs_golden(net, g)


def test_segtran2d_inbn_and_outdrop_vs_reference():
    """--inbn (BatchNorm2d in the in-FPN, train-mode batch statistics) against the reference fixture; --outdrop: identity in
    eval mode, and in train mode a fresh mask per call that zeroes ~p of the out-FPN features."""
    g = golden('seg2d_cfg1_inbn_train')
    c = dict(engine.CONFIGS['cfg1'], size=(64, 64))
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=int(g['A']), in_fpn_use_bn=True)
    net.backbone.drop_connect_rate = 0.0
    net.train()
    y = net(g['x'].to(DEV))
    assert_close(y, g['logits'], 1e-4, 'logits')
    safe = g['logits'].abs() > 1e-5
    assert torch.equal((y.cpu() > 0)[safe], g['labels'][safe])
    pw, cw = engine.loss_weights('fundus', DEV)
    loss, _ = SF.seg_loss(y, engine.map_mask('fundus', g['mask'].to(DEV)), pw, cw)
    assert abs(loss.item() - float(g['loss'])) < 2e-5
    loss.backward()
    assert 'grad:in_bn4b.weight' in g
    _grads_vs_golden(net, g)
    assert int(net.in_bn4b.num_batches_tracked) == 1 and int(net.in_bn3b.num_batches_tracked) == 0

    plain = engine.build_model(c, DEV, dropout_prob=0.3, attractors=32)
    drop = engine.build_model(c, DEV, dropout_prob=0.3, attractors=32, out_fpn_do_dropout=True)
    drop.load_state_dict(plain.state_dict())
    plain.fuse_output_tail = False                             # same op order as the --outdrop model (which never composes the head)
    plain.eval(); drop.eval()
    x = g['x'].to(DEV)
    assert torch.equal(plain(x), drop(x))                      # nn.Dropout is the identity in eval mode
    feats = {}
    drop.out_conv.register_forward_hook(lambda m, i, o: feats.__setitem__('f', i[0].detach()))
    drop.train()
    SF.manual_seed(5); y1 = drop(x); f1 = feats['f']
    y2 = drop(x); f2 = feats['f']
    z1, z2 = (f1 == 0).float().mean().item(), (f2 == 0).float().mean().item()
    assert abs(z1 - 0.3) < 0.02 and abs(z2 - 0.3) < 0.02 and not torch.equal(f1 == 0, f2 == 0)


def test_segtran2d_mince_vs_reference():
    """--nosqueeze --mince --mincescales 4,2,1 --minceprops 1,1,2 --pos bias --posr 2, train mode: 12 x 12 tokens attended on
    3 x 3 / 6 x 6 / 12 x 12 grids, untied query/key, one bias table per scale; logits + loss + gradients."""
    g = golden('seg2d_cfg1_mince_train')
    c = dict(engine.CONFIGS['cfg1'], size=(96, 96))
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=int(g['A']), use_squeezed_transformer=False,
                             use_mince_transformer=True, mince_scales=[4, 2, 1], mince_channel_props=[1, 1, 2],
                             pos_code_type='bias', pos_bias_radius=2)
    net.backbone.drop_connect_rate = 0.0
    net.train()
    y = net(g['x'].to(DEV))
    assert_close(y, g['logits'], 1e-4, 'logits')
    assert (y.cpu() - g['logits']).abs().max().item() < 1e-3
    safe = g['logits'].abs() > 1e-5
    assert torch.equal((y.cpu() > 0)[safe], g['labels'][safe])
    pw, cw = engine.loss_weights('fundus', DEV)
    loss, _ = SF.seg_loss(y, engine.map_mask('fundus', g['mask'].to(DEV)), pw, cw)
    assert abs(loss.item() - float(g['loss'])) < 2e-5
    loss.backward()
    assert 'grad:voxel_fusion.pos_code_layers.2.pos_coder.biases' in g and 'grad:voxel_fusion.translayers.0.key.weight' in g
    _grads_vs_golden(net, g)


@pytest.mark.parametrize('fused_tail', [True, False], ids=['reassociated', 'reference-op-order'])
@pytest.mark.parametrize('tag,train', [('seg3d_cfg4_eval', False), ('seg3d_cfg4_train', True)])
def test_segtran3d_vs_reference(tag, train, fused_tail, monkeypatch):
    from segtran_amd.networks import segtran_shared as ss
    monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_projections', fused_tail)
    monkeypatch.setattr(ss.CrossAttFeatTrans, 'reassociate_min_rows', 0)
    g = golden(tag)
    c = dict(engine.CONFIGS['cfg4'], size=(112, 112, 16))
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=int(g['A']))
    net.fuse_output_tail = net.fuse_input_bridge = fused_tail      # reference op order: neither the output head nor the input bridge is composed
    monkeypatch.setattr(InceptionModule, 'fuse_reductions', fused_tail)                # ... and four independent Inception branches
    net.train() if train else net.eval()
    x, lab = synth_brats(1, 112, 112, 16, 1337)
    assert torch.equal(sample(x), g['x_sample'])
    y = net(x.to(DEV))
    assert_close(sample(y.cpu(), 65536), g['logits'], 1e-4, 'logits')
    ref_bits = np.unpackbits(g['labels'].numpy())[:y.numel()].astype(bool)
    got_bits = (y.detach().cpu() > 0).numpy().reshape(-1)
    # hardened labels must be bit-exact wherever the sign is decided beyond the fp32 summation-order noise: 1e-5 in eval mode;
    # in train mode this fixture normalises with batch-1 BatchNorm statistics, which amplifies rounding (measured logit
    # deviation 4.2e-5 on a scale of 3.4, split-K forward convolutions vs the CPU reference) -> 1e-4
    margin_ok = (y.detach().cpu().abs() > (1e-4 if train else 1e-5)).numpy().reshape(-1)
    assert np.array_equal(got_bits[margin_ok], ref_bits[margin_ok])
    pw, cw = engine.loss_weights('brats', DEV)
    loss, _ = SF.seg_loss(y, engine.map_mask('brats', lab.to(DEV)), pw, cw)
    assert abs(loss.item() - float(g['loss'])) < 2e-5
    loss.backward()
    # Referee: the fixture also holds the SAME reference modules run in fp64.  In train mode this fixture
    # normalises with batch-1 BatchNorm statistics over as few as 49 samples per channel (Mixed_5*), which amplifies fp32
    # summation-order differences: the fp32 CPU reference itself is 1.2e-2 of the gradient scale away from the fp64 result
    # (3.1e-5 in eval mode).  A gradient passes when it is within 1e-3 of the fp32 reference OR no further from the fp64 result
    # than 3x the fp32 reference is -- the noisy side is bounded by the referee, not by a blanket tolerance.
    _grads_vs_golden(net, g, referee=True)


def test_train_step_with_dropout_decreases_loss_and_is_seed_reproducible():
    c = dict(engine.CONFIGS['cfg2'], size=(64, 64))

    def run(seed):
        torch.manual_seed(seed); SF.manual_seed(seed)
        net = engine.build_model(c, DEV, dropout_prob=0.2, attractors=32)
        net.train()
        step = engine.TrainStep(net, engine.init_optimizer(net, 'fundus', t_total=100, warmup_steps=2), 'fundus')
        x, raw = engine.synth_batch(c, 2, DEV)
        return [float(step(x, raw)) for _ in range(6)]

    a, b = run(3), run(3)
    # libsegx kernels are deterministic (no float atomics, counter-based dropout) and no ATen / MIOpen arithmetic is left on the step;
    # the tolerance only covers the drop_connect draw (torch's generator) being consumed identically -- measured: bit-identical runs
    assert max(abs(u - v) for u, v in zip(a, b)) < 2e-3, (a, b)
    assert all(v == v for v in a) and a[-1] < a[0]


def test_unused_and_zero_grad_parameters_follow_reference_semantics():
    """N3: parameters without gradients are skipped by BertAdam (no decay); in-squeeze feat2score gets exact zeros
    and IS decayed."""
    c = dict(engine.CONFIGS['cfg1'], size=(64, 64))
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=32)
    net.train()
    opt = engine.init_optimizer(net, 'fundus', t_total=100, warmup_steps=1, lr=0.05, decay=0.1)
    step = engine.TrainStep(net, opt, 'fundus')
    named = dict(net.named_parameters())
    unused = named['voxel_fusion.translayers.0.in_ator_trans.out_trans.output.group_linear.weight']
    zero_g = named['voxel_fusion.translayers.0.in_ator_trans.out_trans.feat_softaggr.feat2score.weight']
    u0, z0 = unused.detach().clone(), zero_g.detach().clone()
    x, raw = engine.synth_batch(c, 2, DEV)
    step(x, raw); step(x, raw)
    assert torch.equal(unused.detach(), u0)
    assert zero_g.grad.abs().max() == 0 and not torch.equal(zero_g.detach(), z0)


def test_gemm_fullsize_properties():
    """cfg-2 layer-0 shapes: result vs rocBLAS fp32 and linearity in A."""
    from segtran_amd import segx
    L = segx.lib()
    g = torch.Generator(device='cpu').manual_seed(0)
    M, N, K = 6 * 4096, 1792, 1792
    A = torch.randn(M, K, generator=g).to(DEV); A2 = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.02).to(DEV)
    C1 = torch.empty(M, N, device=DEV); C2 = torch.empty_like(C1); C12 = torch.empty_like(C1)
    st = ((0, 0, K, 1), (0, 0, K, 1), (0, 0, N))
    L.gemm(A, W, C1, M, N, K, *st); L.gemm(A2, W, C2, M, N, K, *st); L.gemm(A + A2, W, C12, M, N, K, *st)
    ref = A @ W.t()
    assert_close(C1, ref, 2e-5, 'vs rocBLAS')
    assert_close(C12, C1 + C2, 2e-5, 'linearity')


def test_eval_path_2d_vs_reference():
    """product sliding-window inference (HIP net + infer.hip) vs the reference's test_single_batch fixture."""
    from segtran_amd import test_util2d as T2
    g = golden('eval2d')
    net = engine.build_model(dict(engine.CONFIGS['cfg1'], size=(64, 64)), DEV, dropout_prob=0.0, attractors=int(g['A']))
    net.eval()
    for tag in 'ab':
        cfg = [int(v) for v in g['cfg_' + tag]]
        hard, soft = T2.test_single_batch(net, g['x_' + tag].to(DEV), tuple(cfg[0:2]), tuple(cfg[2:4]), tuple(cfg[4:6]), 'fundus', 3)
        assert hard.dtype == torch.int32 and hard.shape == g['hard_' + tag].shape
        assert_close(soft.cpu(), g['soft_' + tag], 1e-5, 'soft ' + tag)
        safe = (g['soft_' + tag] - 0.5).abs() > 1e-5
        assert torch.equal(hard.cpu()[safe], g['hard_' + tag].int()[safe]), 'hardened label map differs'
    m = T2.calc_batch_metric([g['hard_a'][i].float().to(DEV) for i in range(2)], [g['gt'][i].float().to(DEV) for i in range(2)], 3)
    assert_close(torch.from_numpy(m).float(), g['dice'][:, 1:], 1e-6, 'dice')


def test_eval_path_3d_vs_reference():
    from segtran_amd import test_util3d as T3
    from segtran_amd.dataloaders import datasets3d as D3
    g = golden('eval3d')
    net = engine.build_model(dict(engine.CONFIGS['cfg4'], size=(112, 112, 16)), DEV, dropout_prob=0.0, attractors=int(g['A']))
    net.eval()
    vol = synth_brats(1, 112, 168, 16, int(g['seed']))[0][0].to(DEV)
    hard, soft = T3.test_single_case(net, vol, (112, 112, 16), (112, 112, 16), 2, 56, 16, 'brats')
    assert_close(sample(soft.cpu(), 65536), g['soft'], 1e-5, 'soft')
    ref_bits = np.unpackbits(g['hard'].numpy())[:hard.numel()].astype(bool).reshape(hard.shape)
    safe = ((soft.cpu() - 0.5).abs() > 1e-5).numpy()
    assert np.array_equal((hard.cpu().numpy() > 0)[safe], ref_bits[safe]), 'hardened label map differs'
    metric, valid = T3.calculate_metric_percase(hard, hard, 4)
    assert np.allclose(metric[valid[:, 0] > 0, 0][hard[1:].reshape(3, -1).sum(1).cpu().numpy() > 0], 1.0)
    p = g['probs'].to(DEV)
    for k in (True, False):
        assert torch.equal(D3.make_brats_pred_consistent(p, k).cpu(), g['cons_true' if k else 'cons_false'])
    assert torch.equal(D3.brats_inv_map_label(g['inv_in'].to(DEV)).cpu(), g['inv'])
    assert torch.equal(D3.harden_segmap3d(p).cpu(), g['harden3d'].int())


def test_segtran2d_polyp_cfg3_vs_reference():
    """cfg3 flags: polyp task (2 classes, label map through segx_label_nhot), 3 layers with compression, 88 x 88 -> 11 x 11 tokens."""
    g = golden('seg2d_cfg3_polyp_train')
    net = engine.build_model(dict(engine.CONFIGS['cfg3'], size=(88, 88)), DEV, dropout_prob=0.0, attractors=int(g['A']))
    net.backbone.drop_connect_rate = 0.0
    net.train()
    y = net(g['x'].to(DEV))
    assert_close(y, g['logits'], 1e-4, 'logits')
    safe = g['logits'].abs() > 1e-4                       # batch-1 train-mode BatchNorm (see the 3-D train fixture)
    assert torch.equal((y.cpu() > 0)[safe], g['labels'][safe]), 'hardened label map differs'
    nhot = engine.map_mask('polyp', g['mask'].to(DEV))
    assert torch.equal(nhot.cpu(), g['nhot'].float())
    pw, cw = engine.loss_weights('polyp', DEV)
    loss, _ = SF.seg_loss(y, nhot, pw, cw)
    assert abs(loss.item() - float(g['loss'])) < 5e-5
    loss.backward()
    _grads_vs_golden(net, g, tol=2e-3)


def test_segtran3d_cfg5_two_layers_vs_reference():
    g = golden('seg3d_cfg5_eval')
    c = dict(engine.CONFIGS['cfg5'], size=(112, 112, 16))
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=int(g['A']))
    net.eval()
    x, lab = synth_brats(1, 112, 112, 16, 1337)
    assert torch.equal(sample(x), g['x_sample'])
    y = net(x.to(DEV))
    assert_close(sample(y.cpu(), 65536), g['logits'], 1e-4, 'logits')
    ref_bits = np.unpackbits(g['labels'].numpy())[:y.numel()].astype(bool)
    got = (y.detach().cpu() > 0).numpy().reshape(-1)
    ok = (y.detach().cpu().abs() > 1e-5).numpy().reshape(-1)
    assert np.array_equal(got[ok], ref_bits[ok])
    pw, cw = engine.loss_weights('brats', DEV)
    loss, _ = SF.seg_loss(y, engine.map_mask('brats', lab.to(DEV)), pw, cw)
    assert abs(loss.item() - float(g['loss'])) < 2e-5
    loss.backward()
    _grads_vs_golden(net, g)


@pytest.mark.parametrize('shape', [(1, 3, 72, 104), (2, 3, 40, 56), (1, 3, 576, 576)])       # last: 5184 tokens (> the 4096 a register row holds)
def test_ragged_2d_sizes_vs_oracle(shape):
    """Non-square inputs whose token grid (H/8 x W/8 = 9 x 13, 5 x 7) is odd in both directions: forward against the CPU oracle,
    backward finite.  (The reference accepts any H, W divisible by 8.)"""
    from oracle import segtran_oracle as O
    from segtran_amd.synth import synth_state_dict
    c = dict(engine.CONFIGS['cfg2'], size=shape[2:])
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=32)
    net.eval()
    x = torch.randn(*shape, generator=torch.Generator(device='cpu').manual_seed(5), device='cpu')
    y = net(x.to(DEV))
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    with torch.no_grad(), torch.device('cpu'):
        yo = O.segtran2d_forward(sd, x, [1792, 1792, 896, 448])
    assert_close(y.cpu(), yo, 1e-4, 'logits')
    safe = yo.abs() > 1e-5
    assert torch.equal((y.cpu() > 0)[safe], (yo > 0)[safe])
    y.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


def test_ragged_3d_size_vs_oracle():
    """112 x 120 x 24 volume (token grid 14 x 15 x 3 -> 630 tokens, not a multiple of 4) against the CPU oracle."""
    from oracle import segtran_oracle as O
    from segtran_amd.synth import synth_state_dict
    c = dict(engine.CONFIGS['cfg4'], size=(112, 120, 24))
    net = engine.build_model(c, DEV, dropout_prob=0.0, attractors=32)
    net.eval()
    x, _ = synth_brats(1, 112, 120, 24, 77)
    y = net(x.to(DEV))
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    with torch.no_grad(), torch.device('cpu'):
        yo = O.segtran3d_forward(sd, x, [1024, 1024])
    assert_close(y.cpu(), yo, 1e-4, 'logits')
    safe = yo.abs() > 1e-5
    assert torch.equal((y.cpu() > 0)[safe], (yo > 0)[safe])
    y.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


def test_graphed_train_step_matches_eager_and_draws_fresh_dropout_masks(tile_engine):
    """engine.GraphedTrainStep: the whole train step captured into one hipGraph.  Dropout-free it must reproduce the eager loss trajectory
    (same kernels, same order; the LR schedule comes from the refreshed device table); with dropout every replay must use a NEW part of the
    Philox stream (device-side base advanced by the captured segx_rng_advance) -- and the eager path must be unaffected afterwards."""
    from segtran_amd import segx
    if tile_engine != 'x6':
        pytest.skip('one engine is enough for the capture mechanics')
    c = dict(engine.CONFIGS['cfg1'], size=(64, 64))

    def make(drop):
        torch.manual_seed(7); SF.manual_seed(7)
        net = engine.build_model(c, DEV, dropout_prob=drop, attractors=32)
        net.backbone.drop_connect_rate = 0.0 if drop == 0 else net.backbone.drop_connect_rate
        net.train()
        return engine.TrainStep(net, engine.init_optimizer(net, 'fundus', t_total=50, warmup_steps=4), 'fundus')
    x, raw = engine.synth_batch(c, 2, DEV)
    eager = make(0.0)
    ref = [float(eager(x, raw).detach()) for _ in range(8)]
    g = engine.GraphedTrainStep(make(0.0), x, raw, warmup=3)
    try:
        got = [float(g(x, raw).detach()) for _ in range(5)]
        assert max(abs(a - b) for a, b in zip(got, ref[3:])) < 2e-5, (got, ref[3:])
        assert g.step.opt.step_count == 8 and abs(g.step.opt.get_lr()[0] - eager.opt.get_lr()[0]) < 1e-12
    finally:
        g.close()
    gd = engine.GraphedTrainStep(make(0.2), x, raw, warmup=2)
    try:
        assert gd.span > 0 and gd.span % 4 == 0
        for _ in range(3):
            gd(x, raw)
        torch.cuda.synchronize()
        assert int(gd.rng_base.item()) == 3 * gd.span
    finally:
        gd.close()
    # the device-side base is an additive shift of the stream: dropout at (offset 0, base b) == dropout at (offset b, no base)
    L = segx.lib()
    v = torch.ones(4096, device=DEV); y0 = torch.empty_like(v); y1 = torch.empty_like(v)
    base = torch.full((1,), 1024, dtype=torch.int64, device=DEV)
    L.set_rng_base(base); L.dropout(v, y0, 4096, 0.3, 99, 0); L.set_rng_base(None)
    L.dropout(v, y1, 4096, 0.3, 99, 1024)
    assert torch.equal(y0, y1) and 0.2 < (y0 == 0).float().mean().item() < 0.4


# ---- the HIP backbones stand-alone against the reference's own endpoint tensors -------------------------------
def _load_prefixed(net, prefix):
    from segtran_amd.synth import synth_state_dict
    sd = synth_state_dict({prefix + k: tuple(v.shape) for k, v in net.state_dict().items()})      # the fixture generators hash the PREFIXED names
    net.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return net


def test_hip_efficientnet_b4_endpoints_vs_reference_fixture():
    """efficientnet/model.py:240-283 (extract_endpoints) on the HIP kernels, eval mode, the two fixture sizes (64 x 64 and the non-square 96 x 64:
    quirk N6's nominal-geometry pads (0,1) / (2,2) / (0,1) / (1,2) at the four stride-2 depthwise convolutions), all five endpoints against
    the tensors the real reference produced (tests/golden/effnet_b4.npz), not the oracle."""
    from segtran_amd.efficientnet.model import EfficientNet
    g = golden('effnet_b4')
    net = _load_prefixed(EfficientNet.from_name('efficientnet-b4', stem_stride=1), 'backbone.').to(DEV).eval()
    assert net.endpoint_blk_indices == [int(v) for v in g['endpoint_blk']]
    for tag in 'ab':
        with torch.no_grad():
            ep = net.extract_endpoints(g['x_' + tag].to(DEV))
        for i in range(5):
            f = ep['reduction_%d' % (i + 1)].cpu()
            assert list(f.shape) == g['%s_shape%d' % (tag, i)].tolist()
            want = g['%s_ep%d' % (tag, i)]
            got = f if want.numel() == f.numel() else sample(f, 16384)
            assert_close(got.reshape(-1), want.reshape(-1), 1e-4, 'effnet %s ep%d' % (tag, i))


def test_hip_i3d_features_vs_reference_fixture():
    """aj_i3d.py:325-333 (extract_features) on the HIP kernels at the minimum legal size 16 x 112 x 112 (N7: dynamic 'same' pads, zero-padded
    max-pools), five endpoints against the real reference's tensors (tests/golden/i3d.npz)."""
    from segtran_amd.networks.aj_i3d.aj_i3d import InceptionI3d
    g = golden('i3d')
    x = synth_image2d(1, 16 * 112, int(g['x_seed']), 112).view(1, 3, 16, 112, 112)
    assert torch.equal(sample(x), g['x_sample'])
    net = _load_prefixed(InceptionI3d(do_pool1=False), 'backbone.').to(DEV).eval()
    with torch.no_grad():
        fd = net.extract_features(x.to(DEV))
    for i, n in enumerate(['MaxPool3d_2a_3x3', 'Conv3d_2c_3x3', 'Mixed_3c', 'Mixed_4f', 'Mixed_5c']):
        f = fd[n].cpu()
        assert list(f.shape) == g['shape%d' % i].tolist()
        assert_close(sample(f, 32768), g['ep%d' % i], 1e-4, 'i3d ' + n)
