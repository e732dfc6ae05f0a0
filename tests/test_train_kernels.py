"""CPU: fused loss + multi-tensor BertAdam kernels on the emulator vs the golden fixtures from the reference."""
import pytest
import torch
from segtran_amd import segx
from util import golden, assert_close


def golden_on(name, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in golden(name).items()}


@pytest.mark.parametrize('tag', ['2d', '3d'])
def test_seg_loss_vs_reference(backend, tag):
    from segtran_amd import functional as SF
    g = golden_on('loss', backend.dev)
    lo = g['logits' + tag].clone().requires_grad_(True)
    nc = lo.shape[1]
    cw = torch.ones(nc); cw[0] = 0; cw /= cw.sum()
    loss, stats = SF.seg_loss(lo, g['mask' + tag], g['pw' + tag], cw)
    assert abs(loss.item() - float(g['loss' + tag])) < 2e-6
    assert abs(stats[1].item() - float(g['ce' + tag])) < 2e-6 and abs(stats[2].item() - float(g['dice' + tag])) < 2e-6
    (loss * 1.0).backward()
    assert_close(lo.grad, g['dlogits' + tag], 2e-5, 'dlogits')


@pytest.mark.parametrize('private', [False, True])
def test_bertadam_vs_reference(backend, private):
    """private=False: gradients are views of the flat buffer (data-parallel mode); private=True: autograd owns the gradient
    tensors and the kernel follows a per-step pointer table (single-process mode) -- same numbers either way."""
    from segtran_amd.optimization import BertAdam
    g = golden_on('bertadam', backend.dev)
    params = [torch.nn.Parameter(g['p0_%d' % i].clone()) for i in range(4)]
    groups = [dict(params=[params[0], params[3]], weight_decay=1e-4, lr=2e-4),
              dict(params=[params[1]], weight_decay=1e-5, lr=2e-4),
              dict(params=[params[2]], weight_decay=0.0, lr=2e-4)]
    opt = BertAdam(groups, lr=2e-4, warmup=0.25, t_total=8, weight_decay=1e-4, global_grad_clip=0.1)
    if private:
        opt.release_flat_grads()
        assert all(p.grad is None for p in params)
    order = [0, 3, 1, 2]                                    # optimizer-internal order = group order
    for step in range(4):
        opt.zero_grad()
        # parameters 0..2 receive gradients through autograd; parameter 3 never does (N3)
        loss = sum((params[i] * g['g%d_%d' % (step, i)]).sum() for i in range(3))
        loss.backward()
        if private:
            assert params[3].grad is None and all(params[i].grad.data_ptr() != 0 for i in range(3))
        opt.step()
        for i in range(4):
            assert torch.allclose(params[i].data, g['p%d_%d' % (step + 1, i)], atol=2e-7), (step, i)
    assert torch.equal(params[3].data, g['p0_3'])           # untouched: no update, no weight decay


def test_mt_gather_copies_gradients_into_their_flat_slices(backend):
    """segx_mt_gather: a chunk range of (possibly mis-aligned, possibly absent) source tensors -> the optimizer's flat slices."""
    from segtran_amd.optimization import BertAdam, CHUNK
    dev = backend.dev
    sizes = [70000, 5, 131072 + 3, 9]                       # spans several chunks / tails that are not multiples of 4
    params = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in sizes]
    opt = BertAdam([dict(params=params, weight_decay=0.0, lr=1e-3)], lr=1e-3, warmup=-1, t_total=-1)
    opt.use_gathered_grads()
    sum((p * 1.0).sum() for p in params[:3]).backward()      # parameter 3 never receives a gradient
    opt._ensure_tables()
    g = torch.Generator(device='cpu').manual_seed(4)
    big = torch.randn(70000 + 1, generator=g, device='cpu').to(dev)
    srcs = [big[1:], torch.randn(5, generator=g, device='cpu').to(dev), torch.randn(sizes[2], generator=g, device='cpu').to(dev), None]
    assert srcs[0].data_ptr() % 16 != 0                     # a 4-byte-aligned view: the scalar copy path
    tab = torch.tensor([0 if t is None else t.data_ptr() for t in srcs], dtype=torch.int64, device='cpu').to(dev)
    opt.flat_grad.fill_(-7.0)
    L = segx.lib()
    cf = opt._chunk_first_host
    L.mt_gather(tab, opt._tabs, cf[0], cf[2] - cf[0], CHUNK)            # tensors 0 and 1 only
    for i, (off, n) in enumerate(opt.slices):
        got = opt.flat_grad[off:off + n]
        if i < 2:
            assert torch.equal(got, srcs[i])
        else:
            assert (got == -7.0).all()
    L.mt_gather(tab, opt._tabs, cf[2], cf[4] - cf[2], CHUNK)            # tensor 2, and tensor 3 (no source: untouched)
    off, n = opt.slices[2]
    assert torch.equal(opt.flat_grad[off:off + n], srcs[2])
    off, n = opt.slices[3]
    assert (opt.flat_grad[off:off + n] == -7.0).all()


def test_bertadam_state_dict_round_trip_continues_the_reference_trajectory(backend):
    """moments and the schedule position live in flat buffers -- state_dict()/load_state_dict() must carry them, in the
    reference's per-parameter layout (state[p] = {'step', 'next_m', 'next_v'}, optimization.py:108-114).  Two steps, checkpoint,
    a NEW optimizer on cloned parameters, two more steps: still the golden trajectory of four uninterrupted reference steps."""
    from segtran_amd.optimization import BertAdam
    g = golden_on('bertadam', backend.dev)

    def make(ps):
        groups = [dict(params=[ps[0], ps[3]], weight_decay=1e-4, lr=2e-4), dict(params=[ps[1]], weight_decay=1e-5, lr=2e-4),
                  dict(params=[ps[2]], weight_decay=0.0, lr=2e-4)]
        return BertAdam(groups, lr=2e-4, warmup=0.25, t_total=8, weight_decay=1e-4, global_grad_clip=0.1)

    def run(opt, ps, steps):
        for step in steps:
            opt.zero_grad()
            sum((ps[i] * g['g%d_%d' % (step, i)]).sum() for i in range(3)).backward()
            opt.step()

    ps = [torch.nn.Parameter(g['p0_%d' % i].clone()) for i in range(4)]
    opt = make(ps)
    opt.release_flat_grads()
    run(opt, ps, (0, 1))
    sd = opt.state_dict()
    st = sd['state']
    assert sorted(st) == [0, 2, 3], 'the gradient-less parameter (index 1 = params[3]) carries no state (N3)'
    assert set(st[0]) == {'step', 'next_m', 'next_v'} and st[0]['step'] == 2 and st[0]['next_m'].shape == ps[0].shape
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = make(ps2)
    opt2.release_flat_grads()
    opt2.load_state_dict(sd)
    assert opt2.step_count == 2 and opt2.get_lr() == opt.get_lr()
    run(opt2, ps2, (2, 3))
    for i in range(4):
        assert torch.allclose(ps2[i].data, g['p4_%d' % i], atol=2e-7), i


def test_bertadam_follows_param_group_edits_and_rejects_mixed_schedules(backend):
    from segtran_amd.optimization import BertAdam
    dev = backend.dev
    w = torch.nn.Parameter(torch.ones(8, device=dev))
    opt = BertAdam([dict(params=[w], weight_decay=0.0, lr=1e-2)], lr=1e-2, warmup=-1, t_total=-1, max_grad_norm=-1)
    opt.release_flat_grads()

    def one():
        opt.zero_grad(); (w * 1.0).sum().backward(); before = w.detach().clone(); opt.step(); return (before - w.detach()).abs().max().item()
    d1 = one()
    opt.param_groups[0]['lr'] = 1e-3                        # an LR scheduler / manual decay edits the group AFTER the tables were built
    d2 = one()
    # constant gradient 1, no bias correction: m/sqrt(v) = 0.1/sqrt(0.001) at step 1 and 0.19/sqrt(0.001999) at step 2; the step scales with lr
    want = 0.1 * (0.19 / 0.001999 ** 0.5) / (0.1 / 0.001 ** 0.5)
    assert d1 > 0 and abs(d2 / d1 - want) < 1e-3, (d1, d2, want)
    a, b = torch.nn.Parameter(torch.ones(4, device=dev)), torch.nn.Parameter(torch.ones(4, device=dev))
    bad = BertAdam([dict(params=[a], lr=1e-3), dict(params=[b], lr=1e-3, b1=0.5)], lr=1e-3, warmup=-1, t_total=-1)
    bad.release_flat_grads()
    bad.zero_grad(); (a.sum() + b.sum()).backward()
    with pytest.raises(ValueError, match="share 'b1'"):
        bad.step()


def test_reference_format_optimizer_state_with_four_groups_loads_through_load_model(backend, tmp_path):
    """the reference ALWAYS writes four param_groups -- normal, low_decay, an empty no_decay, high_lr (train2d.py:536-541) -- and
    torch matches groups by position, so init_optimizer must emit the same four (it emitted two or three and `--cp` with a reference 'optim_state'
    raised "different number of parameter groups").  A checkpoint in the reference's wire format goes through train_common.load_model here."""
    import argparse
    from segtran_amd import engine, train_common
    dev = backend.dev

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Linear(4, 4)
            self.head = torch.nn.Linear(4, 2)
    torch.manual_seed(0)
    net = Net().to(dev)
    opt = engine.init_optimizer(net, 'fundus', t_total=100, warmup_steps=10)
    assert [len(g['params']) for g in opt.param_groups] == [2, 2, 0, 0]
    assert [g['weight_decay'] for g in opt.param_groups[:2]] == [opt.defaults['weight_decay'], opt.defaults['weight_decay'] * 0.1]
    # what the reference's BertAdam.state_dict() holds after 7 steps: per-parameter {'step', 'next_m', 'next_v'}, four groups by position
    names = ['head.weight', 'head.bias', 'backbone.weight', 'backbone.bias']
    ps = dict(net.named_parameters())
    g = torch.Generator(device='cpu').manual_seed(5)
    state = {i: dict(step=7, next_m=torch.randn(ps[n].shape, generator=g, device='cpu'), next_v=torch.rand(ps[n].shape, generator=g, device='cpu')) for i, n in enumerate(names)}
    base = dict(schedule='warmup_linear', warmup=0.1, t_total=100, b1=0.9, b2=0.999, e=1e-6, max_grad_norm=0.05)
    groups = [dict(base, params=[0, 1], weight_decay=1e-4, lr=2e-4), dict(base, params=[2, 3], weight_decay=1e-5, lr=2e-4),
              dict(base, params=[], weight_decay=0.0, lr=2e-4), dict(base, params=[], weight_decay=0.0, lr=2e-2)]
    path = str(tmp_path / 'iter_7.pth')
    torch.save({'iter_num': 7, 'model': {k: v.cpu() for k, v in net.state_dict().items()}, 'optim_state': dict(state=state, param_groups=groups)}, path)
    args = argparse.Namespace(net='unet', lr_warmup_steps=10)
    assert train_common.load_model(net, args, path, optimizer=opt, load_optim_state=True) == 7
    assert opt.step_count == 7 and args.lr_warmup_steps == 0
    for i, n in enumerate(names):
        off, cnt = opt.slices[[id(p) for p, _ in opt._all_params()].index(id(ps[n]))]
        assert torch.equal(opt.flat_m[off:off + cnt].cpu(), state[i]['next_m'].reshape(-1)), n
        assert torch.equal(opt.flat_v[off:off + cnt].cpu(), state[i]['next_v'].reshape(-1)), n
