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
