"""Evaluation-path kernels (infer.hip) vs plain PyTorch, on the fiber emulator (CPU) and on the GPU (-m gpu)."""
import pytest
import torch
import torch.nn.functional as F

from segtran_amd import functional as SF


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g, device='cpu') * scale).to(torch.get_default_device())


def close(a, b, tol=3e-5):
    s = max(b.abs().max().item(), 1e-20)
    err = (a - b).abs().max().item()
    assert err <= tol * s, 'err %.3e scale %.3e' % (err, s)


@pytest.mark.parametrize('sshape,win,canvas,origin', [((5, 7), (5, 7), (9, 12), (2, 3)),        # identity resample
                                                     ((4, 6), (9, 11), (9, 14), (0, 3)),       # bilinear up
                                                     ((3, 4, 5), (6, 7, 5), (8, 7, 9), (2, 0, 4))])   # trilinear
def test_window_accum(backend, sshape, win, canvas, origin):
    B, C = 2, 3
    s = rnd(B, C, *sshape, seed=51, scale=2.0)
    acc = rnd(B, C, *canvas, seed=52).abs(); cnt = torch.ones(B, *canvas)
    acc0, cnt0 = acc.clone().cpu(), cnt.clone().cpu()
    SF.window_accum(s, acc, cnt, tuple(origin) + tuple(win))
    with torch.device('cpu'):
        mode = 'bilinear' if len(win) == 2 else 'trilinear'
        p = torch.sigmoid(F.interpolate(s.cpu(), size=win, mode=mode, align_corners=False))
        sl = tuple(slice(o, o + w) for o, w in zip(origin, win))
        acc0[(slice(None), slice(None)) + sl] += p
        cnt0[(slice(None),) + sl] += 1
    close(acc.cpu(), acc0, 1e-6)
    assert torch.equal(cnt.cpu(), cnt0)


@pytest.mark.parametrize('mode', [0, 1])
def test_harden_segmap_modes(backend, mode):
    from oracle import segtran_oracle as O
    B, C, shape = 2, 4, (5, 6, 7)
    acc = rnd(B, C, *shape, seed=53).abs() * 2; cnt = torch.full((B,) + shape, 2.0)
    acc[0, 1:, 0, 0, 0] = 0.0                                   # a voxel with no class on -> background
    acc[0, 2, 0, 0, 1] = 1.0                                    # exactly T after the division (>= is inclusive)
    soft, hard = SF.harden_segmap(acc, cnt, mode=mode)
    with torch.device('cpu'):
        ref_soft = acc.cpu() / cnt.cpu().unsqueeze(1)
        if mode == 1:
            ref_soft = torch.stack([O.make_brats_pred_consistent(ref_soft[b], False) for b in range(B)])
        ref_hard = O.harden_segmap_nd(ref_soft, True)
    close(soft.cpu(), ref_soft, 1e-7)
    assert torch.equal(hard.cpu().int(), ref_hard)
    _, h2 = SF.harden_segmap(ref_soft.to(acc.device), None, mode=0, want_soft=False)      # cnt = None path
    assert torch.equal(h2.cpu().int(), O.harden_segmap_nd(ref_soft, True))


def test_dice_scores(backend):
    from oracle import segtran_oracle as O
    p = (rnd(5, 70, 90, seed=54) > 0).float(); g = (rnd(5, 70, 90, seed=55) > 0.3).float()
    p[4] = 0; g[4] = 0                                          # empty prediction and mask -> smooth / smooth = 1
    d = SF.dice_scores(p.reshape(5, -1), g.reshape(5, -1))
    close(d.cpu(), O.calc_dice(p.cpu(), g.cpu()), 1e-6)
    assert abs(d[4].item() - 1.0) < 1e-6
