"""The trainers' data augmentation on the device (augment.hip; VERDICT r02 missing 1): the 3-D per-sample transforms against arrays produced by
the REFERENCE's own classes (tests/golden/augment3d.npz, make_golden.py case_augment3d: datasets3d.py:491-597 exec-ed from the reference file),
the stages of the 2-D pipeline (train_util.py:15-128) against hand-written known answers -- imgaug / cv2 / torchvision are not installed here.
Runs on the fiber emulator (CPU) and on the HIP build (-m gpu) through the `backend` fixture."""
import numpy as np
import torch

from segtran_amd import functional as SF
from segtran_amd.dataloaders import datasets3d as D3
from util import golden, assert_close


def cpu(v):
    """Expected values live on the host (the hip backend makes cuda the default device)."""
    return torch.tensor(v, device='cpu')


def test_rotflip_crop_3d_vs_reference_classes(backend):
    """RandomRotFlip -> RandomCrop (train3d.py:571-578): same numpy seed -> same rotation count, flip axis, pads and crop offsets as the
    reference, bit-identical voxels; as two transforms and as the fused single gather."""
    g = golden('augment3d')
    img, msk = g['image'].to(backend.dev), g['mask'].to(backend.dev)
    for seed in g['rfc_seeds'].tolist():
        for tag, out in (('crop', (8, 7, 6)), ('pad', (8, 12, 6))):
            np.random.seed(seed)
            smp = D3.RandomCrop(out)(D3.RandomRotFlip()({'image': img, 'mask': msk}))
            np.random.seed(seed)
            fused = D3.RotFlipCrop(out)({'image': img, 'mask': msk})
            for s2 in (smp, fused):
                assert torch.equal(s2['image'].cpu(), g['rfc_img_%d_%s' % (seed, tag)]), (seed, tag)
                assert torch.equal(s2['mask'].cpu(), g['rfc_msk_%d_%s' % (seed, tag)]), (seed, tag)
    np.random.seed(3)                                         # a volume without a modality axis
    smp = D3.RotFlipCrop((8, 7, 6))({'image': img[0], 'mask': msk})
    assert torch.equal(smp['image'].cpu(), g['rfc1_img']) and torch.equal(smp['mask'].cpu(), g['rfc1_msk'])


def test_random_noise_3d_vs_reference(backend):
    g = golden('augment3d')
    img = g['image'].to(backend.dev)
    z = g['noise_z'].to(backend.dev)
    for nz in (1, 0):
        out = D3.RandomNoise(mu=0.05, sigma=0.1, nonzero_only=bool(nz))({'image': img, 'mask': None}, noise=z)['image']
        assert_close(out, g['noise_out_%d' % nz], 1e-6, 'noise nz=%d' % nz)
    assert torch.equal(D3.RandomNoise()({'image': img, 'mask': None}, noise=z)['image'][:, :2].cpu(), g['image'][:, :2])
    # the device stream: clipped to +-2 sigma, zero voxels untouched, N(0, sigma) clipped in between, fresh per call
    SF.manual_seed(7)
    big = torch.ones(1, 64, 64, 16, device=backend.dev); big[:, :4] = 0
    a = D3.RandomNoise(sigma=0.1)({'image': big, 'mask': None})['image'].cpu()
    b = D3.RandomNoise(sigma=0.1)({'image': big, 'mask': None})['image'].cpu()
    d = (a - 1)[:, 4:]
    assert torch.equal(a[:, :4], torch.zeros_like(a[:, :4])) and d.abs().max() <= 0.2 + 1e-6
    assert abs(d.mean().item()) < 2e-3 and abs(d.std().item() - 0.0959) < 3e-3           # std of N(0, 0.1) clipped at 2 sigma: 0.1 * sqrt(0.9205)
    assert not torch.equal(a, b)


def test_axis_map_flips_rot90_and_pad_windows_match_numpy(backend):
    rs = np.random.RandomState(0)
    x = rs.randn(2, 3, 5, 4).astype('float32')
    xt = torch.from_numpy(x).to(backend.dev)
    for k in range(4):
        for fl in (None, 1, 2):
            m = SF.AxisMap((3, 5, 4)).rot90(k, axes=(1, 2))
            ref = np.rot90(x, k, axes=(2, 3))
            if fl is not None:
                m.flip(fl); ref = np.flip(ref, axis=fl + 1)
            assert np.array_equal(m.apply(xt).cpu().numpy(), np.ascontiguousarray(ref)), (k, fl)
    # crop 1 row at the top, pad 2 columns of zeros left and 1 right (iaa.CropAndPad with mixed signs)
    m = SF.AxisMap((3, 5, 4)).window((0, 1, -2), (3, 4, 7))
    ref = np.pad(x[:, :, 1:, :], ((0, 0), (0, 0), (0, 0), (2, 1)))
    assert np.array_equal(m.apply(xt).cpu().numpy(), ref)


def test_resize2d_known_answers(backend):
    dev = backend.dev
    x = torch.arange(16, dtype=torch.float32).reshape(1, 4, 4).to(dev)
    # nearest, cv2 convention floor(dst * in / out): 4 -> 2 picks rows / columns 0 and 2; 2x up-sampling repeats
    assert torch.equal(SF.resize2d(x, (2, 2), 'nearest').cpu(), cpu([[[0., 2.], [8., 10.]]]))
    assert torch.equal(SF.resize2d(x, (8, 8), 'nearest').cpu()[0, ::2, ::2], x.cpu()[0])
    for mode in ('nearest', 'linear', 'cubic'):                # identity size: every mode returns the input
        assert_close(SF.resize2d(x, (4, 4), mode), x, 1e-6, mode)
    # linear, half-pixel centres: 2x down-sampling of a ramp averages pixel pairs
    assert_close(SF.resize2d(x, (2, 2), 'linear'), cpu([[[2.5, 4.5], [10.5, 12.5]]]), 1e-6, 'linear down')
    big = torch.randn(2, 9, 7, generator=torch.Generator(device='cpu').manual_seed(1), device='cpu')
    ref = torch.nn.functional.interpolate(big[None], size=(13, 11), mode='bilinear', align_corners=False)[0]
    assert_close(SF.resize2d(big.to(dev), (13, 11), 'linear'), ref, 1e-5, 'linear vs F.interpolate')
    # cubic, A = -0.75: cv2.INTER_CUBIC and PyTorch's 'bicubic' share the kernel, the half-pixel convention and the replicated border
    ref = torch.nn.functional.interpolate(big[None], size=(13, 11), mode='bicubic', align_corners=False)[0]
    assert_close(SF.resize2d(big.to(dev), (13, 11), 'cubic'), ref, 1e-5, 'cubic vs F.interpolate')
    # a constant image stays constant; quantisation rounds and clamps to the uint8 range
    c = torch.full((1, 5, 6), 200.0, device=dev)
    assert torch.equal(SF.resize2d(c, (9, 4), 'cubic', quantize=True).cpu(), torch.full((1, 9, 4), 200.0, device='cpu'))
    edge = torch.tensor([[[0., 255., 0., 255.]]], device=dev).repeat(1, 4, 1)
    q = SF.resize2d(edge, (4, 9), 'cubic', quantize=True).cpu()
    assert q.min() >= 0 and q.max() <= 255 and torch.equal(q, q.round())


def test_color_ops_and_normalize_known_answers(backend):
    dev = backend.dev
    img = torch.tensor([[[[10., 200.]], [[20., 100.]], [[30., 50.]]]], device=dev)          # [1, 3, 1, 2]: two RGB pixels
    px = img.cpu().reshape(3, 2)
    luma = [0.299 * 10 + 0.587 * 20 + 0.114 * 30, 0.299 * 200 + 0.587 * 100 + 0.114 * 50]   # 18.15, 124.2
    f = torch.tensor([1.5])
    # brightness: f * x, rounded, clamped at 255
    assert torch.equal(SF.color_blend(img, 'brightness', f).cpu().reshape(3, 2), cpu([[15., 255.], [30., 150.], [45., 75.]]))
    # saturation: f * x + (1 - f) * round(luma)
    lq = cpu([float(int(luma[0] + 0.5)), float(int(luma[1] + 0.5))])
    exp = ((1.5 * px - 0.5 * lq[None, :]) + 0.5).floor().clamp(0, 255)
    assert torch.equal(SF.color_blend(img, 'saturation', f).cpu().reshape(3, 2), exp)
    # contrast: pivot = int(mean of the rounded luma image + 0.5) = int((18 + 124) / 2 + 0.5) = 71
    piv = float(int((lq[0] + lq[1]).item() / 2 + 0.5))
    exp = ((1.5 * px - 0.5 * piv) + 0.5).floor().clamp(0, 255)
    assert torch.equal(SF.color_blend(img, 'contrast', f).cpu().reshape(3, 2), exp)
    # grayscale alpha 0.25 (factor = 0.75), float arithmetic without quantisation
    y = SF.color_blend(img, 'grayscale', torch.tensor([0.75]), quantize=False).cpu().reshape(3, 2)
    assert_close(y, 0.75 * px + 0.25 * cpu([luma, luma, luma]), 1e-6, 'grayscale')
    # identity factors leave the image alone; per-sample factors are independent
    two = img.repeat(2, 1, 1, 1)
    out = SF.color_blend(two, 'brightness', torch.tensor([1.0, 0.5])).cpu()
    assert torch.equal(out[0], img.cpu()[0]) and torch.equal(out[1], (img.cpu()[0] * 0.5 + 0.5).floor())
    # ToTensor + Normalize
    n = SF.normalize(img, (0.5, 0.4, 0.3), (0.2, 0.25, 0.5)).cpu().reshape(3, 2)
    exp = (px / 255.0 - cpu([[0.5], [0.4], [0.3]])) / cpu([[0.2], [0.25], [0.5]])
    assert_close(n, exp, 1e-6, 'normalize')


def test_augment2d_pipeline_with_pinned_parameters(backend):
    """The whole 2-D pipeline (Augment2d.__call__) with the per-sample parameters pinned: the result equals the same stages composed by hand
    from numpy flips / rot90; segmentation maps follow the image's geometry with nearest-neighbour sampling."""
    from segtran_amd.dataloaders.augment2d import Augment2d
    dev = backend.dev
    rs = np.random.RandomState(4)
    img = torch.from_numpy(rs.randint(0, 256, (2, 3, 12, 12)).astype('float32')).to(dev)
    seg = torch.from_numpy((rs.rand(2, 1, 12, 12) > 0.5).astype('float32') * 255).to(dev)
    aug = Augment2d((12, 12), randscale=0.25, gray_alpha=0.0, mean=(0.4, 0.4, 0.4), std=(0.2, 0.2, 0.2), seed=0)
    ident = {'brightness': 1.0, 'contrast': 1.0, 'saturation': 1.0, 'order': [0, 1, 2]}
    params = [dict(crop=None, fliplr=True, flipud=False, rot90=1, pad_pos=(0.5, 0.5), jitter=dict(ident, brightness=1.2)),
              dict(crop=None, fliplr=False, flipud=True, rot90=2, pad_pos=(0.5, 0.5), jitter=dict(ident))]
    out, sg = aug(img, seg, params)
    x0 = np.rot90(np.flip(img[0].cpu().numpy(), axis=2), 1, axes=(1, 2))
    x0 = np.clip(np.floor(1.2 * x0 + 0.5), 0, 255)
    x1 = np.rot90(np.flip(img[1].cpu().numpy(), axis=1), 2, axes=(1, 2))
    ref = (np.stack([x0, x1]) / 255.0 - 0.4) / 0.2
    assert_close(out, torch.from_numpy(np.ascontiguousarray(ref).astype('float32')), 1e-5, 'pipeline image')
    s0 = np.rot90(np.flip(seg[0].cpu().numpy(), axis=2), 1, axes=(1, 2)); s1 = np.rot90(np.flip(seg[1].cpu().numpy(), axis=1), 2, axes=(1, 2))
    assert np.array_equal(sg.cpu().numpy(), np.stack([s0, s1]))
    # CropAndPad with keep_size: zero padding enters the image, the size stays, masks stay binary
    params = [dict(crop=(0.25, 0.0, -0.25, 0.25), fliplr=False, flipud=False, rot90=0, pad_pos=(0.5, 0.5), jitter=dict(ident))] * 2
    out, sg = aug(img, seg, params)
    assert out.shape == (2, 3, 12, 12) and sg.shape == (2, 1, 12, 12) and set(sg.cpu().unique().tolist()) <= {0.0, 255.0}
    # drawn parameters: the distributions of the reference pipeline (train_util.py:33-60)
    aug = Augment2d((12, 12), randscale=0.25, seed=1)
    ps = [aug.draw() for _ in range(2000)]
    frac = lambda f: sum(1 for p in ps if f(p)) / len(ps)      # noqa: E731
    assert abs(frac(lambda p: p['crop'] is not None) - 0.5) < 0.04 and abs(frac(lambda p: p['fliplr']) - 0.2) < 0.03
    assert abs(frac(lambda p: p['flipud']) - 0.2) < 0.03 and abs(frac(lambda p: p['rot90'] > 0) - 0.3) < 0.04
    assert all(-0.25 <= c <= 0.25 for p in ps if p['crop'] for c in p['crop']) and all(0.8 <= p['jitter']['brightness'] <= 1.2 for p in ps)
