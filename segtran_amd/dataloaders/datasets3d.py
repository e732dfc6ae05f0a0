"""On-device label map and augmentation of the 3D train step (mirror of reference code/dataloaders/datasets3d.py:16-40, 491-597, 611-665).
The reference hard-codes device='cuda' (N8); here the label's own device is used."""
import numpy as np
import torch


# ---- per-sample transforms of the 3-D trainer (train3d.py:571-578: RandomRotFlip -> RandomCrop -> ToTensor; RandomNoise is commented out
# there but shipped).  Same `sample = {'image': [C, H, W, D] or [H, W, D], 'mask': [H, W, D]}` protocol and the SAME draws from numpy's
# global generator in the same order as the reference, so np.random.seed(s) reproduces its crops / rotations / flips; the arrays are DEVICE
# tensors and every geometric transform is one gather (functional.AxisMap -> segx_axis_gather).  Chained RandomRotFlip + RandomCrop can be
# fused into a single pass with `RotFlipCrop`.
def _maps(sample):
    from .. import functional as SF
    image, mask = sample['image'], sample['mask']
    assert image.dim() == mask.dim() or image.dim() == mask.dim() + 1
    return SF.AxisMap(image.shape[-3:]), SF.AxisMap(mask.shape[-3:])


class RandomRotFlip(object):
    """reference datasets3d.py:547-579: rotate by k * 90 degrees in the (H, W) plane, then flip along a random axis."""

    def draw(self):
        k = np.random.randint(0, 4)
        axis = np.random.randint(0, 3)
        return k, axis

    def extend(self, maps, k, axis):
        for m in maps:
            m.rot90(k, axes=(0, 1)).flip(axis)

    def __call__(self, sample):
        mi, mm = _maps(sample)
        self.extend((mi, mm), *self.draw())
        return {'image': mi.apply(sample['image']), 'mask': mm.apply(sample['mask'])}


class RandomCrop(object):
    """reference datasets3d.py:491-545: zero-pad by (out - n) // 2 + 3 per side where an axis is not larger than the patch, then a random crop."""

    def __init__(self, output_size):
        self.output_size = tuple(int(v) for v in output_size)

    def draw(self, shape):
        o = self.output_size
        pads = (0, 0, 0)
        if shape[0] <= o[0] or shape[1] <= o[1] or shape[2] <= o[2]:
            pads = tuple(max((o[a] - shape[a]) // 2 + 3, 0) for a in range(3))
        h, w, d = (shape[a] + 2 * pads[a] for a in range(3))
        h1 = np.random.randint(0, h - o[0]); w1 = np.random.randint(0, w - o[1]); d1 = np.random.randint(0, d - o[2])
        return pads, (h1, w1, d1)

    def extend(self, maps, pads, starts):
        for m in maps:
            m.window([s - p for s, p in zip(starts, pads)], self.output_size)

    def __call__(self, sample):
        mi, mm = _maps(sample)
        self.extend((mi, mm), *self.draw(mm.O))
        return {'image': mi.apply(sample['image']), 'mask': mm.apply(sample['mask'])}


class RotFlipCrop(object):
    """RandomRotFlip followed by RandomCrop (the trainer's composition) as ONE gather per tensor: same draws in the same order."""

    def __init__(self, output_size):
        self.rf, self.rc = RandomRotFlip(), RandomCrop(output_size)

    def __call__(self, sample):
        mi, mm = _maps(sample)
        self.rf.extend((mi, mm), *self.rf.draw())
        self.rc.extend((mi, mm), *self.rc.draw(mm.O))
        return {'image': mi.apply(sample['image']), 'mask': mm.apply(sample['mask'])}


def collate_augmented(samples, patch_size, noise=None):
    """The DataLoader side of train3d.py:571-578 + the default collate for a list of per-sample dicts ALREADY on the device:
    RandomRotFlip -> RandomCrop(patch_size) as one gather per tensor (optionally RandomNoise), stacked to
    (volumes [B, C, *patch], labels [B, *patch]) -- what TrainStep takes.  Draws come from numpy's global generator, sample by sample."""
    tf = RotFlipCrop(patch_size)
    out = [tf(s) for s in samples]
    if noise is not None:
        out = [noise(s) for s in out]
    vol = torch.stack([s['image'] if s['image'].dim() == 4 else s['image'][None] for s in out])
    return vol, torch.stack([s['mask'] for s in out])


class RandomNoise(object):
    """reference datasets3d.py:581-597: image += clip(sigma * randn, +-2 sigma) + mu on the non-zero voxels.  The normal field comes from the
    device Philox stream (functional.manual_seed); pass `noise=` to __call__ to inject a given field (numpy's stream cannot be reproduced)."""

    def __init__(self, mu=0, sigma=0.1, nonzero_only=True):
        self.mu, self.sigma, self.nonzero_only = mu, sigma, nonzero_only

    def __call__(self, sample, noise=None):
        from .. import functional as SF
        return {'image': SF.add_noise(sample['image'], self.mu, self.sigma, self.nonzero_only, noise), 'mask': sample['mask']}


def RandomResizedCrop(volume, mask, out_size, crop_percents, isotropic=True):
    """reference datasets3d.py:611-665 (train3d.py:713-715, --randscale): random isotropic rescale of the batch by a factor in
    [1 + crop_percents[0], 1 + crop_percents[1]] (trilinear, the n-hot mask is resampled too and STAYS continuous, :661-662),
    zero padding back up to `out_size` where the rescaled volume is smaller, then a random crop of `out_size`.
    volume [B, C, H, W, D], mask [B, K, H, W, D] (float n-hot) -> ([B, C, *out_size], [B, K, *out_size]).

    Random numbers: the SAME draws in the same order from torch's CPU generator as the reference (`torch.rand(1)` per scale,
    three `torch.randint`), so a seeded run crops where the reference crops.  Arithmetic: one libsegx gather pass per tensor
    (segx_resized_crop3d) that computes only the cropped window; the reference's resampled and padded intermediates never exist."""
    from .. import functional as SF
    H, W, D = (int(v) for v in volume.shape[-3:])
    min_scale, max_scale = 1 + crop_percents[0], 1 + crop_percents[1]
    # device='cpu': the reference draws from torch's CPU generator (default device); never from the GPU one, whatever the default device is
    scale_H = torch.rand(1, device='cpu') * (max_scale - min_scale) + min_scale
    if isotropic:
        scale_W = scale_D = scale_H
    else:
        scale_W = torch.rand(1, device='cpu') * (max_scale - min_scale) + min_scale
        scale_D = torch.rand(1, device='cpu') * (max_scale - min_scale) + min_scale
    H2, W2, D2 = int(H * scale_H), int(W * scale_W), int(D * scale_D)          # float32 tensor arithmetic, truncated: as the reference
    Ho, Wo, Do = (int(v) for v in out_size)
    pads = [max(o - n, 0) // 2 for o, n in ((Ho, H2), (Wo, W2), (Do, D2))]      # front pads (:637-645)
    Hp, Wp, Dp = max(H2, Ho), max(W2, Wo), max(D2, Do)                          # padded extents
    starts = [int(torch.randint(n, (1,), device='cpu')) for n in (Hp - Ho + 1, Wp - Wo + 1, Dp - Do + 1)]
    offs = [s - p for s, p in zip(starts, pads)]
    return (SF.resized_crop3d(volume, (H2, W2, D2), (Ho, Wo, Do), offs), SF.resized_crop3d(mask, (H2, W2, D2), (Ho, Wo, Do), offs))


def brats_map_label(mask, binarize=False):
    """int labels [B,H,W,D] (1 NCR/NET, 2 ED, 3 ET, 0 rest) -> float n-hot [B,C,H,W,D]: (bg, ET, WT, TC) or (bg, tumour)."""
    nc = 2 if binarize else 4
    out = torch.zeros((nc,) + tuple(mask.shape), device=mask.device)
    if binarize:
        out[0, mask == 0] = 1
        out[1, mask > 0] = 1
    else:
        out[0, mask == 0] = 1
        out[1, mask == 3] = 1
        out[2, (mask == 3) | (mask == 1) | (mask == 2)] = 1
        out[3, (mask == 3) | (mask == 1)] = 1
    if out.dim() == 5:
        out = out.permute(1, 0, 2, 3, 4)
    return out.contiguous()


def make_brats_pred_consistent(preds_soft, is_conservative):
    """reference datasets3d.py:43-63; preds_soft [4, ...] = (bg, ET, WT, TC).  The permissive rule (is_conservative=False, the one
    test_single_case uses) runs fused with the hardening in segx_harden_segmap; this standalone form covers both rules."""
    out = preds_soft.clone()
    if is_conservative:
        out[1] = torch.min(preds_soft[1:], dim=0)[0]
        out[3] = torch.min(preds_soft[2:], dim=0)[0]
    else:
        out[2] = torch.max(preds_soft[1:], dim=0)[0]
        out[3] = torch.max(preds_soft[[1, 3]], dim=0)[0]
    return out


def brats_inv_map_label(orig_probs):
    """reference datasets3d.py:65-90: n-hot (bg, ET, WT, TC) probabilities -> exclusive label probabilities (0, 1, 2, 3)."""
    inv = torch.zeros_like(orig_probs)
    inv[0] = 1 - orig_probs[2]
    inv[3] = orig_probs[1]
    inv[1] = (orig_probs[3] - orig_probs[1]) * 1.5
    inv[2] = (orig_probs[2] - orig_probs[3]) * 1.5
    return inv


def harden_segmap3d(mask_soft, T=0.5):
    """reference datasets3d.py:92-111: (batch, channel, h, w, d) or (channel, h, w, d) -> int32 0/1 maps (segx_harden_segmap)."""
    from .. import functional as SF
    batched = mask_soft.dim() == 5
    x = mask_soft if batched else mask_soft.unsqueeze(0)
    _, hard = SF.harden_segmap(x.float().contiguous(), None, mode=0, T=T, want_soft=False)
    hard = hard.to(torch.int32)
    return hard if batched else hard[0]
