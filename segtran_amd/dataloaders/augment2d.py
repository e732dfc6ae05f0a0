"""The 2-D trainer's augmentation pipeline on the device (mirror of reference code/train_util.py:15-128 `init_augmentation` /
`init_training_dataset` and of the per-sample application in dataloaders/datasets2d.py:444-480):

    iaa.Resize(orig_input_size) -> Sometimes(0.5, CropAndPad(percent=(-r, r), zero pad, keep_size)) -> Fliplr(0.2) -> Flipud(0.2)
    -> Sometimes(0.3, Rot90((1, 3))) -> [affine: off by default] -> PadToFixedSize -> CropToFixedSize -> Grayscale(alpha)
    -> RandomChoice(ColorJitter brightness 0.2 | contrast 0.2 | saturation 0.2 | all three 0.1) -> ToTensor -> Normalize(mean, std)

The reference runs it per sample on uint8 arrays inside DataLoader workers (imgaug + PIL / torchvision, neither installed here); this class
runs it per BATCH on tensors resident in HBM: images float [B, 3, H, W] on the 0..255 scale (quantised between stages like the uint8 arrays the
reference carries), segmentation maps [B, C, H, W] with the SAME geometric parameters and nearest-neighbour resampling.  Every stage is one
libsegx pass (segx_resize2d, segx_axis_gather, segx_color_blend, segx_normalize).  The random parameters are drawn from a numpy RandomState with
the reference's distributions (imgaug's own stream cannot be reproduced without imgaug); `draw()` exposes them so tests can pin them."""
import numpy as np
import torch

from .. import functional as SF


class Augment2d:
    def __init__(self, orig_input_size, randscale=0.0, gray_alpha=0.5, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), seed=None):
        self.tgt_w, self.tgt_h = (int(v) for v in orig_input_size)             # train_util.py:23 (width, height)
        self.crop_percents = (-randscale, randscale) if randscale > 0 else (0.0, 0.0)
        self.gray_alpha, self.mean, self.std = float(gray_alpha), tuple(mean), tuple(std)
        self.rs = np.random.RandomState(seed)

    def draw(self):
        """the random parameters of ONE sample"""
        rs, p = self.rs, {}
        p['crop'] = tuple(rs.uniform(*self.crop_percents, size=4)) if rs.uniform() < 0.5 else None      # (top, right, bottom, left) fractions
        p['fliplr'], p['flipud'] = rs.uniform() < 0.2, rs.uniform() < 0.2
        p['rot90'] = int(rs.randint(1, 4)) if rs.uniform() < 0.3 else 0
        p['pad_pos'] = (rs.uniform(), rs.uniform())                                                    # PadTo / CropToFixedSize position='uniform'
        choice = int(rs.randint(0, 4))
        jit = {'brightness': 1.0, 'contrast': 1.0, 'saturation': 1.0, 'order': [0, 1, 2]}
        if choice == 0:
            jit['brightness'] = rs.uniform(0.8, 1.2)
        elif choice == 1:
            jit['contrast'] = rs.uniform(0.8, 1.2)
        elif choice == 2:
            jit['saturation'] = rs.uniform(0.8, 1.2)
        else:
            jit.update(brightness=rs.uniform(0.9, 1.1), contrast=rs.uniform(0.9, 1.1), saturation=rs.uniform(0.9, 1.1), order=list(rs.permutation(3)))
        p['jitter'] = jit
        return p

    def geometric(self, x, p, mode):
        """one sample [C, h, w] -> [C, tgt_h, tgt_w]; mode 'cubic' (image, quantised) or 'nearest' (segmentation map)"""
        q = mode == 'cubic'
        H, W = self.tgt_h, self.tgt_w
        ap = lambda m, t: m.apply(t.unsqueeze(1)).squeeze(1)          # noqa: E731  [C, h, w] as C planes of extent (1, h, w)
        if tuple(x.shape[-2:]) != (H, W):
            x = SF.resize2d(x, (H, W), mode, q)
        if p['crop'] is not None:
            # iaa.CropAndPad(percent): negative = crop, positive = zero pad, per side, in pixels of the current size; keep_size resizes back
            t, r, b, l = (int(round(f * n)) for f, n in zip(p['crop'], (H, W, H, W)))
            m = SF.AxisMap((1, H, W)).window((0, -t, -l), (1, H + t + b, W + l + r))
            x = SF.resize2d(ap(m, x), (H, W), mode, q)
        m = SF.AxisMap((1,) + tuple(x.shape[-2:]))
        if p['fliplr']:
            m.flip(2)
        if p['flipud']:
            m.flip(1)
        if p['rot90']:
            m.rot90(p['rot90'], axes=(1, 2))
        h2, w2 = m.O[1], m.O[2]
        if (h2, w2) != (H, W):           # PadToFixedSize then CropToFixedSize (only after a rotation of a non-square target)
            py, px = max(H - h2, 0), max(W - w2, 0)
            top, left = int(round(py * p['pad_pos'][1])), int(round(px * p['pad_pos'][0]))
            hp, wp = h2 + py, w2 + px
            cy, cx = int(round((hp - H) * p['pad_pos'][1])), int(round((wp - W) * p['pad_pos'][0]))
            m.window((0, cy - top, cx - left), (1, H, W))
        return ap(m, x) if (p['fliplr'] or p['flipud'] or p['rot90'] or (h2, w2) != (H, W)) else x

    def photometric(self, img, params):
        """batch [B, 3, H, W] on the 0..255 scale -> normalised float batch"""
        B, dev = img.shape[0], img.device
        if self.gray_alpha > 0:
            img = SF.color_blend(img, 'grayscale', torch.full((B,), 1.0 - self.gray_alpha), True)
        ops = ('brightness', 'contrast', 'saturation')
        for step in range(3):            # ColorJitter applies its three factors in a per-sample random order: one pass per step, identity factors where unused
            for op in ops:
                idx = [b for b in range(B) if ops[params[b]['jitter']['order'][step]] == op and params[b]['jitter'][op] != 1.0]
                if idx:
                    f = torch.ones(B)
                    for b in idx:
                        f[b] = float(params[b]['jitter'][op])
                    img = SF.color_blend(img, op, f, True)
        return SF.normalize(img, self.mean, self.std, 1.0 / 255.0)

    def __call__(self, images, segmaps=None, params=None):
        """images [B, 3, h, w] (0..255), segmaps [B, C, h, w] or None -> (normalised images [B, 3, H, W], segmaps [B, C, H, W])"""
        B = images.shape[0]
        params = params if params is not None else [self.draw() for _ in range(B)]
        img = torch.stack([self.geometric(images[b].float(), params[b], 'cubic') for b in range(B)])
        seg = None if segmaps is None else torch.stack([self.geometric(segmaps[b].float(), params[b], 'nearest') for b in range(B)])
        return self.photometric(img, params), seg


def load_png_pair(image_path, mask_path=None, device='cuda'):
    """The file side of datasets2d.py:444-462 (`np.array(Image.open(...))`): decode on the host with PIL into PINNED memory and hand the
    device a [3, H, W] float tensor (0..255 scale) + the [C, H, W] mask.  (The 3-D loader reads .h5 volumes through h5py, which this image
    does not ship: datasets3d file decode raises instead of guessing.)"""
    from PIL import Image
    im = torch.from_numpy(np.array(Image.open(image_path).convert('RGB'))).permute(2, 0, 1).contiguous()
    out = [im.pin_memory().to(device, non_blocking=True).float() if torch.cuda.is_available() else im.float()]
    if mask_path is not None:
        mk = np.array(Image.open(mask_path))
        mk = torch.from_numpy(mk if mk.ndim == 3 else mk[:, :, None]).permute(2, 0, 1).contiguous()
        out.append(mk.pin_memory().to(device, non_blocking=True).float() if torch.cuda.is_available() else mk.float())
    return out if mask_path is not None else out[0]
