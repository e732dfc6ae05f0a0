"""On-device label maps of the 2D train step (mirror of reference code/dataloaders/datasets2d.py:90-139, 200-223).
Only the in-step label -> n-hot maps are on the hot path; file I/O and imgaug pipelines are out of scope."""
import torch


def fundus_map_mask(mask, exclusive=False):
    """uint8 {0,255} [B,3,H,W] (ch0 = disc region incl. cup, ch1 = cup) -> float n-hot [B,3,H,W] (bg, disc, cup)."""
    assert mask.dim() == 4 and mask.shape[1] >= 2, 'batched [B,3,H,W] masks only'
    out = torch.zeros((mask.shape[0], 3) + tuple(mask.shape[2:]), device=mask.device)
    out[:, 0] = (mask[:, 0] == 0)
    out[:, 1] = (mask[:, 0] >= 1) if not exclusive else ((mask[:, 0] >= 1) & (mask[:, 1] == 0))
    out[:, 2] = (mask[:, 1] >= 1)
    return out


def polyp_map_mask(mask, exclusive=True):
    """single 0/255 channel tiled x3 -> float [B,2,H,W] (bg, polyp)."""
    assert mask.dim() == 4
    out = torch.zeros((mask.shape[0], 2) + tuple(mask.shape[2:]), device=mask.device)
    out[:, 0] = (mask[:, 0] == 0)
    out[:, 1] = (mask[:, 0] > 0)
    return out


def harden_segmap2d(mask_soft, T=0.5):
    """reference datasets2d.py:178-196: per-class threshold; background = none of the others.  (batch, channel, h, w) or
    (channel, h, w) -> int32 0/1 maps (segx_harden_segmap)."""
    from .. import functional as SF
    batched = mask_soft.dim() == 4
    x = mask_soft if batched else mask_soft.unsqueeze(0)
    _, hard = SF.harden_segmap(x.float().contiguous(), None, mode=0, T=T, want_soft=False)
    hard = hard.to(torch.int32)
    return hard if batched else hard[0]
