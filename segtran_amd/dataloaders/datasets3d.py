"""On-device label map of the 3D train step (mirror of reference code/dataloaders/datasets3d.py:16-40).
The reference hard-codes device='cuda' (N8); here the label's own device is used."""
import torch


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
