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
