"""2-D evaluation path (mirror of reference code/test_util2d.py:153-265 for the segtran model type): sliding-window inference with
sigmoid averaging, n-hot hardening, per-class Dice.  Every tensor op is a libsegx kernel: window resampling = interp_linear,
the per-window tail (resample scores -> sigmoid -> accumulate -> count) = segx_window_accum, the per-image tail
(divide, threshold, background rule) = segx_harden_segmap, Dice sums = segx_dice_sums."""
import math
import numpy as np
import torch

from . import functional as SF
from .dataloaders.datasets2d import harden_segmap2d


def _pad_hw(image_batch, hl, hr, wl, wr):
    """zero padding (F.pad constant) as a canvas copy: plumbing, no arithmetic"""
    B, C, H, W = image_batch.shape
    out = image_batch.new_zeros(B, C, H + hl + hr, W + wl + wr)
    out[:, :, hl:hl + H, wl:wl + W] = image_batch
    return out


def test_single_batch(net, image_batch, orig_input_size, patch_size, stride, task_name, num_classes, model_type='segtran'):
    """reference test_util2d.py:153-227.  Returns (preds_hard int32 [B,C,H,W], preds_soft float [B,C,H,W])."""
    if model_type not in ('segtran',):
        raise NotImplementedError("model_type '%s': only segtran is built" % model_type)
    B, C, H, W = image_batch.shape
    dx, dy = orig_input_size
    h_pad, w_pad = max(dx - H, 0), max(dy - W, 0)
    hl_pad, wl_pad = h_pad // 2, w_pad // 2
    add_pad = h_pad > 0 or w_pad > 0
    if add_pad:
        image_batch = _pad_hw(image_batch, hl_pad, h_pad - hl_pad, wl_pad, w_pad - wl_pad)
    H2, W2 = image_batch.shape[2:]
    sx = math.ceil((H2 - dx) / stride[0]) + 1
    sy = math.ceil((W2 - dy) / stride[1]) + 1
    acc = torch.zeros(B, num_classes, H2, W2, device=image_batch.device)
    cnt = torch.zeros(B, H2, W2, device=image_batch.device)
    with torch.no_grad():
        for x in range(sx):
            xs = min(stride[0] * x, H2 - dx)
            for y in range(sy):
                ys = min(stride[1] * y, W2 - dy)
                test_patch = image_batch[:, :, xs:xs + dx, ys:ys + dy]
                if tuple(patch_size) != (dx, dy):
                    test_patch = SF.interp_linear(test_patch, tuple(patch_size))
                scores_raw = net(test_patch.contiguous())
                SF.window_accum(scores_raw, acc, cnt, (xs, ys, dx, dy))          # resample to (dx, dy), sigmoid, +=, count
        preds_soft, preds_hard = SF.harden_segmap(acc, cnt, mode=0)
    if add_pad:
        preds_hard = preds_hard[:, :, hl_pad:hl_pad + H, wl_pad:wl_pad + W]
        preds_soft = preds_soft[:, :, hl_pad:hl_pad + H, wl_pad:wl_pad + W]
    return preds_hard.to(torch.int32), preds_soft


def calc_dice(predictions, gt_mask):
    """reference :233-240; (batch, H, W) -> batch of scores, (H, W) -> one score."""
    single = predictions.dim() == 2
    p = predictions.reshape((1,) + tuple(predictions.shape)) if single else predictions
    g = gt_mask.reshape((1,) + tuple(gt_mask.shape)) if single else gt_mask
    d = SF.dice_scores(p.float().reshape(p.shape[0], -1), g.float().reshape(g.shape[0], -1))
    return d[0] if single else d


def calc_batch_metric(BC_pred_soft, BC_gt, num_classes, do_calc_vcdr_error=False):
    """reference :245-273 without the vCDR column: per instance, resize the soft prediction to the mask size, harden, Dice per class."""
    if do_calc_vcdr_error:
        raise NotImplementedError('vCDR error needs the ellipse fit of utils/losses.py:calc_vcdr (cv2): outside the built path')
    batch_size = len(BC_pred_soft)
    out = np.zeros((batch_size, num_classes - 1))
    for ins in range(batch_size):
        C_pred_soft, C_gt = BC_pred_soft[ins], BC_gt[ins]
        if tuple(C_pred_soft.shape[1:]) != tuple(C_gt.shape[1:]):
            C_pred_soft = SF.interp_linear(C_pred_soft.unsqueeze(0).contiguous(), tuple(C_gt.shape[1:]))[0]
        C_pred = harden_segmap2d(C_pred_soft)
        d = SF.dice_scores(C_pred[1:].float().reshape(num_classes - 1, -1), C_gt[1:].float().reshape(num_classes - 1, -1))
        out[ins] = d.cpu().numpy()
    return out


def test_all_cases(net, batches, task_name, num_classes, orig_input_size, patch_size, stride, mask_prepred_mapping_func=None):
    """reference :20-151 reduced to its arithmetic: iterate (image_batch, mask_batch) pairs already on the device, return the
    per-class mean Dice and the instance count (file export, feature dumps and mask reloading are out of scope)."""
    total = np.zeros(num_classes - 1); count = 0
    for image_batch, mask_batch in batches:
        gt = mask_prepred_mapping_func(mask_batch) if mask_prepred_mapping_func else mask_batch
        _, preds_soft = test_single_batch(net, image_batch, orig_input_size, patch_size, stride, task_name, num_classes)
        m = calc_batch_metric(preds_soft, gt, num_classes)
        total += m.sum(axis=0); count += len(m)
    return total / max(count, 1), count


# reference function names; not pytest tests
test_single_batch.__test__ = False
test_all_cases.__test__ = False
