"""3-D evaluation path (mirror of reference code/test_util3d.py:93-215 for the segtran model): sliding-window inference over a
volume with sigmoid averaging, the BraTS consistency rule and n-hot hardening, Dice / Jaccard per class.  Tensor arithmetic runs
in libsegx (segx_window_accum, segx_harden_segmap, segx_dice_sums, interp_linear)."""
import math
import numpy as np
import torch

from . import functional as SF


def test_single_case(net, image, orig_patch_size, input_patch_size, batch_size, stride_xy, stride_z, task_name, net_type='segtran',
                     num_classes=4):
    """reference test_util3d.py:93-184.  image [C,H,W,D] -> (preds_hard [num_classes,H,W,D] 0/1 floats, preds_soft)."""
    if net_type != 'segtran':
        raise NotImplementedError("net_type '%s': only segtran is built" % net_type)
    if task_name != 'brats':
        raise NotImplementedError('only the BraTS (n-hot, consistency rule) hardening is built; argmax tasks are not in BASELINE')
    C, H, W, D = image.shape
    dx, dy, dz = orig_patch_size
    pads = [max(dx - H, 0), max(dy - W, 0), max(dz - D, 0)]
    lp = [p // 2 for p in pads]
    add_pad = any(p > 0 for p in pads)
    if add_pad:
        padded = image.new_zeros(C, H + pads[0], W + pads[1], D + pads[2])
        padded[:, lp[0]:lp[0] + H, lp[1]:lp[1] + W, lp[2]:lp[2] + D] = image
        image = padded
    _, H2, W2, D2 = image.shape
    sx = math.ceil((H2 - dx) / stride_xy) + 1
    sy = math.ceil((W2 - dy) / stride_xy) + 1
    sz = math.ceil((D2 - dz) / stride_z) + 1
    acc = torch.zeros(1, num_classes, H2, W2, D2, device=image.device)
    cnt = torch.zeros(1, H2, W2, D2, device=image.device)
    with torch.no_grad():
        for x in range(sx):
            xs = min(stride_xy * x, H2 - dx)
            yzs_batch, patches = [], []
            for y in range(sy):
                ys = min(stride_xy * y, W2 - dy)
                for z in range(sz):
                    zs = min(stride_z * z, D2 - dz)
                    patches.append(image[:, xs:xs + dx, ys:ys + dy, zs:zs + dz])
                    yzs_batch.append((ys, zs))
                    if len(patches) == batch_size or (y == sy - 1 and z == sz - 1):
                        test_batch = torch.stack(patches, dim=0)
                        if tuple(input_patch_size) != (dx, dy, dz):
                            test_batch = SF.interp_linear(test_batch, tuple(input_patch_size))
                        scores_raw = net(test_batch.contiguous())
                        for i, (ys_i, zs_i) in enumerate(yzs_batch):
                            SF.window_accum(scores_raw[i:i + 1], acc, cnt, (xs, ys_i, zs_i, dx, dy, dz))
                        patches, yzs_batch = [], []
        preds_soft, preds_hard = SF.harden_segmap(acc, cnt, mode=1)             # make_brats_pred_consistent(False) + harden
    preds_soft, preds_hard = preds_soft[0], preds_hard[0]
    if add_pad:
        sl = (slice(None), slice(lp[0], lp[0] + H), slice(lp[1], lp[1] + W), slice(lp[2], lp[2] + D))
        preds_hard, preds_soft = preds_hard[sl].clone(), preds_soft[sl].clone()
    return preds_hard, preds_soft


def calculate_metric_percase(allcls_pred, allcls_gt, num_classes):
    """reference :186-215 (medpy): Dice and Jaccard per foreground class from the intersection / cardinality sums (binary maps);
    the surface distances (hd95, asd) need medpy's distance transforms and are reported as 0 / invalid, as the reference does for
    empty masks."""
    metric = np.zeros((num_classes - 1, 4)); valid = np.ones((num_classes - 1, 4))
    P, G = allcls_pred[1:].float(), allcls_gt[1:].float()
    sums = SF.segx.lib().dice_sums(P.reshape(num_classes - 1, -1).contiguous(), G.reshape(num_classes - 1, -1).contiguous(),
                                   num_classes - 1, P[0].numel()).cpu().numpy()
    for c in range(num_classes - 1):
        inter, ps, gs = sums[c]
        dice = 2.0 * inter / (ps + gs) if ps + gs > 0 else 0.0                  # medpy.metric.binary.dc
        if gs > 0:
            jc = inter / (ps + gs - inter)                                       # medpy.metric.binary.jc
        else:
            jc = 0.0; valid[c, 1] = 0
        valid[c, 2] = valid[c, 3] = 0
        metric[c] = [dice, jc, 0, 0]
    return metric, valid


# reference function names; not pytest tests
test_single_case.__test__ = False
