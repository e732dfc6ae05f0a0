"""Segtran2d -- host-side mirror of /root/reference/code/networks/segtran2d.py (`--net segtran`, 2D).

Same constructor (`Segtran2d(config)` with `CONFIG.update_config(args)`), same forward signature
([B,3,H,W] -> [B,num_classes,H,W] logits), same attribute surface read by the trainer
(`voxel_fusion.translayers[i]...`, `layers_attn_scores`, `orig_feat_shape`, `feature_maps`,
`backbone._blocks`, `backbone.endpoint_blk_indices`, `num_vis_layers`) and the same state_dict keys.
Built: eff-b* backbones, in_fpn '34' / out_fpn '1234' with scheme 'AN' and GroupNorm (the values the
trainer forces / defaults to).  ResNet / EfficientNet-V2 backbones, BN FPNs, modalities and the global-bias
ablation are outside the hot path and raise.

Kernel status: fusion encoder, every 1x1 conv (MFMA GEMM), GroupNorm and the bilinear resampling with the fused
lateral addition and the foreground-mask pooling (K15) all run on libsegx.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from argparse import Namespace

from .. import functional as SF
from ..efficientnet.model import EfficientNet
from .segtran_shared import (SegtranConfig, bb2feat_dims, SegtranFusionEncoder, CrossAttFeatTrans,  # noqa: F401
                             ExpandedFeatTrans, SegtranInitWeights, gen_all_indices, gen_scaled_positions)


class Segtran2dConfig(SegtranConfig):
    def __init__(self):
        super().__init__()
        self.backbone_type = 'eff-b4'
        self.use_pretrained = True
        self.bb_feat_dims = bb2feat_dims[self.backbone_type]
        self.num_translayers = 1
        self.set_fpn_layers('default', Namespace(in_fpn_layers='34', out_fpn_layers='1234', in_fpn_scheme='AN',
                                                 out_fpn_scheme='AN', translayer_compress_ratios=[1, 1]), do_print=False)
        self.bb_feat_upsize = True
        self.in_fpn_use_bn = False
        self.out_fpn_use_bn = False
        self.resnet_bn_to_gn = False
        self.G = 8
        self.pos_dim = 2
        self.max_pos_size = (100, 100)
        self.num_classes = 2
        self.num_modalities = 0
        self.use_global_bias = False
        self.device = 'cuda'

    def update_config(self, args):
        self.try_assign(args, 'num_classes', 'backbone_type', 'use_pretrained', 'bb_feat_upsize', 'in_fpn_use_bn',
                        'use_squeezed_transformer', 'num_attractors', 'num_translayers', 'num_modes',
                        'trans_output_type', 'mid_type', 'pos_code_type', 'pos_code_weight', 'pos_bias_radius',
                        'ablate_multihead', 'out_fpn_do_dropout', 'has_FFN_in_squeeze', 'attn_clip', 'qk_have_bias',
                        'tie_qk_scheme', 'num_modalities', 'device', 'eval_robustness', 'use_global_bias',
                        'use_attn_consist_loss', 'use_mince_transformer', 'mince_scales', 'mince_channel_props')
        a = args if isinstance(args, dict) else args.__dict__
        if 'dropout_prob' in a and a['dropout_prob'] >= 0:
            self.hidden_dropout_prob = a['dropout_prob']
            self.attention_probs_dropout_prob = a['dropout_prob']
        self.bb_feat_dims = bb2feat_dims[self.backbone_type]
        self.set_fpn_layers('args', args, do_print=False)


CONFIG = Segtran2dConfig()


class _Conv1x1(nn.Conv2d):
    """nn.Conv2d(k=1) parameter container whose forward is libsegx's MFMA GEMM."""

    def forward(self, x):
        return SF.conv1x1(x, self.weight, self.bias)


def _up(x, size, base=None):
    """bilinear resize (align_corners=False) with the FPN lateral `base` added in the same libsegx pass."""
    return SF.interp_linear(x, size, base)


class Segtran2d(SegtranInitWeights):
    def __init__(self, config):
        super().__init__(config)
        self.config = config
        self.device = config.device
        self.trans_in_dim, self.trans_out_dim = config.trans_in_dim, config.trans_out_dim
        self.num_translayers = config.num_translayers
        self.bb_feat_upsize = config.bb_feat_upsize
        self.G = config.G
        if config.use_global_bias or config.num_modalities > 0 or config.out_fpn_use_bn:
            raise NotImplementedError('global-bias / multi-modality / BN-out-FPN variants are outside the hot path')
        self.use_global_bias = False
        self.voxel_fusion = SegtranFusionEncoder(config, 'Fusion')
        self.backbone_type = config.backbone_type
        if not self.backbone_type.startswith('eff-'):
            raise NotImplementedError("backbone '%s': only EfficientNet-V1 (eff-b0..b7) is built" % self.backbone_type)
        stem_stride = 1 if self.bb_feat_upsize else 2
        self.use_pretrained = config.use_pretrained
        bb_name = self.backbone_type.replace('eff', 'efficientnet')
        if self.use_pretrained:                                 # segtran2d.py:98-101 (advprop checkpoint, local file: no network)
            self.backbone = EfficientNet.from_pretrained(bb_name, advprop=True, ignore_missing_keys=True, stem_stride=stem_stride)
        else:
            self.backbone = EfficientNet.from_name(bb_name, stem_stride=stem_stride)
        self.in_fpn_layers, self.in_fpn_scheme = config.in_fpn_layers, config.in_fpn_scheme
        self.out_fpn_layers, self.out_fpn_scheme = config.out_fpn_layers, config.out_fpn_scheme
        if self.in_fpn_layers != [3, 4] or self.out_fpn_layers != [1, 2, 3, 4] or self.in_fpn_scheme != 'AN' \
                or self.out_fpn_scheme != 'AN':
            raise NotImplementedError("only --infpn 34 --outfpn 1234 with the 'AN' scheme (reference defaults) are built")
        pool_stride = 2 ** int(np.min(self.in_fpn_layers)) * (1 if self.bb_feat_upsize else 2)
        self.mask_pool = nn.AvgPool2d((pool_stride, pool_stride))
        d = self.bb_feat_dims = config.bb_feat_dims
        self.in_fpn23_conv = _Conv1x1(d[2], d[3], 1)            # unused with in_fpn '34' (N3), kept for the checkpoint
        self.in_fpn34_conv = _Conv1x1(d[3], d[4], 1)
        self.in_fpn_bridgeconv = _Conv1x1(d[4], self.trans_in_dim, 1) if d[4] != self.trans_in_dim else nn.Identity()
        self.in_fpn_use_bn = config.in_fpn_use_bn
        if self.in_fpn_use_bn:                                  # --inbn (segtran2d.py:143-146): default BatchNorm2d (eps 1e-5, momentum 0.1)
            self.in_bn3b = nn.BatchNorm2d(d[3])
            self.in_bn4b = nn.BatchNorm2d(d[4])
        else:
            self.in_gn3b = nn.GroupNorm(self.G, d[3])
            self.in_gn4b = nn.GroupNorm(self.G, d[4])
        self.out_fpn_do_dropout = config.out_fpn_do_dropout     # --outdrop (:308-310)
        # out_fpn_bridgeconv -> (+ up-sampled transformer output) -> out_conv is a chain of LINEAR maps: by default the class
        # projection is composed into the bridge weights and applied before the up-sampling (see out_head_forward)
        self.fuse_output_tail = True
        self.num_classes = config.num_classes
        self.num_modalities = 0
        self.do_out_fpn = True
        self.out_fpn12_conv = _Conv1x1(d[1], d[2], 1)
        self.out_fpn23_conv = _Conv1x1(d[2], d[3], 1)
        self.out_fpn34_conv = _Conv1x1(d[3], d[4], 1)            # unused (N3)
        self.out_fpn_bridgeconv = _Conv1x1(d[3], self.trans_out_dim, 1) if d[3] != self.trans_out_dim else nn.Identity()
        self.out_gn2b = nn.GroupNorm(self.G, d[2])
        self.out_gn3b = nn.GroupNorm(self.G, d[3])
        self.out_gn4b = nn.GroupNorm(self.G, d[4])               # unused (N3)
        self.out_conv = _Conv1x1(self.trans_out_dim, self.num_classes, 1)
        self.out_fpn_dropout = nn.Dropout(config.hidden_dropout_prob)
        self.apply(self.init_weights)
        self.apply(self.tie_qk)
        self.apply(self.add_identity_bias)
        self.translayer_dims = config.translayer_dims
        self.num_vis_layers = 1 + 2 * self.num_translayers
        self.feature_maps, self.layers_attn_scores, self.orig_feat_shape = [], None, None
        self.keep_feature_maps = False       # the reference always stores them (visualisation); opt-in here

    def get_mask(self, batch):
        ks = self.mask_pool.kernel_size
        return SF.nonzero_mask(batch, ks if isinstance(ks, tuple) else (ks, ks))       # 0/1 floats [B, H/8, W/8]

    def in_fpn_forward(self, feats, nonzero_mask, B):
        f3, f4 = feats[3], feats[4]
        cur = _up(f4, f3.shape[2:], base=self.in_fpn34_conv(f3))                                   # 'AN': add, then normalise
        cur = SF.bn_act(cur, self.in_bn4b) if self.in_fpn_use_bn else SF.group_norm(cur, self.in_gn4b)
        cur = self.in_fpn_bridgeconv(cur)
        H2, W2 = cur.shape[2:]
        vfeat = SF.transpose12(cur.reshape(B, self.trans_in_dim, H2 * W2))             # NCHW map -> channels-last tokens [B, N, C] (LDS-tiled, both ways)
        return vfeat, nonzero_mask.reshape(B, -1), H2, W2

    def out_fpn_forward(self, feats, vfeat_fused, B0):
        cur = SF.group_norm(_up(feats[2], feats[1].shape[2:], base=self.out_fpn12_conv(feats[1])), self.out_gn2b)
        cur = SF.group_norm(_up(feats[3], cur.shape[2:], base=self.out_fpn23_conv(cur)), self.out_gn3b)
        out = _up(vfeat_fused, cur.shape[2:], base=self.out_fpn_bridgeconv(cur))
        if self.out_fpn_do_dropout:
            out = SF.dropout(out, self.out_fpn_dropout.p, self.training)
        return out

    def out_head_forward(self, feats, fused_tokens, grid_shape, size):
        """out_conv(out_fpn_bridgeconv(cur) + up(vfeat_fused)) re-associated as  (W_out W_bridge) cur + up(W_out vfeat_fused):
        pointwise convolutions compose, and a pointwise convolution commutes with bilinear resampling (the blend weights sum
        to 1, so the bias commutes too).  Same function and same gradients for every parameter (chain rule through the composed
        weight), fp32 rounding aside; the trans_out_dim-channel map at the out-FPN resolution is never formed (reference op
        order, :304-306 then :427-436: `fuse_output_tail = False`)."""
        # r05: the two GroupNorms are folded into the pointwise convolutions that consume them (SF.up_group_norm_conv, as in Segtran3d.out_head_forward): a level
        # is written by the resampling pass that also leaves its statistics and read by the convolution; sizes the fused pass does not serve take the plain ops
        size1 = feats[1].shape[2:]
        base3 = SF.up_group_norm_conv(feats[2], size1, self.out_fpn12_conv(feats[1]), self.out_gn2b, self.out_fpn23_conv.weight, self.out_fpn23_conv.bias)
        wo, bo = self.out_conv.weight, self.out_conv.bias
        if isinstance(self.out_fpn_bridgeconv, nn.Identity):
            wl, bl = wo, bo
        else:
            wl, bl = SF.compose_conv1x1(wo, bo, self.out_fpn_bridgeconv.weight, self.out_fpn_bridgeconv.bias)
        lateral = SF.up_group_norm_conv(feats[3], size1, base3, self.out_gn3b, wl, bl)
        scores = _up(SF.conv1x1_tokens(fused_tokens, grid_shape, wo), size1, base=lateral)
        return _up(scores, size)

    def forward(self, batch):
        SF.defer_bn_ticks()
        try:
            return self._forward(batch)
        finally:
            SF.flush_bn_ticks()               # one multi-tensor `num_batches_tracked += 1` for all BatchNorm layers of this pass

    def _forward(self, batch):
        self.feature_maps = []
        B, C, H, W = batch.shape
        if H % 8 or W % 8:
            raise ValueError('Segtran2d needs H and W divisible by 8 (reference segtran2d.py:379), got %dx%d' % (H, W))
        nonzero_mask = self.get_mask(batch)
        ep = self.backbone.extract_endpoints(batch)
        feats = tuple(ep['reduction_%d' % i] for i in range(1, 6))
        vfeat, vmask, H2, W2 = self.in_fpn_forward(feats, nonzero_mask, B)
        xy_shape = torch.Size((H2, W2))
        voxels_pos = gen_scaled_positions(xy_shape, (H // H2, W // W2), batch.device)            # [N, 2], batch-invariant
        fused = self.voxel_fusion(vfeat, voxels_pos, vmask, xy_shape)
        self.layers_attn_scores = self.voxel_fusion.layers_attn_scores
        self.orig_feat_shape = xy_shape
        if self.keep_feature_maps:
            self.feature_maps = [vfeat.transpose(1, 2).view(B, -1, H2, W2)] + \
                [lv.view(B, H2, W2, -1).permute(0, 3, 1, 2) for lv in self.voxel_fusion.layers_vfeat]
        if self.fuse_output_tail and not self.out_fpn_do_dropout:
            return self.out_head_forward(feats, fused, (H2, W2), (H, W))
        fused = fused.view(B, H2, W2, self.trans_out_dim).permute(0, 3, 1, 2)
        out = self.out_fpn_forward(feats, fused, B)
        return _up(self.out_conv(out), (H, W))
