"""Polyformer layer (mirror of reference code/networks/polyformer.py:8-103): a squeeze-and-expansion attention pair without FFN
(`has_FFN=False`, 4 modes) applied to a 2x-pooled feature map and added back to it, for insertion into a segmentation backbone
(few-shot domain adaptation).  All arithmetic on libsegx: avg_pool2, batched transpose, the CrossAttFeatTrans kernels (with the
LayerNorm-free mode aggregate of the no-FFN branch), bilinear resampling with the residual add fused."""
import torch
import torch.nn as nn
from torch.nn import Parameter

from .. import functional as SF
from .segtran_shared import CrossAttFeatTrans, SegtranInitWeights, SegtranConfig


class PolyformerLayer(SegtranInitWeights):
    def __init__(self, name, config):
        super().__init__(config)
        self.name = name
        self.chan_axis = config.chan_axis
        if self.chan_axis != 1:
            raise NotImplementedError('chan_axis=1 (NCHW feature maps) is the built layout')
        self.feat_dim, self.num_attractors, self.qk_have_bias = config.feat_dim, config.num_attractors, config.qk_have_bias
        self.in_ator_trans = CrossAttFeatTrans(config, name + '-in-squeeze')
        self.ator_out_trans = CrossAttFeatTrans(config, name + '-squeeze-out')
        self.attractors = Parameter(torch.randn(1, self.num_attractors, self.feat_dim))
        self.infeat_norm_layer = nn.LayerNorm(self.feat_dim, eps=1e-12, elementwise_affine=False)
        self.poly_do_layernorm = config.poly_do_layernorm
        self.apply(self.init_weights)
        self.apply(self.tie_qk)
        self.apply(self.add_identity_bias)

    def forward(self, in_feat):
        B, C, H, W = in_feat.shape
        half0 = SF.avg_pool2(in_feat)                                        # :40  [B, C, h, w]
        h2, w2 = half0.shape[2:]
        # :41,46  `in_feat_half0.transpose(chan_axis, -1).reshape(B, -1, C)` swaps dims 1 and 3: the tokens run W-MAJOR, [B, w*h, C]
        wmajor = SF.transpose12(half0.reshape(B * C, h2, w2)).reshape(B, C, w2 * h2)
        vfeat = SF.transpose12(wmajor)                                       # [B, w*h, C]
        if self.poly_do_layernorm:
            vfeat = SF.layer_norm(vfeat)                                     # :44-45 (no affine)
        new_att = self.in_ator_trans(self.attractors, vfeat)                 # attractors shared by the batch (:48-49)
        vout = self.ator_out_trans(vfeat, new_att)                           # :50
        # :51-52  `vfeat_out.transpose(chan_axis, -1).reshape(in_feat_half0.shape)`: the w-major token index is read back as (row, col)
        # of the [h, w] map -- for the square maps it is used on, the attention output is added back spatially TRANSPOSED (quirk N10)
        out_half = SF.transpose12(vout).reshape(B, C, h2, w2)
        return SF.interp_linear(out_half, (H, W), base=in_feat)              # :53-55  upsample + residual in one pass


class Polyformer(nn.Module):
    def __init__(self, feat_dim, chan_axis=1, args=None):
        config = SegtranConfig()
        if args is None:
            config.num_attractors, config.num_modes, config.tie_qk_scheme = 256, 4, 'loose'
            config.qk_have_bias, config.pos_code_type = True, 'lsinu'
        else:
            config.num_attractors = args.num_attractors
            config.num_modes = args.num_modes if args.num_modes != -1 else 4
            config.tie_qk_scheme, config.qk_have_bias, config.pos_code_type = args.tie_qk_scheme, args.qk_have_bias, args.pos_code_type
        config.num_layers = 1
        config.in_feat_dim = config.feat_dim = config.min_feat_dim = feat_dim
        config.v_has_bias = False
        config.has_FFN = False
        config.ablate_multihead = False
        config.chan_axis = chan_axis
        config.poly_do_layernorm = False
        super().__init__()
        self.polyformer_layers = nn.Sequential(*[PolyformerLayer(str(i), config) for i in range(config.num_layers)])

    def forward(self, in_feat):
        return self.polyformer_layers(in_feat)
