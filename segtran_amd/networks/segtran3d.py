"""Segtran3d -- host-side mirror of /root/reference/code/networks/segtran3d.py (`--net segtran`, 3D / BraTS).

Same constructor / forward signature ([B,C,H,W,D] -> [B,num_classes,H,W,D]) / attribute surface /
state_dict keys.  Built: I3D backbone, 'bridgeconv' 4->3 input bridge, in_fpn '34' / out_fpn '1234' ('AN',
GroupNorm), D_pool_K depth pooling with 'interp' un-pooling -- i.e. the configuration train3d.py forces.
The reference's hard-coded device='cuda' (segtran3d.py:464, N8) is replaced by the input's device.

Kernel status: everything runs on libsegx -- fusion encoder, every 1x1x1 convolution, the 3^3 / 7^3 convolutions (implicit GEMM, forward /
backward-data / backward-weight), 'same' max pools, GroupNorm, the trilinear resampling (with the fused lateral add), BatchNorm3d+ReLU.
Two exact re-associations of consecutive linear maps are on by default (switches `fuse_output_tail`, `fuse_input_bridge`; both orders are
parity-tested): the class projection composed into the out-FPN bridge, and the 4 -> 3 input bridge composed into the stem filters.
"""
import os as _os
import torch
import torch.nn as nn
import torch.nn.functional as F
from argparse import Namespace

from .. import functional as SF
from .aj_i3d.aj_i3d import InceptionI3d
from .segtran_shared import (SegtranConfig, bb2feat_dims, SegtranFusionEncoder, CrossAttFeatTrans,  # noqa: F401
                             ExpandedFeatTrans, SegtranInitWeights, gen_all_indices, gen_scaled_positions)


class Segtran3dConfig(SegtranConfig):
    def __init__(self):
        super().__init__()
        self.backbone_type = 'i3d'
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
        self.pos_dim = 3
        self.max_pos_size = (20, 20, 20)
        self.input_scale = (1., 1., 1.)
        self.num_classes = 2
        self.num_attractors = 1024
        self.orig_in_channels = 1
        self.inchan_to3_scheme = 'bridgeconv'
        self.D_groupsize = 1
        self.D_pool_K = 2
        self.out_fpn_upsampleD_scheme = 'interp'
        self.device = 'cuda'

    def update_config(self, args):
        self.try_assign(args, 'num_classes', 'backbone_type', 'use_pretrained', 'bb_feat_upsize', 'in_fpn_use_bn',
                        'use_squeezed_transformer', 'num_attractors', 'num_translayers', 'num_modes',
                        'trans_output_type', 'mid_type', 'pos_code_type', 'pos_code_weight', 'pos_bias_radius',
                        'ablate_multihead', 'out_fpn_do_dropout', 'has_FFN_in_squeeze', 'attn_clip', 'qk_have_bias',
                        'tie_qk_scheme', 'orig_in_channels', 'inchan_to3_scheme', 'D_groupsize', 'D_pool_K',
                        'out_fpn_upsampleD_scheme', 'input_scale', 'device', 'eval_robustness', 'use_attn_consist_loss',
                        'use_mince_transformer', 'mince_scales', 'mince_channel_props')
        a = args if isinstance(args, dict) else args.__dict__
        if 'dropout_prob' in a and a['dropout_prob'] >= 0:
            self.hidden_dropout_prob = a['dropout_prob']
            self.attention_probs_dropout_prob = a['dropout_prob']
        self.bb_feat_dims = bb2feat_dims[self.backbone_type]
        self.set_fpn_layers('args', args, do_print=False)


CONFIG = Segtran3dConfig()


class _Conv1x1x1(nn.Conv3d):
    def forward(self, x):
        return SF.conv1x1(x, self.weight, self.bias)


def _up(x, size, base=None):
    """trilinear resize (align_corners=False) with the FPN lateral `base` added in the same libsegx pass."""
    return SF.interp_linear(x, size, base)


class Segtran3d(SegtranInitWeights):
    def __init__(self, config):
        super().__init__(config)
        self.config = config
        self.device = config.device
        self.orig_in_channels = config.orig_in_channels
        self.trans_in_dim, self.trans_out_dim = config.trans_in_dim, config.trans_out_dim
        self.num_translayers = config.num_translayers
        self.bb_feat_upsize = config.bb_feat_upsize
        self.G = config.G
        self.voxel_fusion = SegtranFusionEncoder(config, 'Fusion')
        self.backbone_type = config.backbone_type
        if not self.backbone_type.startswith('i3d'):
            raise NotImplementedError('Only support i3d as the 3D backbone')
        self.backbone = InceptionI3d(do_pool1=not self.bb_feat_upsize)
        self.use_pretrained = config.use_pretrained
        if self.use_pretrained:                                 # segtran3d.py:99-104: aj_rgb_imagenet.pth beside aj_i3d.py
            import os
            from .aj_i3d import aj_i3d as _aj
            cand = [os.path.join(os.path.dirname(_aj.__file__), 'aj_rgb_imagenet.pth')]
            if os.environ.get('SEGX_PRETRAINED_DIR'):
                cand.append(os.path.join(os.environ['SEGX_PRETRAINED_DIR'], 'aj_rgb_imagenet.pth'))
            path = next((c for c in cand if os.path.exists(c)), None)
            if path is None:
                raise RuntimeError('pretrained I3D weights (aj_rgb_imagenet.pth) not found in %s -- the reference ships it as a '
                                   'git-lfs blob; place it there or build with use_pretrained=False' % cand)
            self.backbone.load_state_dict(torch.load(path, map_location=torch.device('cpu')))
        self.inchan_to3_scheme, self.D_groupsize = config.inchan_to3_scheme, config.D_groupsize
        self.eff_in_channels = self.orig_in_channels * self.D_groupsize
        self.D_pool_K = config.D_pool_K
        self.out_fpn_upsampleD_scheme = config.out_fpn_upsampleD_scheme
        self.input_scale = config.input_scale
        if self.D_groupsize != 1 or self.inchan_to3_scheme != 'bridgeconv' or self.out_fpn_upsampleD_scheme != 'interp' \
                or config.out_fpn_use_bn or not self.bb_feat_upsize:
            raise NotImplementedError("only inchan_to3_scheme='bridgeconv', D_groupsize=1, 'interp' depth un-pooling "
                                      "(what train3d.py:180-195 forces) are built")
        self.in_bridge_to3 = _Conv1x1x1(self.eff_in_channels, 3, 1) if self.eff_in_channels != 3 else nn.Identity()
        self.in_fpn_layers, self.in_fpn_scheme = config.in_fpn_layers, config.in_fpn_scheme
        self.out_fpn_layers, self.out_fpn_scheme = config.out_fpn_layers, config.out_fpn_scheme
        if self.in_fpn_layers != [3, 4] or self.out_fpn_layers != [1, 2, 3, 4] or self.in_fpn_scheme != 'AN' \
                or self.out_fpn_scheme != 'AN':
            raise NotImplementedError("only --infpn 34 --outfpn 1234 with the 'AN' scheme (reference defaults) are built")
        self.mask_pool = nn.AvgPool3d((4, 8, 8))
        d = self.bb_feat_dims = config.bb_feat_dims
        self.in_fpn23_conv = _Conv1x1x1(d[2], d[3], 1)           # unused (N3)
        self.in_fpn34_conv = _Conv1x1x1(d[3], d[4], 1)
        self.in_fpn_bridgeconv = _Conv1x1x1(d[4], self.trans_in_dim, 1) if d[4] != self.trans_in_dim else nn.Identity()
        self.in_fpn_use_bn = config.in_fpn_use_bn
        if self.in_fpn_use_bn:                                   # --inbn (segtran3d.py:177-180)
            self.in_bn3b = nn.BatchNorm3d(d[3])
            self.in_bn4b = nn.BatchNorm3d(d[4])
        else:
            self.in_gn3b = nn.GroupNorm(self.G, d[3])
            self.in_gn4b = nn.GroupNorm(self.G, d[4])
        self.out_fpn_do_dropout = config.out_fpn_do_dropout      # --outdrop (:392-394)
        self.fuse_output_tail = True                             # see out_head_forward
        self.fuse_input_bridge = True                            # see _forward: in_bridge_to3 composed into the stem filters
        self.stem_space_to_depth = _os.environ.get('SEGX_STEM_S2D', '1') != '0'      # r06: ... as a stride-(2, 2, 1) convolution over a space-to-depth image (False: r02's 8-channel stride-2 form)
        self.num_classes = config.num_classes
        self.do_out_fpn = True
        self.out_fpn_out_dim = self.out_feat_dim = self.trans_out_dim
        self.out_fpn12_conv3d = _Conv1x1x1(d[1], d[2], 1)
        self.out_fpn23_conv3d = _Conv1x1x1(d[2], d[3], 1)
        self.out_fpn34_conv3d = _Conv1x1x1(d[3], d[4], 1)        # unused (N3)
        self.out_fpn_bridgeconv3d = _Conv1x1x1(d[3], self.trans_out_dim, 1)
        self.out_gn2b = nn.GroupNorm(self.G, d[2])
        self.out_gn3b = nn.GroupNorm(self.G, d[3])
        self.out_gn4b = nn.GroupNorm(self.G, d[4])               # unused (N3)
        self.out_conv3d = _Conv1x1x1(self.out_feat_dim, self.num_classes, 1)
        self.out_fpn_dropout = nn.Dropout(config.hidden_dropout_prob)
        self.apply(self.init_weights)
        self.apply(self.tie_qk)
        self.apply(self.add_identity_bias)
        self.translayer_dims = config.translayer_dims
        self.num_vis_layers = 1 + 2 * self.num_translayers
        self.layers_attn_scores, self.orig_feat_shape = None, None

    def get_mask(self, batch):
        return SF.nonzero_mask(batch, self.mask_pool.kernel_size)                       # 0/1 floats [B, D/4, H/8, W/8]

    def in_fpn_forward(self, feats, nonzero_mask):
        f3, f4 = feats[3], feats[4]
        if self.in_fpn_use_bn:
            cur = SF.bn_act(_up(f4, f3.shape[2:], base=self.in_fpn34_conv(f3)), self.in_bn4b)
        else:
            cur = SF.up_group_norm(f4, f3.shape[2:], self.in_fpn34_conv(f3), self.in_gn4b)
        cur = self.in_fpn_bridgeconv(cur)
        dp = [cur.shape[2] // self.D_pool_K, cur.shape[3], cur.shape[4]]
        cur = _up(cur, dp)                                                        # depth pooling by interpolation (:319)
        m = (_up(nonzero_mask.unsqueeze(1), dp).squeeze(1) >= 0.5)
        B, Fd, D2, H2, W2 = cur.shape
        vfeat = SF.transpose12(cur.reshape(B, Fd, D2 * H2 * W2))                    # NCDHW map -> channels-last tokens [B, N, C] (LDS-tiled, both ways)
        return vfeat, m.reshape(B, -1), D2, H2, W2

    def out_fpn_forward(self, feats, vfeat_fused):
        cur = SF.up_group_norm(feats[2], feats[1].shape[2:], self.out_fpn12_conv3d(feats[1]), self.out_gn2b)
        cur = SF.up_group_norm(feats[3], cur.shape[2:], self.out_fpn23_conv3d(cur), self.out_gn3b)
        out = _up(vfeat_fused, cur.shape[2:], base=self.out_fpn_bridgeconv3d(cur))
        if self.D_pool_K > 1:
            out = _up(out, [out.shape[2] * self.D_pool_K, out.shape[3], out.shape[4]])
        if self.out_fpn_do_dropout:
            out = SF.dropout(out, self.out_fpn_dropout.p, self.training)
        return out

    def out_head_forward(self, feats, fused_tokens, grid_shape, size):
        """out_conv3d(upD(out_fpn_bridgeconv3d(cur) + up(vfeat_fused))) re-associated as
        upD((W_out W_bridge) cur + up(W_out vfeat_fused)): pointwise convolutions compose and commute with trilinear resampling
        (blend weights sum to 1).  Same function, same parameter gradients (chain rule through the composed weight), fp32
        rounding aside -- and the 1024-channel maps at the out-FPN resolution (2 x 2.5 GB at cfg4), the 832 -> 1024 bridge GEMMs
        over 150528 voxels (fwd / bwd-data / bwd-weight) and their resampling passes never exist.  Reference op order
        (:364-367, :381-386, :490): `fuse_output_tail = False`."""
        # r05: both GroupNorms of the pyramid are FOLDED into the pointwise convolutions that consume them (SF.up_group_norm_conv: per-sample weights W * sc_b and
        # biases W sh_b + bias): the 2 - 3.5 GB levels are written once, by the resampling pass that also leaves their statistics, and read once, by the convolution
        size1 = feats[1].shape[2:]
        base3 = SF.up_group_norm_conv(feats[2], size1, self.out_fpn12_conv3d(feats[1]), self.out_gn2b, self.out_fpn23_conv3d.weight, self.out_fpn23_conv3d.bias)
        wo, bo = self.out_conv3d.weight, self.out_conv3d.bias
        if isinstance(self.out_fpn_bridgeconv3d, nn.Identity):
            wl, bl = wo, bo
        else:
            wl, bl = SF.compose_conv1x1(wo, bo, self.out_fpn_bridgeconv3d.weight, self.out_fpn_bridgeconv3d.bias)
        lateral = SF.up_group_norm_conv(feats[3], size1, base3, self.out_gn3b, wl, bl)
        scores = _up(SF.conv1x1_tokens(fused_tokens, grid_shape, wo), size1, base=lateral)
        if self.D_pool_K > 1:
            scores = _up(scores, [scores.shape[2] * self.D_pool_K, scores.shape[3], scores.shape[4]])
        return _up(scores.permute(0, 1, 3, 4, 2), size)

    def forward(self, batch):
        SF.defer_bn_ticks()
        try:
            return self._forward(batch)
        finally:
            SF.flush_bn_ticks()               # one multi-tensor `num_batches_tracked += 1` for all BatchNorm layers of this pass

    def _forward(self, batch):
        B, C, H, W, D = batch.shape
        assert C == self.orig_in_channels
        if H % 8 or W % 8 or D % 8:
            raise ValueError('Segtran3d needs H, W, D divisible by 8 (reference segtran3d.py:450), got %s' % ((H, W, D),))
        stem = self.backbone.Conv3d_1a_7x7
        if self.fuse_input_bridge and not isinstance(self.in_bridge_to3, nn.Identity) and stem.conv3d.bias is None and self.eff_in_channels < 8:
            # in_bridge_to3 (a pointwise linear map with bias, :420) feeds the 7x7x7 stem directly: the two are composed into ONE 8-channel
            # convolution ([x, 1, 0, 0, 0] -> 64; the bias rides on the constant-one channel, which the zero padding switches off outside the
            # volume exactly like the padded bridge output).  Same function and, through SF.stem_compose's chain rule, the same gradient for
            # every parameter; what disappears is the stride-2 transposed convolution onto the 3-channel image (5.4 ms of the cfg4 step) and
            # the bridge's own backward GEMMs.  The foreground mask still needs the bridged image itself (:425) -- forward only.
            with torch.no_grad():      # r05: straight from the raw batch (SF.bridge_mask): no K = 4 GEMM, no permuting copy of the bridged image
                nonzero_mask = SF.bridge_mask(batch, self.in_bridge_to3.weight, self.in_bridge_to3.bias, self.mask_pool.kernel_size)
            if self.stem_space_to_depth and 2 * C == 8 and stem._stride == (2, 2, 2) and stem._kernel_shape == (7, 7, 7):
                # r06: the same composition WITHOUT its three all-zero channels and its constant channel, on a space-to-depth image along W (SF.stem_bridge_conv_s2d):
                # -43 % of the stem's FLOPs, unit stride along W (its weight gradient moves from the fp32 engine onto the bf16x6 engine)
                conv_out = SF.stem_bridge_conv_s2d(batch, stem.conv3d.weight, self.in_bridge_to3.weight, self.in_bridge_to3.bias, stem._stride)
            else:
                wc = SF.stem_compose(stem.conv3d.weight, self.in_bridge_to3.weight, self.in_bridge_to3.bias, 8)
                conv_out = SF.conv3d_same(SF.bridge_input(batch, 8), wc, stem._stride)
            fd = self.backbone.extract_features(None, stem_conv_out=conv_out)
        else:
            rgb = self.in_bridge_to3(batch).permute(0, 1, 4, 2, 3)                # [B,3,D,H,W]  (reference op order: fuse_input_bridge = False)
            nonzero_mask = self.get_mask(rgb)
            fd = self.backbone.extract_features(rgb)
        feats = (fd['MaxPool3d_2a_3x3'], fd['Conv3d_2c_3x3'], fd['Mixed_3c'], fd['Mixed_4f'], fd['Mixed_5c'])
        vfeat, vmask, D2, H2, W2 = self.in_fpn_forward(feats, nonzero_mask)
        xyz_shape = torch.Size((D2, H2, W2))
        voxels_pos = gen_scaled_positions(xyz_shape, ((D // D2) / self.input_scale[2], (H // H2) / self.input_scale[0],
                                                     (W // W2) / self.input_scale[1]), batch.device)
        fused = self.voxel_fusion(vfeat, voxels_pos, vmask, xyz_shape)
        self.layers_attn_scores = self.voxel_fusion.layers_attn_scores
        self.orig_feat_shape = xyz_shape
        if self.fuse_output_tail and not self.out_fpn_do_dropout:
            return self.out_head_forward(feats, fused, (D2, H2, W2), (H, W, D))
        fused = fused.view(B, D2, H2, W2, self.trans_out_dim).permute(0, 4, 1, 2, 3)
        out = self.out_fpn_forward(feats, fused)                                  # [B, F, D, H, W]
        # the class projection is pointwise, so it commutes exactly with the (H,W,D) permutation of :488-490;
        # doing it first permutes 4 channels instead of 1024.
        scores_small = self.out_conv3d(out).permute(0, 1, 3, 4, 2)
        return _up(scores_small, (H, W, D))
