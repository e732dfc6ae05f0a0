"""Squeeze-and-Expansion transformer on libsegx -- host-side mirror of the reference plugin surface.

Mirrors, name for name, what callers of the reference touch in `code/networks/segtran_shared.py`
(SegtranConfig :90-196, ExpandedFeatTrans :329-476, CrossAttFeatTrans :478-610, SqueezedAttFeatTrans
:787-816, SegtranFusionEncoder :819-975, LearnedSinuPosEmbedder :979-998, SegtranPosEncoder :1177-1238,
SegtranInitWeights :1241-1264) and keeps the same parameter names/shapes, so reference checkpoints load
unchanged.  The arithmetic is NOT ATen: every forward/backward step is a hand-written HIP kernel of
libsegx reached through segtran_amd.functional.

Layout choices that differ from the reference on purpose (internal, invisible in state_dict / outputs):
  * per-mode tensors are mode-major [M, B, U, F] and never bounced through [B, M*F, U] (:416-419, :449);
  * the attractor query projection is computed once per step, not per sample (:812 expands to B);
  * the positional code is computed once per forward as [N, C] (batch-invariant, :1231 + segtran2d.py:392)
    and sliced per layer, instead of being regenerated as [B, N, C] for every layer;
  * attention diagnostics (:569-587) stay on the device: no `.item()` host syncs in the hot loop.
Quirks reproduced on purpose: N1 dropped residual, N2 tied q/k, N3 unused parameters, N4 eps=1e-12,
N5 conditional clip, N11 untied q/k under --mince.  Built besides the squeezed default: the non-squeezed encoder with sliding positional
biases (`--nosqueeze --pos bias`) and the Mince multi-scale transformer (`--mince`).  Not built (no BASELINE config uses them; they raise):
multi-head ablation, 'rand' / 'sinu' positional embedders, attention-consistency loss.
"""
import copy
import math
import numpy as np
import torch
import torch.nn as nn
from torch.nn import Parameter

from .. import functional as SF
from ..functional import GemmSpec

bb2feat_dims = {'resnet34': [64, 64, 128, 256, 512], 'resnet50': [64, 256, 512, 1024, 2048],
                'resnet101': [64, 256, 512, 1024, 2048], 'resibn101': [64, 256, 512, 1024, 2048],
                'eff-b0': [16, 24, 40, 112, 1280], 'eff-b1': [16, 24, 40, 112, 1280],
                'eff-b2': [16, 24, 48, 120, 1408], 'eff-b3': [24, 32, 48, 136, 1536],
                'eff-b4': [24, 32, 56, 160, 1792], 'effv2m': [24, 48, 80, 176, 512],
                'i3d': [64, 192, 480, 832, 1024]}


def gen_all_indices(shape, device):
    """Coordinates of every cell of a grid, [*shape, len(shape)] (reference :28-36)."""
    grids = torch.meshgrid(*[torch.arange(s, device=device) for s in shape], indexing='ij')
    return torch.stack(grids, dim=len(shape))


def gen_scaled_positions(shape, scales, device):
    """gen_all_indices(shape).view(-1, nd).float() * scales (segtran2d.py:375-389, segtran3d.py:443-468), built from device-side aranges and
    python scalars only: no host-to-device tensor upload, so the forward pass can be captured into a hipGraph.  Same fp32 products."""
    axes = [torch.arange(int(n), device=device, dtype=torch.float32) * float(sc) for n, sc in zip(shape, scales)]
    return torch.stack(torch.meshgrid(*axes, indexing='ij'), dim=len(axes)).view(-1, len(axes))


class SegtranConfig:
    """Application-independent settings; same field names and defaults as the reference (:90-196)."""

    def __init__(self):
        self.feat_dim = -1
        self.in_feat_dim = -1
        self.num_modes = 4
        self.use_squeezed_transformer = True
        self.num_attractors = 256
        self.tie_qk_scheme = 'shared'
        self.mid_type = 'shared'
        self.trans_output_type = 'private'
        self.act_fun = torch.nn.functional.gelu
        self.has_FFN = True
        self.has_FFN_in_squeeze = False
        self.pos_code_type = 'lsinu'
        self.pos_code_weight = 1.
        self.pos_bias_radius = 7
        self.qk_have_bias = True
        self.v_has_bias = False
        self.attn_clip = 500
        self.base_initializer_range = 0.02
        self.query_idbias_scale = 10
        self.feattrans_lin1_idbias_scale = 10
        self.pool_modes_feat = 'softmax'
        self.use_mince_transformer = False
        self.mince_scales = None
        self.mince_channel_props = None
        self.hidden_dropout_prob = 0.1
        self.attention_probs_dropout_prob = 0.1
        self.out_fpn_do_dropout = False
        self.eval_robustness = False
        self.ablate_multihead = False
        self.use_attn_consist_loss = False

    def try_assign(self, args, *keys):
        ok = False
        src = args if isinstance(args, dict) else args.__dict__
        for key in keys:
            if key in src:
                self.__dict__[key] = src[key]
                ok = True
        return ok

    def set_fpn_layers(self, config_name, fpn_settings, do_print=True):
        self.in_fpn_layers = [int(c) for c in fpn_settings.in_fpn_layers]
        self.out_fpn_layers = [int(c) for c in fpn_settings.out_fpn_layers]
        if self.out_fpn_layers[-1] > self.in_fpn_layers[-1]:
            raise ValueError("in_fpn_layers=%s is not compatible with out_fpn_layers=%s"
                             % (self.in_fpn_layers, self.out_fpn_layers))
        self.orig_in_feat_dim = self.bb_feat_dims[self.in_fpn_layers[-1]]
        ratios = fpn_settings.translayer_compress_ratios
        self.translayer_compress_ratios = ratios
        assert len(ratios) == self.num_translayers + 1, \
            "Length of {} != 1 + num_translayers {}".format(ratios, self.num_translayers)
        abs_ratios = np.cumprod(ratios)                       # adjacent ratios -> absolute (:177-183)
        self.translayer_dims = [int(self.orig_in_feat_dim / r) for r in abs_ratios]
        self.trans_in_dim = self.translayer_dims[0]
        self.min_feat_dim = np.min(self.translayer_dims)
        self.trans_out_dim = self.translayer_dims[-1]
        self.in_fpn_scheme = fpn_settings.in_fpn_scheme
        self.out_fpn_scheme = fpn_settings.out_fpn_scheme
        if do_print:
            print("'%s' orig in-feat: %d, in-feat: %d, out-feat: %d, in-scheme: %s, out-scheme: %s, translayer_dims: %s"
                  % (config_name, self.orig_in_feat_dim, self.trans_in_dim, self.trans_out_dim,
                     self.in_fpn_scheme, self.out_fpn_scheme, self.translayer_dims))


def multi_resize_shape(shape, scales):
    """reference :38-43."""
    return [tuple(int(v / scale) for v in shape) for scale in scales]


def fracs_to_indices(feat_dim, mince_channel_props):
    """reference :68-87: channel boundaries of the mince scales; the last scale takes the remainder."""
    fracs = np.array(mince_channel_props, dtype=float)
    fracs /= fracs.sum()
    n = len(fracs)
    indices, nums = [0] * (n + 1), [0] * n
    for i in range(n - 1):
        nums[i] = int(fracs[i] * feat_dim)
        indices[i + 1] = nums[i] + indices[i]
    indices[-1] = feat_dim
    nums[-1] = indices[-1] - indices[-2]
    return indices, nums


def _unsupported(config):
    if config.ablate_multihead or config.eval_robustness:
        raise NotImplementedError('multi-head ablation / robustness variants are outside the MI355X hot path (SURVEY.md 8(f))')
    if config.mid_type != 'shared' or config.trans_output_type != 'private' or config.pool_modes_feat != 'softmax':
        raise NotImplementedError("only mid_type='shared', trans_output_type='private', pool_modes_feat='softmax' "
                                  "(the values train2d.py:245-249 forces) are built")


# ---- parameter containers with the reference's names (forward passes live in the owners) -------------
class MMSharedMid(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_modes, self.feat_dim = config.num_modes, config.feat_dim
        self.shared_linear = nn.Linear(self.feat_dim, self.feat_dim)


class MMPrivateOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_modes, self.feat_dim = config.num_modes, config.feat_dim
        fa = self.feat_dim * self.num_modes
        self.group_linear = nn.Conv1d(fa, fa, 1, groups=self.num_modes)
        self.resout_norm_layer = nn.LayerNorm(self.feat_dim, eps=1e-12, elementwise_affine=True)


class LearnedSoftAggregate(nn.Module):
    def __init__(self, num_feat, group_dim, keepdim=False):
        super().__init__()
        self.group_dim, self.keepdim = group_dim, keepdim
        self.feat2score = nn.Linear(num_feat, 1)


class ExpandedFeatTrans(nn.Module):
    """Value projection + expansion FFN + mode aggregation (reference :329-476)."""

    def __init__(self, config, name):
        super().__init__()
        _unsupported(config)
        self.config, self.name = config, name
        self.in_feat_dim, self.feat_dim, self.num_modes = config.in_feat_dim, config.feat_dim, config.num_modes
        self.feat_dim_allmode = self.feat_dim * self.num_modes
        self.has_FFN = config.has_FFN
        self.hidden_dropout_prob = config.hidden_dropout_prob
        self.first_linear = nn.Linear(self.in_feat_dim, self.feat_dim_allmode, bias=config.v_has_bias)
        self.first_norm_layer = nn.LayerNorm(self.feat_dim, eps=1e-12, elementwise_affine=True)
        self.base_initializer_range = config.base_initializer_range
        self.feat_softaggr = LearnedSoftAggregate(self.feat_dim, group_dim=1, keepdim=False)
        self.intermediate = MMSharedMid(config)
        self.output = MMPrivateOutput(config)
        if not config.use_mince_transformer or config.mince_scales is None:                # :344-354
            self.num_scales, self.mince_scales = 0, None
        else:
            self.mince_scales, self.num_scales = config.mince_scales, len(config.mince_scales)
            self.mince_channel_props = config.mince_channel_props
            self.mince_channel_indices, _ = fracs_to_indices(self.feat_dim, self.mince_channel_props)

    def add_identity_bias(self):
        if self.config.feattrans_lin1_idbias_scale > 0:
            F_ = self.feat_dim
            eye = torch.diag(torch.ones(F_)) * self.base_initializer_range * self.config.feattrans_lin1_idbias_scale
            w = self.first_linear.weight.data
            w[:F_, :F_] = w[:F_, :F_] * 0.5 + eye.to(w.device)

    def _fuse_mince(self, v, probs, geoshape):
        """Mince branch (:421-443): every scale fuses ITS slice of each mode's value channels on ITS down-sampled token grid with
        ITS attention, the result is resized back to the full grid and the slices are concatenated.  v [B, U, M*F] -> [M, B, U, F]."""
        B, U, _ = v.shape
        M, Fd = self.num_modes, self.feat_dim
        v4 = v.view(B, U, M, Fd)
        shapes = multi_resize_shape(geoshape, self.mince_scales)
        parts = []
        for s_, scale in enumerate(self.mince_scales):
            Lc, Rc = self.mince_channel_indices[s_], self.mince_channel_indices[s_ + 1]
            fs = Rc - Lc
            vs = SF.interp_tokens(v4[..., Lc:Rc].reshape(B, U, M * fs), geoshape, scale_factor=1. / scale)      # :431
            Us = vs.shape[1]
            fused = SF.bgemm(probs[s_], vs,                                                                    # :436
                             GemmSpec(Us, fs, Us, (Us * Us, B * Us * Us, Us, 1), (Us * M * fs, fs, 1, M * fs),
                                      (Us * M * fs, fs, M * fs), (B, Us, M * fs), nb=(B, M)))
            parts.append(SF.interp_tokens(fused, shapes[s_], out_shape=geoshape).view(B, U, M, fs))            # :439
        return torch.cat(parts, dim=-1).permute(2, 0, 1, 3).contiguous()                                       # :443

    def forward(self, input_feat, attention_probs, in_geoshape=None, value_last=False, modes_interleaved=False):
        """input_feat [B, U2, IF]; attention_probs MODE-MAJOR [M, B, U1, U2] (mince: a list, one per scale; modes_interleaved: [B, U1, M, U2]) -> [B, U1, F].
        value_last (one mode, bias-free value projection): fuse first, project after -- (P X) Wv^T instead of P (X Wv^T); the
        projection then runs over the U1 fused rows instead of the U2 input tokens (see CrossAttFeatTrans.forward)."""
        B, U2, IF = input_feat.shape
        M, Fd = self.num_modes, self.feat_dim
        drop = self.hidden_dropout_prob if self.training else 0.0
        if value_last:
            assert M == 1 and self.first_linear.bias is None and self.num_scales == 0
            U1 = attention_probs.shape[2]
            px = SF.bgemm(attention_probs, input_feat,
                          GemmSpec(U1, IF, U2, (U1 * U2, B * U1 * U2, U2, 1), (U2 * IF, 0, 1, IF), (U1 * IF, 0, IF), (B, U1, IF), nb=(B, 1)))
            fused = SF.linear(px, self.first_linear.weight).view(1, B, U1, Fd)
            v = None
        else:
            v = SF.linear(input_feat, self.first_linear.weight, self.first_linear.bias)   # [B, U2, M*F]
        if value_last:
            pass
        elif self.num_scales > 0:
            U1 = U2
            fused = self._fuse_mince(v, attention_probs, tuple(int(g) for g in in_geoshape))
        else:
            U1 = attention_probs.shape[1] if modes_interleaved else attention_probs.shape[2]
            p_strides = (U1 * M * U2, U2, M * U2, 1) if modes_interleaved else (U1 * U2, B * U1 * U2, U2, 1)
            fuse_spec = GemmSpec(U1, Fd, U2, p_strides, (U2 * M * Fd, Fd, 1, M * Fd),
                                 (U1 * Fd, B * U1 * Fd, Fd), (M, B, U1, Fd), nb=(B, M))
            if self.has_FFN and U2 < U1 and CrossAttFeatTrans.reassociate_projections and B * U1 >= CrossAttFeatTrans.reassociate_min_rows:
                # Squeeze-out layer: U1 tokens gather from U2 << U1 attractors, then MMSharedMid (:232-251) applies one [F, F]
                # linear map to every fused row:  (P v) Wmid^T + b  ==  P (v Wmid^T) + b.  The map is applied to the U2 value
                # rows instead (M*B*U2 rows instead of M*B*U1: 39 vs 632 GFLOP at cfg2), and the fusion GEMM carries the bias +
                # GELU + dropout epilogue.  The fused values themselves are needed by nothing else (N1: the residual is dropped).
                mid, out = self.intermediate.shared_linear, self.output
                u = SF.linear(v.view(B, U2, M, Fd), mid.weight)                           # [B, U2, M, F], bias added after fusion
                fuse_spec.bias_mode = SF.BIAS_N
                h = SF.bgemm(attention_probs, u, fuse_spec, bias=mid.bias, gelu=True, drop_p=drop)
                return self._ffn_tail(h, B, U1, drop)
            # fused[m,b] = probs[m,b] @ v[b, :, m*F:(m+1)*F]   (no [B,M*F,U] transposes)
            fused = SF.bgemm(attention_probs, v, fuse_spec)
        agg = self.feat_softaggr.feat2score
        if not self.has_FFN:
            # LearnedSoftAggregate over M modes, then first_norm_layer (:452-457).
            if M == 1:
                # squeeze path: the aggregate is the identity, its parameters get exact-zero gradients (N3); LN + aggregate fused
                y = SF.modes_aggr(fused.view(1, B * U1, Fd), self.first_norm_layer.weight, self.first_norm_layer.bias,
                                  agg.weight, agg.bias, 0.0)
                return y.view(B, U1, Fd)
            # Polyformer: aggregate the raw mode features (scores from the un-normalised features), then LayerNorm
            y = SF.modes_aggr(fused.view(M, B * U1, Fd), None, None, agg.weight, agg.bias, 0.0)
            y = SF.layer_norm(y, self.first_norm_layer.weight, self.first_norm_layer.bias)
            return y.view(B, U1, Fd)
        mid = self.intermediate.shared_linear
        h = SF.linear(fused, mid.weight, mid.bias, gelu=True, drop_p=drop)               # MMSharedMid :232-251
        return self._ffn_tail(h, B, U1, drop)

    def _ffn_tail(self, h, B, U1, drop):
        """MMPrivateOutput (per-mode linear, dropout, LayerNorm) + LearnedSoftAggregate on h [M, B, U1, F]."""
        M, Fd = self.num_modes, self.feat_dim
        out, agg = self.output, self.feat_softaggr.feat2score
        R = B * U1
        z = SF.bgemm(h, out.group_linear.weight,                                          # MMPrivateOutput :267
                     GemmSpec(R, Fd, Fd, (0, R * Fd, Fd, 1), (0, Fd * Fd, Fd, 1), (0, R * Fd, Fd), (M, R, Fd),
                              nb=(1, M), bias_mode=SF.BIAS_N, bias_b1=Fd),
                     bias=out.group_linear.bias)
        # N1: the reference adds the shortcut (:269) but normalises the un-added tensor (:272) -> no residual.
        y = SF.modes_aggr(z, out.resout_norm_layer.weight, out.resout_norm_layer.bias, agg.weight, agg.bias, drop)
        return y.view(B, U1, Fd)


class CrossAttFeatTrans(nn.Module):
    """Multi-mode cross attention (reference :478-610)."""
    reassociate_projections = True        # False: the reference's op order everywhere (key/value projections over all tokens)
    interleave_modes = True               # squeeze-out (re-associated): scores / probs as [B, U1, M, U2] (False: mode-major [M, B, U1, U2], rounds 1-5; tools/ab_switch.py)
    reassociate_min_rows = 4096           # below this many token rows (batch x tokens) the step is launch-bound: keep the op order

    def __init__(self, config, name):
        super().__init__()
        _unsupported(config)
        self.config, self.name = config, name
        self.num_modes, self.in_feat_dim, self.feat_dim = config.num_modes, config.in_feat_dim, config.feat_dim
        self.attention_mode_dim = self.in_feat_dim // self.num_modes
        self.att_size_allmode = self.num_modes * self.attention_mode_dim
        self.query = nn.Linear(self.in_feat_dim, self.att_size_allmode, bias=config.qk_have_bias)
        self.key = nn.Linear(self.in_feat_dim, self.att_size_allmode, bias=config.qk_have_bias)
        self.base_initializer_range = config.base_initializer_range
        self.out_trans = ExpandedFeatTrans(config, name)
        self.attention_probs_dropout_prob = config.attention_probs_dropout_prob
        self.keep_attn_scores = config.use_attn_consist_loss
        self.tie_qk_scheme = config.tie_qk_scheme
        self.attn_clip = float(config.attn_clip)
        self.pos_code_weight = config.pos_code_weight if config.pos_code_type == 'bias' else 1      # :493-497
        self.attention_scores = None
        self.attn_max_dev = None          # device-side running max of the (positive) scores of the last call

    def tie_qk(self, tie_qk_scheme=None):
        if tie_qk_scheme is not None:
            self.tie_qk_scheme = tie_qk_scheme
        if self.tie_qk_scheme == 'shared':                    # N2: one Parameter object under two names
            self.key.weight = self.query.weight
            if self.key.bias is not None:
                self.key.bias = self.query.bias
        elif self.tie_qk_scheme == 'loose':
            self.key.weight.data.copy_(self.query.weight)
            if self.key.bias is not None:
                self.key.bias.data.copy_(self.query.bias)

    def add_identity_bias(self):
        d = self.attention_mode_dim
        eye = torch.diag(torch.ones(d)) * self.base_initializer_range * self.config.query_idbias_scale
        eye = eye.repeat([1, self.in_feat_dim // d])
        w = self.key.weight.data
        w[:d] = w[:d] * 0.5 + eye.to(w.device)

    def forward(self, in_query, in_key=None, pos_biases=None):
        """in_query [B or 1, U1, C] (a leading 1 = shared by the whole batch, e.g. the attractors);
        in_key [B, U2, C].  Returns [B, U1, F].  pos_biases: a `SlidingPosBiasCode` (table + token grid) --
        the [U1, U2] bias matrix of the reference is never materialised."""
        if in_key is None:
            in_key = in_query
        B, U2, C = in_key.shape
        U1, M, d = in_query.shape[1], self.num_modes, self.attention_mode_dim
        shared_q = in_query.shape[0] == 1 and B > 1
        gmax = torch.zeros(1, dtype=torch.float32, device=in_key.device)
        drop = self.attention_probs_dropout_prob if self.training else 0.0
        if (self.reassociate_projections and B * U2 >= self.reassociate_min_rows and M == 1 and in_query.shape[0] == 1 and 2 * U1 <= U2 and pos_biases is None
                and self.out_trans.first_linear.bias is None and self.out_trans.num_scales == 0):
            # In-squeeze layer: few shared queries (the attractors) attend to many tokens with ONE mode.  The key and value
            # projections are linear maps of the U2 tokens that are immediately contracted with U1-row operands, so they are
            # re-associated onto the U1 side:   q (X Wk^T + bk)^T = (q Wk) X^T + (q.bk) 1^T   and   P (X Wv^T) = (P X) Wv^T.
            # Same function, same parameter gradients (autograd through the same GEMM op), fp32 rounding aside; the two
            # [B*U2, C] x [C, C] projection GEMMs (158 GFLOP each at cfg2, plus 2x that in backward) become U1-row GEMMs.
            alpha = 1.0 / math.sqrt(d)
            q = SF.linear(in_query, self.query.weight, self.query.bias)                  # :559 (U1 attractor rows)
            wk, bk = self.key.weight, self.key.bias
            qk = SF.bgemm(q, wk, GemmSpec(U1, C, d, (0, 0, d, 1), (0, 0, 1, C), (0, 0, C), (U1, C), alpha=alpha))
            rb = None
            if bk is not None:
                rb = SF.bgemm(q, bk.view(1, d), GemmSpec(U1, 1, d, (0, 0, d, 1), (0, 0, d, 1), (0, 0, 1), (U1,), alpha=alpha))
            scores = SF.bgemm(qk, in_key, GemmSpec(U1, U2, C, (0, 0, C, 1), (U2 * C, 0, C, 1), (U1 * U2, B * U1 * U2, U2),
                                                   (1, B, U1, U2), nb=(B, 1), bias_mode=SF.BIAS_M), bias=rb, gmax=gmax)
            self.attn_max_dev = gmax
            probs = SF.softmax(scores, self.attn_clip, gmax, drop)
            self.attention_scores = scores if self.keep_attn_scores else None
            return self.out_trans(in_key, probs, value_last=True)
        k = SF.linear(in_key, self.key.weight, self.key.bias)                            # :560
        if (self.reassociate_projections and B * U1 >= self.reassociate_min_rows and pos_biases is None and not shared_q and in_query.shape[0] == B
                and U2 * (C + U1 * (M - 1)) < U1 * C):
            # Squeeze-out layer: MANY tokens (queries) attend to few attractors.  The query projection is a linear map of the U1
            # tokens that is immediately contracted with the U2 keys, so it is folded into the key side:
            #   (X Wq_m^T + 1 bq_m^T) k_m^T = X (Wq_m^T k_m^T) + 1 (k_m bq_m)^T
            # G[b,m] = Wq_m^T k[b,m]^T is a [C, U2] operand per (sample, mode), the bias term a [U2] row vector per (sample, mode);
            # cost 2 C^2 U2 + 2 U1 U2 C M instead of 2 U1 C^2 + 2 U1 U2 C per sample (the test above; cfg2: 100 vs 180 GFLOP).
            alpha = 1.0 / math.sqrt(d)
            wq, bq = self.query.weight, self.query.bias
            beta = None
            if bq is not None:
                beta = SF.bgemm(bq, k, GemmSpec(1, U2, d, (0, d, d, 1), (U2 * M * d, d, M * d, 1), (M * U2, U2, U2),
                                                (B, M, U2), nb=(B, M), alpha=alpha))
            if self.interleave_modes and U2 % 4 == 0:
                # r06: G [B, C, M, U2] and scores / probs [B, U1, M, U2] -- the modes interleaved with the attractor index, so that a token's M score rows are ONE
                # row of M U2 entries.  Forward is the same M products per sample (other strides, same operands); in backward the token gradient
                # dX = sum_m dS_m G_m^T becomes ONE contraction of depth M U2 (= 1024) per sample instead of M short ones (K = 256: epilogue-bound, §5i)
                # written as 0.7 GB of slabs and summed by a further pass (SF._grad_operand).  ExpandedFeatTrans reads the interleaved probabilities in place.
                G = SF.bgemm(wq, k, GemmSpec(C, U2, d, (0, d * C, 1, C), (U2 * M * d, d, M * d, 1), (C * M * U2, U2, M * U2),
                                             (B, C, M, U2), nb=(B, M), alpha=alpha))
                scores = SF.bgemm(in_query, G, GemmSpec(U1, U2, C, (U1 * C, 0, C, 1), (C * M * U2, U2, 1, M * U2),
                                                        (U1 * M * U2, U2, M * U2), (B, U1, M, U2), nb=(B, M),
                                                        bias_mode=SF.BIAS_N, bias_b0=M * U2, bias_b1=U2), bias=beta, gmax=gmax)
                self.attn_max_dev = gmax
                probs = SF.softmax(scores, self.attn_clip, gmax, drop)
                self.attention_scores = scores.permute(2, 0, 1, 3) if self.keep_attn_scores else None
                return self.out_trans(in_key, probs, modes_interleaved=True)
            G = SF.bgemm(wq, k, GemmSpec(C, U2, d, (0, d * C, 1, C), (U2 * M * d, d, M * d, 1), (M * C * U2, C * U2, U2),
                                         (B, M, C, U2), nb=(B, M), alpha=alpha))
            scores = SF.bgemm(in_query, G, GemmSpec(U1, U2, C, (U1 * C, 0, C, 1), (M * C * U2, C * U2, 1, U2),
                                                    (U1 * U2, B * U1 * U2, U2), (M, B, U1, U2), nb=(B, M),
                                                    bias_mode=SF.BIAS_N, bias_b0=M * U2, bias_b1=U2), bias=beta, gmax=gmax)
            self.attn_max_dev = gmax
            probs = SF.softmax(scores, self.attn_clip, gmax, drop)
            self.attention_scores = scores if self.keep_attn_scores else None
            return self.out_trans(in_key, probs)
        q = SF.linear(in_query, self.query.weight, self.query.bias)                      # :559
        # scores[m,b] = q[b,:,m*d:(m+1)*d] k[b,:,m*d:(m+1)*d]^T / sqrt(d), max tracked in the epilogue (:566-570)
        scores = SF.bgemm(q, k, GemmSpec(U1, U2, d, (0 if shared_q else U1 * C, d, C, 1), (U2 * C, d, C, 1),
                                         (U1 * U2, B * U1 * U2, U2), (M, B, U1, U2), nb=(B, M),
                                         alpha=1.0 / math.sqrt(d)), gmax=gmax)
        self.attn_max_dev = gmax
        if pos_biases is not None:                                                       # :578-580 then :590-592
            assert U1 == U2 == pos_biases.numel, 'positional biases need self-attention over the whole token grid'
            scores = SF.pos_bias_add(scores, pos_biases.table, pos_biases.grid_shape, self.pos_code_weight,
                                     self.attn_clip, gmax)
            probs = SF.softmax(scores, self.attn_clip, None, drop)                       # clamp already applied
        else:
            probs = SF.softmax(scores, self.attn_clip, gmax, drop)                       # :578-605
        self.attention_scores = scores if self.keep_attn_scores else None
        return self.out_trans(in_key, probs)


class CrossMinceAttFeatTrans(nn.Module):
    """Multi-scale ("mince") self-attention (reference :612-785): the token grid is down-sampled by every mince scale, each scale
    attends with an equal slice of every mode's Q/K channels and owns a slice of the value channels (ExpandedFeatTrans).
    As in the reference this is NOT a CrossAttFeatTrans, so SegtranInitWeights neither ties query/key nor adds the identity
    bias to it (:1259-1263)."""

    def __init__(self, config, name):
        super().__init__()
        _unsupported(config)
        self.config, self.name = config, name
        self.num_modes, self.in_feat_dim, self.feat_dim = config.num_modes, config.in_feat_dim, config.feat_dim
        self.attention_mode_dim = self.in_feat_dim // self.num_modes
        self.att_size_allmode = self.num_modes * self.attention_mode_dim
        self.query = nn.Linear(self.in_feat_dim, self.att_size_allmode, bias=config.qk_have_bias)
        self.key = nn.Linear(self.in_feat_dim, self.att_size_allmode, bias=config.qk_have_bias)
        assert config.use_mince_transformer and config.mince_scales
        self.mince_scales, self.num_scales = config.mince_scales, len(config.mince_scales)
        self.mince_qk_channel_indices, _ = fracs_to_indices(self.attention_mode_dim, [1] * self.num_scales)     # :633-634
        self.base_initializer_range = config.base_initializer_range
        self.pos_code_weight = config.pos_code_weight if config.pos_code_type == 'bias' else 1
        self.out_trans = ExpandedFeatTrans(config, name)
        self.attention_probs_dropout_prob = config.attention_probs_dropout_prob
        self.keep_attn_scores = config.use_attn_consist_loss
        self.tie_qk_scheme = config.tie_qk_scheme
        self.attn_clip = float(config.attn_clip)
        self.attention_scores = None
        self.attn_max_dev = None

    tie_qk = CrossAttFeatTrans.tie_qk
    add_identity_bias = CrossAttFeatTrans.add_identity_bias

    def forward(self, in_query, query_geoshape, in_key=None, key_geoshape=None, pos_biases=None):
        """in_query [B, U, C] over the grid `query_geoshape`; pos_biases: None or one SlidingPosBiasCode / None per scale."""
        if in_key is not None and in_key is not in_query:
            raise NotImplementedError('the mince transformer is built for self-attention (its only use, reference :953)')
        geoshape = tuple(int(g) for g in query_geoshape)
        B, U, C = in_query.shape
        M, d = self.num_modes, self.attention_mode_dim
        q = SF.linear(in_query, self.query.weight, self.query.bias)                      # :707
        k = SF.linear(in_query, self.key.weight, self.key.bias)                          # :708
        q4, k4 = q.view(B, U, M, d), k.view(B, U, M, d)
        drop = self.attention_probs_dropout_prob if self.training else 0.0
        probs, maxes = [], []
        for s_, scale in enumerate(self.mince_scales):
            Lc, Rc = self.mince_qk_channel_indices[s_], self.mince_qk_channel_indices[s_ + 1]
            ds = Rc - Lc
            qs = SF.interp_tokens(q4[..., Lc:Rc].reshape(B, U, M * ds), geoshape, scale_factor=1. / scale)      # :725-731
            ks = SF.interp_tokens(k4[..., Lc:Rc].reshape(B, U, M * ds), geoshape, scale_factor=1. / scale)
            Us = qs.shape[1]
            gmax = torch.zeros(1, dtype=torch.float32, device=in_query.device)
            scores = SF.bgemm(qs, ks, GemmSpec(Us, Us, ds, (Us * M * ds, ds, M * ds, 1), (Us * M * ds, ds, M * ds, 1),
                                               (Us * Us, B * Us * Us, Us), (M, B, Us, Us), nb=(B, M),
                                               alpha=1.0 / math.sqrt(d)), gmax=gmax)                             # :735-736
            maxes.append(gmax)
            pb = pos_biases[s_] if pos_biases is not None else None
            if pb is not None:                                                                                  # :747-749 then :760-763
                assert pb.numel == Us
                scores = SF.pos_bias_add(scores, pb.table, pb.grid_shape, self.pos_code_weight, self.attn_clip, gmax)
                probs.append(SF.softmax(scores, self.attn_clip, None, drop))
            else:
                probs.append(SF.softmax(scores, self.attn_clip, gmax, drop))
        self.attn_max_dev = maxes
        self.attention_scores = None
        return self.out_trans(in_query, probs, geoshape)


class SqueezedAttFeatTrans(nn.Module):
    """Squeezed attention: N tokens -> A attractors -> N tokens (reference :787-816)."""

    def __init__(self, config, name):
        super().__init__()
        self.config, self.name = config, name
        self.in_feat_dim, self.num_attractors = config.in_feat_dim, config.num_attractors
        config1 = copy.copy(config)                           # in-squeeze: no compression, one mode, no FFN
        config1.feat_dim = config1.in_feat_dim
        config1.num_modes = 1
        config1.has_FFN = config.has_FFN_in_squeeze
        if config.use_mince_transformer:
            raise ValueError('Squeezed transformer cannot be used with Mince transformer; specify --nosqueeze (reference :836-839)')
        self.in_ator_trans = CrossAttFeatTrans(config1, name + '-in-squeeze')
        self.ator_out_trans = CrossAttFeatTrans(config, name + '-squeeze-out')
        self.attractors = Parameter(torch.randn(1, self.num_attractors, self.in_feat_dim))
        self.attention_scores = None

    def forward(self, in_feat, pos_biases=None):
        new_attractors = self.in_ator_trans(self.attractors, in_feat, pos_biases)        # query shared by the batch
        out_feat = self.ator_out_trans(in_feat, new_attractors, pos_biases)
        self.attention_scores = self.ator_out_trans.attention_scores
        return out_feat


class LearnedSinuPosEmbedder(nn.Module):
    def __init__(self, pos_dim, pos_embed_dim, omega=1, affine=False):
        super().__init__()
        assert omega == 1 and not affine
        self.pos_dim, self.pos_embed_dim = pos_dim, pos_embed_dim
        self.pos_fc = nn.Linear(pos_dim, pos_embed_dim, bias=True)

    def forward(self, pos_normed):
        """pos_normed [N, pos_dim] (batch-invariant) -> [N, C]; [B, N, pos_dim] input is accepted when all
        samples share the coordinates (what Segtran2d/3d produce) and yields [B, N, C] by expansion."""
        if pos_normed.dim() == 3:
            code = SF.pos_embed(pos_normed[0], self.pos_fc.weight, self.pos_fc.bias)
            return code.unsqueeze(0).expand(pos_normed.shape[0], -1, -1)
        return SF.pos_embed(pos_normed, self.pos_fc.weight, self.pos_fc.bias)


class SlidingPosBiasCode:
    """What SlidingPosBiases*.forward hands to the attention layers here: the learnable offset table and the
    token grid.  bias[i, j] = table[pos(j) - pos(i) + R] inside the radius, 0 outside (reference :1051-1072)."""
    __slots__ = ('table', 'grid_shape', 'numel')

    def __init__(self, table, grid_shape):
        self.table, self.grid_shape = table, tuple(int(s) for s in grid_shape)
        self.numel = int(np.prod(self.grid_shape))


class _SlidingPosBiases(nn.Module):
    """Sliding-window positional biases (reference :1002-1175).  One Parameter `biases` [(2R+1)]^pos_dim.  The
    reference also registers index buffers all_h1s/... (72 MB at max_pos_size 100x100) used to scatter the table
    into a padded [H,W,H+2R,W+2R] tensor; the kernel computes the offset instead, so those buffers do not exist
    here and are ignored when a reference checkpoint is loaded."""
    _INDEX_BUFFERS = ('all_h1s', 'all_w1s', 'all_d1s', 'all_h2s', 'all_w2s', 'all_d2s')

    def __init__(self, pos_dim, pos_bias_radius=7, max_pos_size=None):
        super().__init__()
        assert pos_dim == self.POS_DIM
        self.pos_dim, self.R, self.max_pos_size = pos_dim, pos_bias_radius, max_pos_size
        self.biases = Parameter(torch.zeros([2 * pos_bias_radius + 1] * pos_dim))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for b in self._INDEX_BUFFERS:
            state_dict.pop(prefix + b, None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, feat_shape, device=None):
        spatial = tuple(feat_shape)[-self.pos_dim:]
        if self.max_pos_size is not None:
            assert all(s <= m for s, m in zip(spatial, self.max_pos_size)), 'feature map exceeds max_pos_size'
        return SlidingPosBiasCode(self.biases, spatial)


class SlidingPosBiases2D(_SlidingPosBiases):
    POS_DIM = 2


class SlidingPosBiases3D(_SlidingPosBiases):
    POS_DIM = 3


class SegtranPosEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.feat_dim = config.trans_in_dim
        self.pos_embed_dim = self.feat_dim
        self.pos_code_type = config.pos_code_type
        if self.pos_code_type == 'lsinu':
            self.pos_coder = LearnedSinuPosEmbedder(config.pos_dim, self.pos_embed_dim, omega=1, affine=False)
        elif self.pos_code_type == 'bias':
            cls = SlidingPosBiases2D if config.pos_dim == 2 else SlidingPosBiases3D
            self.pos_coder = cls(config.pos_dim, config.pos_bias_radius, config.max_pos_size)
        else:
            raise NotImplementedError("pos_code_type='%s': 'lsinu' (default) and 'bias' are built" % self.pos_code_type)
        self.cached_pos_code = None
        self.cached_feat_shape = None

    def forward(self, orig_feat_shape, voxels_pos):
        """voxels_pos [N, pos_dim] or [B, N, pos_dim] (identical rows per sample).  Eval mode caches (:1208-1226)."""
        if self.pos_code_type == 'bias':
            return self.pos_coder(orig_feat_shape)               # nothing to cache: the code is (table, grid)
        vp = voxels_pos[0] if voxels_pos.dim() == 3 else voxels_pos
        if (not self.training) and self.cached_pos_code is not None and self.cached_feat_shape == tuple(vp.shape) \
                and self.cached_pos_code.device == vp.device:
            return self.cached_pos_code
        code = self.pos_coder(vp / vp.max())                                              # :1231
        if not self.training:
            self.cached_pos_code, self.cached_feat_shape = code.detach(), tuple(vp.shape)
        return code


class SegtranFusionEncoder(nn.Module):
    """Stack of squeezed-attention layers with per-layer pre-norm + positional code (reference :819-975)."""

    def __init__(self, config, name):
        super().__init__()
        self.name = name
        self.num_translayers = config.num_translayers
        self.pos_code_type = config.pos_code_type
        self.translayer_compress_ratios = config.translayer_compress_ratios
        self.translayer_dims = config.translayer_dims
        self.hidden_dropout_prob = config.hidden_dropout_prob
        self.use_squeezed_transformer = config.use_squeezed_transformer
        self.use_mince_transformer = config.use_mince_transformer
        if self.use_squeezed_transformer and self.use_mince_transformer:
            raise ValueError('Squeezed transformer cannot be used with Mince transformer; specify --nosqueeze (reference :836-839)')
        if self.use_squeezed_transformer and self.pos_code_type == 'bias':
            raise ValueError("Squeezed transformer cannot use positional biases; specify --nosqueeze (reference :836-844)")
        self.pos_code_weight = config.pos_code_weight if self.pos_code_type != 'bias' else 0      # :847-850
        if self.use_mince_transformer:                                                            # :852-861
            self.num_scales, self.mince_scales = len(config.mince_scales), config.mince_scales
            if self.pos_code_type == 'bias':
                self.pos_code_layers = nn.ModuleList([SegtranPosEncoder(config) for _ in range(self.num_scales)])
            else:
                self.pos_code_layer = SegtranPosEncoder(config)
        else:
            self.num_scales = 0
            self.pos_code_layer = SegtranPosEncoder(config)
        # --nosqueeze: plain multi-mode self-attention over all N tokens (:873-878); --mince: its multi-scale variant
        TransformerClass = SqueezedAttFeatTrans if self.use_squeezed_transformer else \
            (CrossMinceAttFeatTrans if self.use_mince_transformer else CrossAttFeatTrans)
        layers = []
        for i in range(self.num_translayers):
            c2 = copy.copy(config)
            c2.in_feat_dim, c2.feat_dim = self.translayer_dims[i], self.translayer_dims[i + 1]
            layers.append(TransformerClass(c2, '%s%d' % (name, i)))
        self.translayers = nn.ModuleList(layers)
        dims = self.translayer_dims[:-1]
        self.comb_norm_layers = nn.ModuleList([nn.LayerNorm(d, eps=1e-12, elementwise_affine=False) for d in dims])
        self.vfeat_norm_layers = nn.ModuleList([nn.LayerNorm(d, eps=1e-12, elementwise_affine=True) for d in dims])
        self.use_attn_consist_loss = config.use_attn_consist_loss
        if self.use_attn_consist_loss:
            raise NotImplementedError('attention-consistency loss is outside the hot path')
        self.layers_vfeat, self.layers_attn_scores, self.orig_feat_shape = [], None, None

    def forward(self, vfeat, voxels_pos, vmask, orig_feat_shape):
        """vfeat [B,N,C0]; voxels_pos [N,pd] or [B,N,pd]; vmask [B,N,1] or [B,N] (bool/float)."""
        self.layers_vfeat = []
        B, N, _ = vfeat.shape
        mask = vmask.reshape(B, N).to(torch.float32)
        geoshape = tuple(int(g) for g in orig_feat_shape)
        if self.num_scales > 0 and self.pos_code_type == 'bias':                          # one bias table per scale (:920-923)
            pos_code = None
            biases = [self.pos_code_layers[s_](shp, voxels_pos)
                      for s_, shp in enumerate(multi_resize_shape(geoshape, self.mince_scales))]
        else:
            pos_code = self.pos_code_layer(orig_feat_shape, voxels_pos)                   # [N, C0], once per forward
            biases = pos_code if self.pos_code_type == 'bias' else None
        for i, translayer in enumerate(self.translayers):
            nl = self.vfeat_norm_layers[i]
            drop = self.hidden_dropout_prob if (self.training and i == 0) else 0.0       # :944-945
            if biases is not None:          # codes go to the attention scores; tokens get LN_affine only (:937-940)
                feat = SF.prenorm(vfeat, nl.weight, nl.bias, None, mask, 0.0, drop)
            else:
                feat = SF.prenorm(vfeat, nl.weight, nl.bias, pos_code, mask, self.pos_code_weight, drop)
            if self.num_scales > 0:                                                       # :952-953
                vfeat = translayer(feat, geoshape, pos_biases=biases)
            elif biases is not None:
                vfeat = translayer(feat, pos_biases=biases)
            else:
                vfeat = translayer(feat)
            self.layers_vfeat.append(vfeat)
        self.layers_attn_scores = None
        self.orig_feat_shape = orig_feat_shape
        return vfeat


class SegtranInitWeights(nn.Module):
    """Initialisation mix-in (reference :1241-1264)."""

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        self.config = config

    def init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            if not (np.array(module.weight.shape) < self.config.min_feat_dim).all():
                module.weight.data.normal_(mean=0.0, std=self.config.base_initializer_range)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def tie_qk(self, module):
        if isinstance(module, CrossAttFeatTrans) and module.tie_qk_scheme != 'none':
            module.tie_qk()

    def add_identity_bias(self, module):
        if isinstance(module, (CrossAttFeatTrans, ExpandedFeatTrans)):
            module.add_identity_bias()
