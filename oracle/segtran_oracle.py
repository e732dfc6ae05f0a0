"""CPU oracle for the Segtran `--net segtran` train-step hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch fp32 *restatement* of the reference
algorithm (askerlee/segtran), written functionally over a flat ``state_dict`` whose keys are
the reference's own checkpoint keys.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing under ``segtran_amd/`` does, and
the product path never falls back to it.

Pinning: the reference ships no tests / golden vectors.  The oracle is
pinned against the reference *itself*, imported on CPU in the build container by
``tests/golden/make_golden.py`` (which also writes the fixtures in ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` re-checks oracle-vs-fixture everywhere (no reference needed).

Every function cites the reference file:line it restates (paths relative to
``/root/reference/code``).  Quirks N1-N8 of SURVEY.md section 8(a) are reproduced on purpose.
"""
import math
import torch
import torch.nn.functional as F

LN_EPS = 1e-12          # networks/segtran_shared.py:262,286,361,889-893,985  (N4)
BN_EPS = 1e-3           # efficientnet/utils.py:533 ; networks/aj_i3d/aj_i3d.py:65
BN_MOM = 0.01           # 1-0.99 (efficientnet/model.py:43) ; aj_i3d.py:65
GN_GROUPS = 8           # networks/segtran2d.py:34


def _ln(x, sd, prefix, affine=True):
    C = x.shape[-1]
    w = sd[prefix + '.weight'] if affine else None
    b = sd[prefix + '.bias'] if affine else None
    return F.layer_norm(x, (C,), w, b, LN_EPS)


# ----------------------------------------------------------------------------------------
# Squeeze-and-Expansion transformer                      networks/segtran_shared.py
# ----------------------------------------------------------------------------------------
def expanded_feat_trans(sd, p, input_feat, probs, num_modes, has_ffn):
    """ExpandedFeatTrans.forward  (segtran_shared.py:404-476).

    input_feat [B,U2,IF]; probs [B,M,U1,U2] -> [B,U1,F].  Dropout-free (eval / p=0)."""
    B, U2, _ = input_feat.shape
    M = num_modes
    W_v = sd[p + '.first_linear.weight']                       # [M*F, IF], bias=False (:360, v_has_bias)
    Fd = W_v.shape[0] // M
    v = input_feat @ W_v.t()                                   # :414
    v4 = v.view(B, U2, M, Fd).permute(0, 2, 1, 3)              # :416-419  == [B,M,U2,F]
    fused = probs @ v4                                         # :447      [B,M,U1,F]
    if not has_ffn:                                            # :452-457  in-squeeze branch
        sc = fused @ sd[p + '.feat_softaggr.feat2score.weight'].t() + sd[p + '.feat_softaggr.feat2score.bias']
        aggr = (fused * sc.softmax(dim=1)).sum(dim=1)          # LearnedSoftAggregate :318-325
        return _ln(aggr, sd, p + '.first_norm_layer')
    # MMSharedMid (:232-251): one [F,F] linear shared by all modes + exact-erf GELU.
    h = F.gelu(fused @ sd[p + '.intermediate.shared_linear.weight'].t()
               + sd[p + '.intermediate.shared_linear.bias'])
    # MMPrivateOutput (:266-275): Conv1d(groups=M, k=1) == per-mode private linear.
    Wg = sd[p + '.output.group_linear.weight'].view(M, Fd, Fd)   # [M*F, F, 1]
    bg = sd[p + '.output.group_linear.bias'].view(M, 1, Fd)
    z = torch.einsum('bmuf,mgf->bmug', h, Wg) + bg
    # N1: the residual `x + shortcut` (:269) is computed but :272 views `x`, so it is dropped.
    zn = _ln(z, sd, p + '.output.resout_norm_layer')
    sc = zn @ sd[p + '.feat_softaggr.feat2score.weight'].t() + sd[p + '.feat_softaggr.feat2score.bias']
    return (zn * sc.softmax(dim=1)).sum(dim=1)                 # :466-467


def cross_att_feat_trans(sd, p, in_query, in_key, num_modes, has_ffn, attn_clip=500., pos_biases=None,
                         pos_code_weight=1.0, stats=None):
    """CrossAttFeatTrans.forward  (segtran_shared.py:553-610)."""
    M = num_modes
    Wq, Wk = sd[p + '.query.weight'], sd[p + '.key.weight']    # N2: tied => identical tensors
    bq, bk = sd.get(p + '.query.bias'), sd.get(p + '.key.bias')
    q = F.linear(in_query, Wq, bq)                             # :559
    k = F.linear(in_key, Wk, bk)                               # :560
    B, U1, A = q.shape
    d = A // M
    q4 = q.view(B, U1, M, d).permute(0, 2, 1, 3)               # :548-551
    k4 = k.view(B, -1, M, d).permute(0, 2, 1, 3)
    s = (q4 @ k4.transpose(-1, -2)) / math.sqrt(d)             # :566-567
    smax = s.max().item()                                      # :570
    if stats is not None:
        stats.append(smax)
    if smax > attn_clip:                                       # :578-580  N5: only when global max > clip
        s = s.clamp(-attn_clip, attn_clip)
    if pos_biases is not None:                                 # :590-592
        s = s + pos_code_weight * pos_biases
    probs = s.softmax(dim=-1)                                  # :601
    return expanded_feat_trans(sd, p + '.out_trans', in_key, probs, M, has_ffn)   # :608


def fracs_to_indices(feat_dim, props):
    """segtran_shared.py:68-87 -- channel boundaries of the mince scales (the last scale takes the remainder)."""
    fr = [float(v) for v in props]
    tot = sum(fr)
    idx = [0]
    for f in fr[:-1]:
        idx.append(idx[-1] + int(f / tot * feat_dim))
    idx.append(feat_dim)
    return idx


def multi_resize_shape(shape, scales):
    """segtran_shared.py:38-43."""
    return [tuple(int(s / sc) for s in shape) for sc in scales]


def resize_flat_features(x, geoshape, scale=None, orig_geoshape=None):
    """segtran_shared.py:47-66: tokens [B,M,N,C] -> grid [B,M*C,*geoshape] -> F.interpolate (scale_factor OR size; a given
    scale_factor is also the coordinate step, ATen's recompute_scale_factor=None rule) -> tokens [B,M,N',C]."""
    B, M, N, C = x.shape
    g = x.permute(0, 1, 3, 2).reshape(B, M * C, *geoshape)
    mode = ('linear', 'bilinear', 'trilinear')[len(geoshape) - 1]
    g = F.interpolate(g, size=orig_geoshape, scale_factor=scale, mode=mode, align_corners=False)
    return g.reshape(B, M, C, -1).permute(0, 1, 3, 2)


def cross_mince_att_feat_trans(sd, p, in_feat, geoshape, num_modes, mince_scales, mince_channel_props, attn_clip=500.,
                               pos_biases=None, pos_code_weight=1.0, stats=None):
    """CrossMinceAttFeatTrans.forward (segtran_shared.py:699-785) + the mince branch of ExpandedFeatTrans.forward (:421-443):
    self-attention computed separately on S down-sampled copies of the token grid, each scale owning a slice of every mode's
    Q/K channels (equal split, :633-634) and of the value channels (mince_channel_props, :353-354); fused values are resized
    back and concatenated.  Query and key are NOT tied here (SegtranInitWeights.tie_qk only matches CrossAttFeatTrans, :1259)."""
    M = num_modes
    q = F.linear(in_feat, sd[p + '.query.weight'], sd.get(p + '.query.bias'))          # :707
    k = F.linear(in_feat, sd[p + '.key.weight'], sd.get(p + '.key.bias'))              # :708
    B, U, A = q.shape
    d = A // M
    q4 = q.view(B, U, M, d).permute(0, 2, 1, 3)
    k4 = k.view(B, U, M, d).permute(0, 2, 1, 3)
    qk_idx = fracs_to_indices(d, [1] * len(mince_scales))
    probs = []
    for s_, scale in enumerate(mince_scales):
        L, R = qk_idx[s_], qk_idx[s_ + 1]
        qs = resize_flat_features(q4[..., L:R], geoshape, 1. / scale)                  # :725-731
        ks = resize_flat_features(k4[..., L:R], geoshape, 1. / scale)
        sc = (qs @ ks.transpose(-1, -2)) / math.sqrt(d)                                # :735-736 (sqrt of the FULL mode dim)
        smax = sc.max().item()
        if stats is not None:
            stats.append(smax)
        if smax > attn_clip:                                                           # :747-749
            sc = sc.clamp(-attn_clip, attn_clip)
        if pos_biases is not None and pos_biases[s_] is not None:                      # :760-763
            sc = sc + pos_code_weight * pos_biases[s_]
        probs.append(sc.softmax(dim=-1))                                               # :769
    po = p + '.out_trans'
    W_v = sd[po + '.first_linear.weight']
    Fd = W_v.shape[0] // M
    v4 = (in_feat @ W_v.t()).view(B, U, M, Fd).permute(0, 2, 1, 3)                     # :414-419
    v_idx = fracs_to_indices(Fd, mince_channel_props)
    shapes = multi_resize_shape(geoshape, mince_scales)
    parts = []
    for s_, scale in enumerate(mince_scales):
        L, R = v_idx[s_], v_idx[s_ + 1]
        vs = resize_flat_features(v4[..., L:R], geoshape, 1. / scale)                  # :431
        fs = probs[s_] @ vs                                                            # :436
        parts.append(resize_flat_features(fs, shapes[s_], orig_geoshape=tuple(geoshape)))   # :439
    fused = torch.cat(parts, dim=-1)                                                   # :443
    return _expanded_tail(sd, po, fused, M)


def _expanded_tail(sd, p, fused, M):
    """ExpandedFeatTrans.forward after the value fusion, FFN branch (:459-476); see expanded_feat_trans."""
    Fd = fused.shape[-1]
    h = F.gelu(fused @ sd[p + '.intermediate.shared_linear.weight'].t() + sd[p + '.intermediate.shared_linear.bias'])
    Wg = sd[p + '.output.group_linear.weight'].view(M, Fd, Fd)
    bg = sd[p + '.output.group_linear.bias'].view(M, 1, Fd)
    z = torch.einsum('bmuf,mgf->bmug', h, Wg) + bg
    zn = _ln(z, sd, p + '.output.resout_norm_layer')                                   # N1: residual dropped
    sc = zn @ sd[p + '.feat_softaggr.feat2score.weight'].t() + sd[p + '.feat_softaggr.feat2score.bias']
    return (zn * sc.softmax(dim=1)).sum(dim=1)


def squeezed_att_feat_trans(sd, p, in_feat, num_modes=4, attn_clip=500., stats=None, ffn_in_squeeze=False):
    """SqueezedAttFeatTrans.forward  (segtran_shared.py:809-816)."""
    B = in_feat.shape[0]
    att = sd[p + '.attractors'].expand(B, -1, -1)              # :812
    # in-squeeze: num_modes=1, feat_dim=in_feat_dim, no FFN unless --squeezeuseffn (:796-799)
    att2 = cross_att_feat_trans(sd, p + '.in_ator_trans', att, in_feat, 1, ffn_in_squeeze, attn_clip, stats=stats)
    return cross_att_feat_trans(sd, p + '.ator_out_trans', in_feat, att2, num_modes, True, attn_clip, stats=stats)


def polyformer_layer(sd, p, in_feat, num_modes=4, attn_clip=500., do_layernorm=False):
    """PolyformerLayer.forward (networks/polyformer.py:36-57): 2x average pool, squeeze-and-expansion attention pair WITHOUT FFN
    (M modes aggregated on the raw features, then first_norm_layer), bilinear up-sampling, residual."""
    B, C = in_feat.shape[:2]
    half0 = F.avg_pool2d(in_feat, 2)                                          # :40
    half = half0.transpose(1, -1)                                             # :41 (chan_axis = 1)
    if do_layernorm:
        half = F.layer_norm(half, (C,), None, None, LN_EPS)                   # :44-45
    vfeat = half.reshape(B, -1, C)                                            # :46
    att = sd[p + '.attractors'].expand(B, -1, -1)                             # :48
    att2 = cross_att_feat_trans(sd, p + '.in_ator_trans', att, vfeat, num_modes, False, attn_clip)
    vout = cross_att_feat_trans(sd, p + '.ator_out_trans', vfeat, att2, num_modes, False, attn_clip)
    out_half = vout.transpose(1, -1).reshape(half0.shape)                     # :51-52 (sic: the same transpose/reshape pair as the reference)
    return in_feat + F.interpolate(out_half, size=in_feat.shape[2:], mode='bilinear', align_corners=False)


def unet_forward(sd, x, training=False, use_polyformer=True, num_modes=4, running=None):
    """UNet.forward (networks/unet2d/unet_model.py:36-55) on the parts of unet_parts.py: DoubleConv = (Conv2d 3x3 pad 1 + bias -> BatchNorm2d ->
    ReLU) x 2 (:9-26), Down = MaxPool2d(2) + DoubleConv (:29-41), Up = bilinear x2 with align_corners=True, zero-pad to the skip, cat [skip, up],
    DoubleConv with in // 2 middle channels (:44-70), OutConv 1x1 (:73-78); the Polyformer layer sits before the class projection.
    running (dict, optional): receives the updated running statistics of the BatchNorm layers (training mode)."""
    def dconv(p, t):
        for i in (0, 3):
            t = F.conv2d(t, sd['%s.%d.weight' % (p, i)], sd['%s.%d.bias' % (p, i)], padding=1)
            q = '%s.%d' % (p, i + 1)
            rm, rv = sd[q + '.running_mean'].clone(), sd[q + '.running_var'].clone()
            t = F.relu(F.batch_norm(t, rm, rv, sd[q + '.weight'], sd[q + '.bias'], training, 0.1, 1e-5))
            if running is not None:
                running[q + '.running_mean'], running[q + '.running_var'] = rm, rv
        return t

    def up(p, deep, skip):
        u = F.interpolate(deep, scale_factor=2, mode='bilinear', align_corners=True)
        dy, dx = skip.shape[2] - u.shape[2], skip.shape[3] - u.shape[3]
        u = F.pad(u, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return dconv(p + '.conv.double_conv', torch.cat([skip, u], dim=1))

    x1 = dconv('inc.double_conv', x)
    x2 = dconv('down1.maxpool_conv.1.double_conv', F.max_pool2d(x1, 2))
    x3 = dconv('down2.maxpool_conv.1.double_conv', F.max_pool2d(x2, 2))
    x4 = dconv('down3.maxpool_conv.1.double_conv', F.max_pool2d(x3, 2))
    x5 = dconv('down4.maxpool_conv.1.double_conv', F.max_pool2d(x4, 2))
    y = up('up4', up('up3', up('up2', up('up1', x5, x4), x3), x2), x1)
    if use_polyformer:
        y = polyformer_layer(sd, 'polyformer.polyformer_layers.0', y, num_modes)
    return F.conv2d(y, sd['outc.conv.weight'], sd['outc.conv.bias'])


def learned_sinu_pos_embed(sd, p, pos_normed):
    """LearnedSinuPosEmbedder.forward  (segtran_shared.py:989-998), omega=1, no affine."""
    z = F.linear(pos_normed, sd[p + '.pos_fc.weight'], sd[p + '.pos_fc.bias'])
    mix = torch.stack((torch.sin(z[..., 0::2]), torch.cos(z[..., 1::2])), dim=-1).view(z.shape)
    return F.layer_norm(mix, (z.shape[-1],), None, None, LN_EPS)


def sliding_pos_biases(table, shape):
    """SlidingPosBiases2D/3D.forward (segtran_shared.py:1051-1072, 1152-1175), stated as the
    relative-offset lookup it is equivalent to:  bias[i,j] = table[dj-di+R] if |d|<=R else 0."""
    R = (table.shape[0] - 1) // 2
    grids = torch.meshgrid(*[torch.arange(s) for s in shape], indexing='ij')
    coords = torch.stack([g.reshape(-1) for g in grids], dim=1)           # [N, pos_dim]
    delta = coords[None, :, :] - coords[:, None, :]                        # [N1, N2, pos_dim]  (2nd - 1st)
    ok = (delta.abs() <= R).all(dim=-1)
    idx = (delta + R).clamp(0, 2 * R)
    vals = table[tuple(idx[..., i] for i in range(idx.shape[-1]))]
    return torch.where(ok, vals, torch.zeros((), dtype=table.dtype))


def fusion_encoder(sd, p, vfeat, voxels_pos, vmask, translayer_dims, num_modes=4, attn_clip=500.,
                   pos_code_weight=1.0, stats=None, layers_out=None, squeezed=True, pos_code_type='lsinu',
                   feat_shape=None, mince_scales=None, mince_channel_props=None):
    """SegtranFusionEncoder.forward  (segtran_shared.py:907-975).  squeezed=False is --nosqueeze (plain multi-mode
    self-attention, :873-878); pos_code_type 'bias' (:937-940, needs feat_shape) adds sliding positional biases to
    the attention scores instead of positional embeddings to the tokens."""
    pos_normed = voxels_pos / voxels_pos.max()                 # SegtranPosEncoder.forward :1231
    for i in range(len(translayer_dims) - 1):
        vn = _ln(vfeat, sd, '%s.vfeat_norm_layers.%d' % (p, i))                       # :916
        biases = None
        if pos_code_type == 'bias' and mince_scales:
            # --mince: one bias table per scale, applied on that scale's resized grid (:859-861, :920-923)
            biases = [sliding_pos_biases(sd['%s.pos_code_layers.%d.pos_coder.biases' % (p, s_)], shp)
                      for s_, shp in enumerate(multi_resize_shape(feat_shape, mince_scales))]
            fn = vn
        elif pos_code_type == 'bias':
            biases = sliding_pos_biases(sd[p + '.pos_code_layer.pos_coder.biases'], feat_shape)   # :927, :1235-1238
            fn = vn                                                                    # :940
        else:
            pos = learned_sinu_pos_embed(sd, p + '.pos_code_layer.pos_coder', pos_normed)  # :927 (regenerated per layer)
            comb = vn + pos_code_weight * pos[:, :, :translayer_dims[i]]               # :930-932
            fn = F.layer_norm(comb, (comb.shape[-1],), None, None, LN_EPS)             # :934 (no affine)
        fm = fn * vmask                                                                # :946
        lp = '%s.translayers.%d' % (p, i)
        if mince_scales:                                                               # :952-953
            vfeat = cross_mince_att_feat_trans(sd, lp, fm, tuple(feat_shape), num_modes, mince_scales, mince_channel_props,
                                               attn_clip, biases, pos_code_weight if biases is not None else 1.0, stats)
        elif squeezed:
            vfeat = squeezed_att_feat_trans(sd, lp, fm, num_modes, attn_clip, stats)
        else:                                                                          # :955 self-attention
            vfeat = cross_att_feat_trans(sd, lp, fm, fm, num_modes, True, attn_clip, pos_biases=biases,
                                         pos_code_weight=pos_code_weight if biases is not None else 1.0, stats=stats)
        if layers_out is not None:
            layers_out.append(vfeat)
    return vfeat


def gen_all_indices(shape):
    """segtran_shared.py:28-36 - row-major coordinates of every cell, [*shape, len(shape)]."""
    grids = torch.meshgrid(*[torch.arange(s) for s in shape], indexing='ij')
    return torch.stack(grids, dim=len(shape))


# ----------------------------------------------------------------------------------------
# EfficientNet-B4 endpoints                               efficientnet/model.py, utils.py
# ----------------------------------------------------------------------------------------
def _round_filters(f, width=1.4, divisor=8):                   # efficientnet/utils.py:82-108
    f = f * width
    nf = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if nf < 0.9 * f:
        nf += divisor
    return int(nf)


def effnet_b4_blocks(nominal_image_size=380):
    """Per-block (kernel, stride, expand, in, out, se_ch, static_pad) for B4.

    efficientnet/utils.py:491-541 (block table, width 1.4 / depth 1.8), model.py:189-212.
    N6: static 'same' padding comes from the NOMINAL 380-px geometry, and the stem is assumed
    stride 2 for that bookkeeping (model.py:178) even when stem_stride=1."""
    base = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
            (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
    size = math.ceil(nominal_image_size / 2)
    blocks, seg_end = [], []

    def pad_for(sz, k, s):                                     # utils.py:262-270
        o = math.ceil(sz / s)
        tot = max((o - 1) * s + k - sz, 0)
        return (tot // 2, tot - tot // 2)                      # (front, back), same for H and W

    for (r, k, s, e, i, o) in base:
        i, o, r = _round_filters(i), _round_filters(o), int(math.ceil(1.8 * r))
        for j in range(r):
            cin, st = (i, s) if j == 0 else (o, 1)
            blocks.append(dict(k=k, s=st, e=e, cin=cin, cout=o, se=max(1, int(cin * 0.25)),
                               pad=pad_for(size, k, st)))
            if j == 0:
                size = math.ceil(size / s)
        seg_end.append(len(blocks))
    endpoint_blk = [seg_end[i] for i in (0, 1, 2, 4)]          # model.py:184,211-212 -> [2,6,10,22]
    return blocks, endpoint_blk


def _bn(x, sd, p, training, eps=BN_EPS):
    return F.batch_norm(x, sd[p + '.running_mean'].clone(), sd[p + '.running_var'].clone(),
                        sd[p + '.weight'], sd[p + '.bias'], training, BN_MOM, eps)


def _swish(x):
    return x * torch.sigmoid(x)


def _pad2d(x, pad):
    f, b = pad
    return F.pad(x, (f, b, f, b)) if (f or b) else x


def mbconv(sd, p, x, blk, training):
    """MBConvBlock.forward (efficientnet/model.py:82-126); drop_connect off (rate 0 / eval)."""
    inp = x
    if blk['e'] != 1:
        x = _swish(_bn(F.conv2d(x, sd[p + '._expand_conv.weight']), sd, p + '._bn0', training))
    x = F.conv2d(_pad2d(x, blk['pad']), sd[p + '._depthwise_conv.weight'], None, blk['s'], 0, 1, x.shape[1])
    x = _swish(_bn(x, sd, p + '._bn1', training))
    sq = F.adaptive_avg_pool2d(x, 1)
    sq = _swish(F.conv2d(sq, sd[p + '._se_reduce.weight'], sd[p + '._se_reduce.bias']))
    sq = F.conv2d(sq, sd[p + '._se_expand.weight'], sd[p + '._se_expand.bias'])
    x = torch.sigmoid(sq) * x
    x = _bn(F.conv2d(x, sd[p + '._project_conv.weight']), sd, p + '._bn2', training)
    if blk['s'] == 1 and blk['cin'] == blk['cout']:
        x = x + inp
    return x


def effnet_b4_endpoints(sd, p, x, training=False):
    """EfficientNet.extract_endpoints (efficientnet/model.py:240-283), stem_stride=1."""
    blocks, ep_idx = effnet_b4_blocks()
    x = F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[p + '._conv_stem.weight'], None, 1)   # k3, stride 1, pad (1,1)
    x = _swish(_bn(x, sd, p + '._bn0', training))
    feats, prev = [], x
    for i, blk in enumerate(blocks):
        x = mbconv(sd, '%s._blocks.%d' % (p, i), x, blk, training)
        if i in ep_idx:                                         # endpoint = INPUT of block i (:275-277)
            feats.append(prev)
        prev = x
    x = _swish(_bn(F.conv2d(x, sd[p + '._conv_head.weight']), sd, p + '._bn1', training))
    feats.append(x)
    return feats


# ----------------------------------------------------------------------------------------
# Inception-I3D features                                   networks/aj_i3d/aj_i3d.py
# ----------------------------------------------------------------------------------------
def _same_pad3d(x, kernel, stride):
    """aj_i3d.py:8-30 / 68-90 (N7): dynamic TF-'same' padding, ZERO fill, front = pad//2."""
    pads = []
    for dim in (2, 1, 0):                                      # F.pad order: W, H, T
        s, k, size = stride[dim], kernel[dim], x.shape[2 + dim]
        tot = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
        pads += [tot // 2, tot - tot // 2]
    return F.pad(x, pads) if any(pads) else x


def unit3d(sd, p, x, kernel, stride=(1, 1, 1), training=False):
    """Unit3D.forward (aj_i3d.py:75-97): same-pad -> Conv3d(no bias) -> BN(eps 1e-3) -> ReLU."""
    x = F.conv3d(_same_pad3d(x, kernel, stride), sd[p + '.conv3d.weight'], None, stride)
    return F.relu(_bn(x, sd, p + '.bn', training))


def maxpool3d_same(x, kernel, stride):
    return F.max_pool3d(_same_pad3d(x, kernel, stride), kernel, stride)


def inception(sd, p, x, training=False):
    """InceptionModule.forward (aj_i3d.py:121-126)."""
    b0 = unit3d(sd, p + '.b0', x, (1, 1, 1), training=training)
    b1 = unit3d(sd, p + '.b1b', unit3d(sd, p + '.b1a', x, (1, 1, 1), training=training), (3, 3, 3), training=training)
    b2 = unit3d(sd, p + '.b2b', unit3d(sd, p + '.b2a', x, (1, 1, 1), training=training), (3, 3, 3), training=training)
    b3 = unit3d(sd, p + '.b3b', maxpool3d_same(x, (3, 3, 3), (1, 1, 1)), (1, 1, 1), training=training)
    return torch.cat([b0, b1, b2, b3], dim=1)


def i3d_features(sd, p, x, training=False):
    """InceptionI3d.extract_features (aj_i3d.py:325-333) with do_pool1=False (:206-210).

    Returns the five maps Segtran3d consumes (segtran3d.py:430-432)."""
    t = training
    x = unit3d(sd, p + '.Conv3d_1a_7x7', x, (7, 7, 7), (2, 2, 2), t)
    f0 = x                                                      # 'MaxPool3d_2a_3x3' is Identity
    x = unit3d(sd, p + '.Conv3d_2b_1x1', x, (1, 1, 1), training=t)
    x = unit3d(sd, p + '.Conv3d_2c_3x3', x, (3, 3, 3), training=t)
    f1 = x
    x = maxpool3d_same(x, (1, 3, 3), (1, 2, 2))
    x = inception(sd, p + '.Mixed_3b', x, t)
    x = inception(sd, p + '.Mixed_3c', x, t)
    f2 = x
    x = maxpool3d_same(x, (3, 3, 3), (2, 2, 2))
    for name in ('4b', '4c', '4d', '4e', '4f'):
        x = inception(sd, p + '.Mixed_' + name, x, t)
    f3 = x
    x = maxpool3d_same(x, (2, 2, 2), (2, 2, 2))
    x = inception(sd, p + '.Mixed_5b', x, t)
    x = inception(sd, p + '.Mixed_5c', x, t)
    return [f0, f1, f2, f3, x]


# ----------------------------------------------------------------------------------------
# Model assembly                                           networks/segtran2d.py, segtran3d.py
# ----------------------------------------------------------------------------------------
def _conv1x1(x, sd, p):
    w = sd[p + '.weight']
    return (F.conv2d if w.dim() == 4 else F.conv3d)(x, w, sd[p + '.bias'])


def _gn(x, sd, p):
    return F.group_norm(x, GN_GROUPS, sd[p + '.weight'], sd[p + '.bias'])


def _up(x, size):
    return F.interpolate(x, size=tuple(size), mode='bilinear' if x.dim() == 4 else 'trilinear', align_corners=False)


def segtran2d_forward(sd, x, translayer_dims, num_modes=4, training=False, attn_clip=500., stats=None, aux=None,
                      fusion_kw=None, in_fpn_use_bn=False):
    """Segtran2d.forward (segtran2d.py:314-438): eff-b4, in_fpn '34', out_fpn '1234', scheme 'AN'.
    in_fpn_use_bn (--inbn, :143-146, :252): a default nn.BatchNorm2d (eps 1e-5, momentum 0.1) instead of the GroupNorm."""
    B, _, H, W = x.shape
    mask = F.avg_pool2d(x.abs(), 8).sum(dim=1) > 0                                   # get_mask :229-233
    f = effnet_b4_endpoints(sd, 'backbone', x, training)                             # :344-348
    cur = _conv1x1(f[3], sd, 'in_fpn34_conv') + _up(f[4], f[3].shape[2:])            # in_fpn_forward :245-252
    if in_fpn_use_bn:
        cur = F.batch_norm(cur, sd['in_bn4b.running_mean'].clone(), sd['in_bn4b.running_var'].clone(), sd['in_bn4b.weight'],
                           sd['in_bn4b.bias'], training, 0.1, 1e-5)
    else:
        cur = _gn(cur, sd, 'in_gn4b')
    H2, W2 = cur.shape[2:]
    vfeat = cur.permute(0, 2, 3, 1).reshape(B, H2 * W2, -1)                          # :264-266
    vmask = mask.reshape(B, -1, 1)
    pos = gen_all_indices((H2, W2)).view(-1, 2).float() * torch.tensor([[H // H2, W // W2]])   # :372-389
    voxels_pos = pos.unsqueeze(0).repeat(B, 1, 1)
    y = fusion_encoder(sd, 'voxel_fusion', vfeat, voxels_pos, vmask, translayer_dims, num_modes, attn_clip, stats=stats,
                       **(dict(fusion_kw, feat_shape=(H2, W2)) if fusion_kw else {}))
    if aux is not None:
        aux['vfeat'] = vfeat; aux['vmask'] = vmask; aux['fused'] = y
    y = y.view(B, H2, W2, -1).permute(0, 3, 1, 2)                                    # :421-423
    cur = _conv1x1(f[1], sd, 'out_fpn12_conv') + _up(f[2], f[1].shape[2:])           # out_fpn_forward :286-294
    cur = _gn(cur, sd, 'out_gn2b')
    cur = _conv1x1(cur, sd, 'out_fpn23_conv') + _up(f[3], cur.shape[2:])
    cur = _gn(cur, sd, 'out_gn3b')
    out = _conv1x1(cur, sd, 'out_fpn_bridgeconv') + _up(y, cur.shape[2:])            # :304-306
    return _up(_conv1x1(out, sd, 'out_conv'), (H, W))                                # :427-436


def segtran3d_forward(sd, x, translayer_dims, num_modes=4, training=False, attn_clip=500., D_pool_K=2, stats=None):
    """Segtran3d.forward (segtran3d.py:398-498): i3d, bridgeconv 4->3, D_pool_K=2, 'interp' unpool."""
    B, C, H, W, D = x.shape
    rgb = F.conv3d(x, sd['in_bridge_to3.weight'], sd['in_bridge_to3.bias']).permute(0, 1, 4, 2, 3)   # :421-423
    mask = (F.avg_pool3d(rgb.abs(), (4, 8, 8)).sum(dim=1) > 0)                        # get_mask :266-270
    f = i3d_features(sd, 'backbone', rgb, training)                                   # :428-432
    cur = _conv1x1(f[3], sd, 'in_fpn34_conv') + _up(f[4], f[3].shape[2:])             # :300-307
    cur = _gn(cur, sd, 'in_gn4b')
    dp = [cur.shape[2] // D_pool_K, cur.shape[3], cur.shape[4]]
    cur = _up(cur, dp)                                                                # :319
    m = _up(mask.float().unsqueeze(1), dp).squeeze(1) >= 0.5                          # :321-323
    D2, H2, W2 = dp
    vfeat = cur.permute(0, 2, 3, 4, 1).reshape(B, D2 * H2 * W2, -1)
    vmask = m.reshape(B, -1, 1)
    scale = torch.tensor([[D // D2, H // H2, W // W2]], dtype=torch.float32)          # :454-457 (input_scale 1)
    pos = gen_all_indices((D2, H2, W2)).view(-1, 3).float() * scale
    voxels_pos = pos.unsqueeze(0).repeat(B, 1, 1)
    y = fusion_encoder(sd, 'voxel_fusion', vfeat, voxels_pos, vmask, translayer_dims, num_modes, attn_clip, stats=stats)
    y = y.view(B, D2, H2, W2, -1).permute(0, 4, 1, 2, 3)                              # :478-480
    cur = _conv1x1(f[1], sd, 'out_fpn12_conv3d') + _up(f[2], f[1].shape[2:])          # :347-354
    cur = _gn(cur, sd, 'out_gn2b')
    cur = _conv1x1(cur, sd, 'out_fpn23_conv3d') + _up(f[3], cur.shape[2:])
    cur = _gn(cur, sd, 'out_gn3b')
    out = _conv1x1(cur, sd, 'out_fpn_bridgeconv3d') + _up(y, cur.shape[2:])           # :364-367
    out = _up(out, [out.shape[2] * D_pool_K, out.shape[3], out.shape[4]])             # :381-386
    out = out.permute(0, 1, 3, 4, 2)                                                  # :488
    return _up(_conv1x1(out, sd, 'out_conv3d'), (H, W, D))                            # :490-496


# ----------------------------------------------------------------------------------------
# Train-step glue                                          train2d.py, train3d.py, utils/losses.py
# ----------------------------------------------------------------------------------------
def fundus_map_mask(mask, exclusive=False):
    """dataloaders/datasets2d.py:90-139 (4-D branch): uint8 {0,255} -> n-hot [bg,disc,cup]; exclusive (:110-111): disc without cup."""
    out = torch.zeros(mask.shape[0], 3, *mask.shape[2:])
    out[:, 0] = (mask[:, 0] == 0)
    out[:, 1] = (mask[:, 0] >= 1) if not exclusive else ((mask[:, 0] >= 1) & (mask[:, 1] == 0))
    out[:, 2] = (mask[:, 1] >= 1)
    return out


def polyp_map_mask(mask):
    """dataloaders/datasets2d.py:200-223."""
    out = torch.zeros(mask.shape[0], 2, *mask.shape[2:])
    out[:, 0] = (mask[:, 0] == 0)
    out[:, 1] = (mask[:, 0] > 0)
    return out


def brats_map_label(label):
    """dataloaders/datasets3d.py:16-40 (binarize=False): [B,H,W,D] int -> [B,4,H,W,D] (bg, ET, WT, TC)."""
    out = torch.zeros((4,) + tuple(label.shape))
    out[0, label == 0] = 1
    out[1, label == 3] = 1
    out[2, (label == 3) | (label == 1) | (label == 2)] = 1
    out[3, (label == 3) | (label == 1)] = 1
    return out.permute(1, 0, *range(2, out.dim()))


def random_resized_crop(volume, mask, out_size, crop_percents, isotropic=True):
    """dataloaders/datasets3d.py:611-665 (train3d.py:713-715, --randscale): random rescale (one draw when isotropic) of volume and n-hot
    mask by trilinear interpolation, zero padding up to out_size, random crop of out_size.  RNG: torch's global CPU generator, draws in
    the reference's order.  The mask stays continuous (:661-662)."""
    H, W, D = volume.shape[-3:]
    lo, hi = 1 + crop_percents[0], 1 + crop_percents[1]
    sH = torch.rand(1) * (hi - lo) + lo
    if isotropic:
        sW = sD = sH
    else:
        sW = torch.rand(1) * (hi - lo) + lo
        sD = torch.rand(1) * (hi - lo) + lo
    H2, W2, D2 = int(H * sH), int(W * sW), int(D * sD)
    v2 = F.interpolate(volume, size=(H2, W2, D2), mode='trilinear', align_corners=False)
    m2 = F.interpolate(mask, size=(H2, W2, D2), mode='trilinear', align_corners=False)
    Ho, Wo, Do = out_size
    if H2 < Ho or W2 < Wo or D2 < Do:
        ph, pw, pd = max(Ho - H2, 0), max(Wo - W2, 0), max(Do - D2, 0)
        pads = (pd // 2, pd - pd // 2, pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)
        v2, m2 = F.pad(v2, pads, 'constant', 0), F.pad(m2, pads, 'constant', 0)
    H2, W2, D2 = v2.shape[2:]
    h0 = int(torch.randint(H2 - Ho + 1, (1,))); w0 = int(torch.randint(W2 - Wo + 1, (1,))); d0 = int(torch.randint(D2 - Do + 1, (1,)))
    return v2[:, :, h0:h0 + Ho, w0:w0 + Wo, d0:d0 + Do].clone(), m2[:, :, h0:h0 + Ho, w0:w0 + Wo, d0:d0 + Do].clone()


def bce_pos_weight(bce_weight):
    """train2d.py:813-814 / train3d.py."""
    w = torch.tensor(bce_weight, dtype=torch.float32)
    return w * (len(bce_weight) - 1) / w.sum()


def dice_loss_indiv(score, gt):
    """utils/losses.py:47-60."""
    score = score.reshape(score.shape[0], -1)
    gt = gt.float().reshape(gt.shape[0], -1)
    inter = (score * gt).sum(dim=1)
    dice = (2 * inter + 1e-5) / ((score * score).sum(dim=1) + (gt * gt).sum(dim=1) + 1e-5)
    return (1 - dice).mean()


def seg_loss(logits, mask_nhot, pos_weight, dice_w=0.5):
    """train2d.py:1219-1242,1314-1318 / train3d.py:731-756: 0.5*BCE(pos_weight) + 0.5*sum_c w_c Dice_c.

    `logits` are already at mask resolution.  Returns (loss, ce, dice_total, [dice_c])."""
    nc = logits.shape[1]
    perm = (0,) + tuple(range(2, logits.dim())) + (1,)
    ce = F.binary_cross_entropy_with_logits(logits.permute(*perm), mask_nhot.permute(*perm), pos_weight=pos_weight)
    soft = torch.sigmoid(logits)
    cw = torch.ones(nc); cw[0] = 0; cw = cw / cw.sum()                                # train2d.py:1123-1127
    dices = [dice_loss_indiv(soft[:, c], mask_nhot[:, c]) for c in range(1, nc)]
    dice = sum(d * cw[c + 1] for c, d in enumerate(dices))
    return (1 - dice_w) * ce + dice_w * dice, ce, dice, dices


def warmup_linear(x, warmup):
    """optimization.py:25-31."""
    return x / warmup if x < warmup else max((x - 1.) / (warmup - 1.), 0)


def global_clip_(grads, max_norm):
    """nn.utils.clip_grad_norm_ as called at train2d.py:1324-1325 (L2 over all grads, in place)."""
    total = torch.norm(torch.stack([g.norm(2) for g in grads]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def bertadam_step(params, grads, state, lr, weight_decays, warmup, t_total, b1=0.9, b2=0.999, e=1e-6,
                  max_grad_norm=0.05):
    """BertAdam.step (optimization.py:90-164): per-tensor clip 0.05, no bias correction,
    decoupled weight decay, warmup-linear LR.  `state[i]` = dict(step, m, v); grads may be None (N3)."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        st = state[i]
        if not st:
            st.update(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p))
        if max_grad_norm > 0:
            n = g.norm(2)
            g = g * torch.clamp(max_grad_norm / (n + 1e-6), max=1.0)
        st['m'].mul_(b1).add_(g, alpha=1 - b1)
        st['v'].mul_(b2).addcmul_(g, g, value=1 - b2)
        upd = st['m'] / (st['v'].sqrt() + e)
        if weight_decays[i] > 0:
            upd = upd + weight_decays[i] * p
        lr_s = lr * warmup_linear(st['step'] / t_total, warmup) if t_total != -1 else lr
        p.add_(-lr_s * upd)
        st['step'] += 1


# ----------------------------------------------------------------------------------------
# Evaluation path                     test_util2d.py, test_util3d.py
# ----------------------------------------------------------------------------------------
def harden_segmap_nd(mask_soft, batched, T=0.5):
    """harden_segmap2d / harden_segmap3d (datasets2d.py:178-196, datasets3d.py:92-111)."""
    hard = (mask_soft >= T).int()
    if batched:
        hard[:, 0] = (hard[:, 1:].sum(dim=1) == 0)
    else:
        hard[0] = (hard[1:].sum(dim=0) == 0)
    return hard


def make_brats_pred_consistent(preds_soft, is_conservative):
    """datasets3d.py:43-63; preds_soft [4, ...] = (bg, ET, WT, TC)."""
    out = preds_soft.clone()
    if is_conservative:
        out[1] = torch.min(preds_soft[1:], dim=0)[0]
        out[3] = torch.min(preds_soft[2:], dim=0)[0]
    else:
        out[2] = torch.max(preds_soft[1:], dim=0)[0]
        out[3] = torch.max(preds_soft[[1, 3]], dim=0)[0]
    return out


def brats_inv_map_label(orig_probs):
    """datasets3d.py:65-90."""
    inv = torch.zeros_like(orig_probs)
    inv[0] = 1 - orig_probs[2]
    inv[3] = orig_probs[1]
    inv[1] = (orig_probs[3] - orig_probs[1]) * 1.5
    inv[2] = (orig_probs[2] - orig_probs[3]) * 1.5
    return inv


def test_single_batch(net_fn, image_batch, orig_input_size, patch_size, stride, num_classes):
    """test_util2d.py:153-227 for model_type 'segtran': zero-pad to the window size, slide, resize window -> patch, net,
    resize scores -> window, sigmoid, average over the overlap count, harden."""
    B, C, H, W = image_batch.shape
    dx, dy = orig_input_size
    h_pad, w_pad = max(dx - H, 0), max(dy - W, 0)
    hl, wl = h_pad // 2, w_pad // 2
    if h_pad or w_pad:
        image_batch = F.pad(image_batch, (wl, w_pad - wl, hl, h_pad - hl))                  # :174-176
    H2, W2 = image_batch.shape[2:]
    sx = math.ceil((H2 - dx) / stride[0]) + 1                                               # :180-181
    sy = math.ceil((W2 - dy) / stride[1]) + 1
    soft = torch.zeros(B, num_classes, H2, W2); cnt = torch.zeros(B, H2, W2)
    for x in range(sx):
        xs = min(stride[0] * x, H2 - dx)
        for y in range(sy):
            ys = min(stride[1] * y, W2 - dy)
            patch = F.interpolate(image_batch[:, :, xs:xs + dx, ys:ys + dy], size=patch_size, mode='bilinear', align_corners=False)
            with torch.no_grad():
                scores = net_fn(patch)
            scores = F.interpolate(scores, size=orig_input_size, mode='bilinear', align_corners=False)   # :209-210
            soft[:, :, xs:xs + dx, ys:ys + dy] += torch.sigmoid(scores)                     # :212-214
            cnt[:, xs:xs + dx, ys:ys + dy] += 1
    soft = soft / cnt.unsqueeze(1)                                                           # :216
    hard = harden_segmap_nd(soft, True)
    if h_pad or w_pad:
        hard = hard[:, :, hl:hl + H, wl:wl + W]; soft = soft[:, :, hl:hl + H, wl:wl + W]
    return hard, soft


def test_single_case(net_fn, image, orig_patch_size, input_patch_size, batch_size, stride_xy, stride_z, num_classes=4):
    """test_util3d.py:93-184, task 'brats', net_type 'segtran'."""
    C, H, W, D = image.shape
    dx, dy, dz = orig_patch_size
    pads = [max(dx - H, 0), max(dy - W, 0), max(dz - D, 0)]
    lp = [p // 2 for p in pads]
    if any(pads):
        # N9: the reference's F.pad tuple (0, 0, dl, dr, wl, wr, hl, hr) on a [C,H,W,D] tensor pads D by nothing, W by the D pads,
        # H by the W pads and the CHANNEL dim by the H pads (:118-120) -- a volume smaller than the patch cannot run there.
        raise NotImplementedError('reference pads the wrong dims for volumes smaller than the patch (N9); no oracle for that case')
    _, H2, W2, D2 = image.shape
    sx = math.ceil((H2 - dx) / stride_xy) + 1
    sy = math.ceil((W2 - dy) / stride_xy) + 1
    sz = math.ceil((D2 - dz) / stride_z) + 1
    soft = torch.zeros((num_classes,) + tuple(image.shape[1:])); cnt = torch.zeros_like(image[0])
    for x in range(sx):
        xs = min(stride_xy * x, H2 - dx)
        yzs, patches = [], []
        for y in range(sy):
            ys = min(stride_xy * y, W2 - dy)
            for z in range(sz):
                zs = min(stride_z * z, D2 - dz)
                patches.append(image[:, xs:xs + dx, ys:ys + dy, zs:zs + dz]); yzs.append((ys, zs))
                if len(patches) == batch_size or (y == sy - 1 and z == sz - 1):
                    tb = F.interpolate(torch.stack(patches, 0), size=input_patch_size, mode='trilinear', align_corners=False)
                    with torch.no_grad():
                        sc = net_fn(tb)
                    pr = torch.sigmoid(F.interpolate(sc, size=orig_patch_size, mode='trilinear', align_corners=False))
                    for i, (ys_i, zs_i) in enumerate(yzs):
                        soft[:, xs:xs + dx, ys_i:ys_i + dy, zs_i:zs_i + dz] += pr[i]
                        cnt[xs:xs + dx, ys_i:ys_i + dy, zs_i:zs_i + dz] += 1
                    patches, yzs = [], []
    soft = soft / cnt.unsqueeze(0)
    soft = make_brats_pred_consistent(soft, False)                                           # :168-170
    hard = torch.zeros_like(soft)
    hard[1:] = (soft[1:] >= 0.5)
    hard[0] = (hard[1:].sum(dim=0) == 0)
    if any(pads):
        sl = (slice(None), slice(lp[0], lp[0] + H), slice(lp[1], lp[1] + W), slice(lp[2], lp[2] + D))
        hard, soft = hard[sl].clone(), soft[sl].clone()
    return hard, soft


def calc_dice(predictions, gt_mask):
    """test_util2d.py:233-240."""
    gt = gt_mask.float(); pr = predictions.float()
    inter = torch.sum(pr * gt, dim=(-1, -2)); y = torch.sum(gt * gt, dim=(-1, -2)); z = torch.sum(pr * pr, dim=(-1, -2))
    return (2 * inter + 1e-5) / (z + y + 1e-5)
