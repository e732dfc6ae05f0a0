"""Import the read-only reference (/root/reference) on CPU.  ONLY usable in the build container.

Used by tests/golden/make_golden.py to (a) validate oracle/ against the real reference and
(b) emit golden fixtures.  Nothing here travels to the GPU box in a usable form (it needs
/root/reference); nothing in segtran_amd/ imports it.  Recipe documented in SURVEY.md 8(c).
"""
import sys, types, io, contextlib
from argparse import Namespace
import torch

REF = '/root/reference/code'


def _install_stubs():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if 'timm' not in sys.modules:
        tm = types.ModuleType('timm.models')
        for n in ('tf_efficientnetv2_s_in21k', 'tf_efficientnetv2_m_in21k', 'tf_efficientnetv2_l_in21k'):
            setattr(tm, n, None)
        t = types.ModuleType('timm'); t.models = tm
        sys.modules['timm'] = t; sys.modules['timm.models'] = tm
    if 'train_util' not in sys.modules:
        tu = types.ModuleType('train_util'); tu.batch_norm = None
        sys.modules['train_util'] = tu


COMMON = dict(use_pretrained=False, bb_feat_upsize=True, in_fpn_use_bn=False, use_squeezed_transformer=True,
              num_modes=4, trans_output_type='private', mid_type='shared', pos_code_type='lsinu',
              pos_code_weight=1.0, pos_bias_radius=7, ablate_multihead=False, out_fpn_do_dropout=False,
              has_FFN_in_squeeze=False, attn_clip=500, qk_have_bias=True, tie_qk_scheme='shared',
              device='cpu', eval_robustness=False, use_attn_consist_loss=False,
              use_mince_transformer=False, mince_scales=None, mince_channel_props=None, dropout_prob=0.2,
              in_fpn_layers='34', out_fpn_layers='1234', in_fpn_scheme='AN', out_fpn_scheme='AN')


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def ref_shared():
    _install_stubs()
    import networks.segtran_shared as ss
    return ss


def ref_segtran2d(num_classes=3, num_attractors=256, num_translayers=3, compress=(1, 1, 2, 2), **over):
    _install_stubs()
    from networks.segtran2d import Segtran2d, CONFIG
    kw = dict(COMMON); kw.update(over)
    a = Namespace(num_classes=num_classes, backbone_type='eff-b4', num_attractors=num_attractors,
                  num_translayers=num_translayers, translayer_compress_ratios=list(compress),
                  num_modalities=0, use_global_bias=False, **kw)
    def build():
        CONFIG.update_config(a)
        return Segtran2d(CONFIG)
    return quiet(build)


def cpu_torch():
    """`torch` stand-in for exec-ing reference functions that hard-code device='cuda' (test_util2d.py:184, test_util3d.py:130):
    factory calls are redirected to the CPU, everything else is the real module."""
    class _T:
        def __getattr__(s, k): return getattr(torch, k)
        def _cpu(s, fn, *a, **kw):
            if kw.get('device') == 'cuda': kw['device'] = 'cpu'
            return fn(*a, **kw)
        def tensor(s, *a, **kw): return s._cpu(torch.tensor, *a, **kw)
        def zeros(s, *a, **kw): return s._cpu(torch.zeros, *a, **kw)
        def ones(s, *a, **kw): return s._cpu(torch.ones, *a, **kw)
    return _T()


def ref_segtran3d(num_classes=4, num_attractors=1024, num_translayers=1, compress=(1, 1), **over):
    _install_stubs()
    import networks.segtran3d as s3
    from networks.segtran3d import Segtran3d, CONFIG

    class _T:                                   # works around segtran3d.py:464 device='cuda'
        def __getattr__(s, k): return getattr(torch, k)
        def tensor(s, *a, **kw):
            if kw.get('device') == 'cuda': kw['device'] = 'cpu'
            return torch.tensor(*a, **kw)
    s3.torch = _T()
    kw = dict(COMMON); kw.update(over)
    a = Namespace(num_classes=num_classes, backbone_type='i3d', num_attractors=num_attractors,
                  num_translayers=num_translayers, translayer_compress_ratios=list(compress),
                  orig_in_channels=4, inchan_to3_scheme='bridgeconv', D_groupsize=1, D_pool_K=2,
                  out_fpn_upsampleD_scheme='interp', input_scale=(1, 1, 1), **kw)
    def build():
        CONFIG.update_config(a)
        return Segtran3d(CONFIG)
    return quiet(build)


def ref_efficientnet_b4():
    _install_stubs()
    from efficientnet.model import EfficientNet
    return quiet(EfficientNet.from_name, 'efficientnet-b4', stem_stride=1)


def ref_i3d():
    _install_stubs()
    from networks.aj_i3d.aj_i3d import InceptionI3d
    return InceptionI3d(do_pool1=False)


def ref_bertadam():
    _install_stubs()
    from optimization import BertAdam
    return BertAdam


def ref_dice():
    _install_stubs()
    from utils.losses import dice_loss_indiv
    return dice_loss_indiv
