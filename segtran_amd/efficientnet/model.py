"""EfficientNet-B* feature extractor with the reference's customisations (stem_stride, endpoints).

Host-side mirror of /root/reference/code/efficientnet/{model,utils}.py for the segtran path: same module /
parameter names (so lukemelas-format and reference checkpoints load), same static-'same'-padding quirk N6
(padding derived from the NOMINAL image size, the stem bookkept as stride 2 even when stem_stride=1:
model.py:169-178, utils.py:248-275), same endpoint rule (endpoints are the INPUTS of blocks
`endpoint_blk_indices`, model.py:184,211-212,275-277).

Kernel status: every 1x1 convolution (127 of the 160 convs of B4) runs on libsegx's MFMA GEMM; the 32 depthwise
k3/k5 convolutions, BatchNorm+swish (fused), the squeeze-excite pooling/gating planes and the drop_connect+skip add
run on libsegx's HBM-bound kernels (backbone.hip); the dense 3x3 stem is an implicit GEMM on the MFMA engine
(conv3d.hip with depth 1); the squeeze-excite excitation MLP has its own wave-per-dot-product kernels.  No ATen
arithmetic is left in this file: the drop_connect draw and the skip add happen inside the last BatchNorm pass of a block.
"""
import math
import torch
from torch import nn
from torch.nn import functional as F

from .. import functional as SF

BN_MOM, BN_EPS = 1 - 0.99, 1e-3

# (num_repeat, kernel, stride, expand, in, out, se_ratio)  -- EfficientNet-B0 base table (utils.py:514-522)
_BASE = [(1, 3, 1, 1, 32, 16, 0.25), (2, 3, 2, 6, 16, 24, 0.25), (2, 5, 2, 6, 24, 40, 0.25), (3, 3, 2, 6, 40, 80, 0.25),
         (3, 5, 1, 6, 80, 112, 0.25), (4, 5, 2, 6, 112, 192, 0.25), (1, 3, 1, 6, 192, 320, 0.25)]
# name -> (width, depth, nominal resolution, dropout)   (utils.py:476-488)
_PARAMS = {'efficientnet-b0': (1.0, 1.0, 224, 0.2), 'efficientnet-b1': (1.0, 1.1, 240, 0.2),
           'efficientnet-b2': (1.1, 1.2, 260, 0.3), 'efficientnet-b3': (1.2, 1.4, 300, 0.3),
           'efficientnet-b4': (1.4, 1.8, 380, 0.4), 'efficientnet-b5': (1.6, 2.2, 456, 0.4),
           'efficientnet-b6': (1.8, 2.6, 528, 0.5), 'efficientnet-b7': (2.0, 3.1, 600, 0.5)}


def round_filters(filters, width, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats, depth):
    return int(math.ceil(depth * repeats))


class Conv2dStaticSamePadding(nn.Conv2d):
    """TF-'SAME' conv whose padding is fixed at construction from a given image size (N6)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, image_size=None, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size, stride, **kwargs)
        ih = iw = image_size
        kh, kw = self.weight.shape[-2:]
        sh, sw = self.stride
        oh, ow = math.ceil(ih / sh), math.ceil(iw / sw)
        ph = max((oh - 1) * sh + (kh - 1) * self.dilation[0] + 1 - ih, 0)
        pw = max((ow - 1) * sw + (kw - 1) * self.dilation[1] + 1 - iw, 0)
        self.static_pad = (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)
        self.pointwise = (kh == 1 and kw == 1 and sh == 1 and sw == 1 and self.groups == 1)

    def forward(self, x, pass_input=False):
        if self.pointwise:
            return SF.conv1x1(x, self.weight, self.bias, pass_input=pass_input)       # libsegx MFMA GEMM
        if self.groups == self.in_channels and self.groups == self.out_channels and self.bias is None:
            return SF.dwconv2d(x, self.weight, self.stride[0], self.static_pad)   # libsegx depthwise stencil
        assert self.groups == 1 and self.bias is None and self.dilation == (1, 1)
        return SF.conv2d_dense(x, self.weight, self.stride[0], self.static_pad)   # the 3x3 stem: libsegx implicit GEMM


def swish(x):
    return x * torch.sigmoid(x)


def drop_connect(x, p, training):
    """utils.py:129-154 -- per-sample stochastic depth."""
    if not training:
        return x
    keep = 1 - p
    r = keep + torch.rand([x.shape[0], 1, 1, 1], dtype=x.dtype, device=x.device)
    return x / keep * torch.floor(r)


class MBConvBlock(nn.Module):
    def __init__(self, k, s, e, cin, cout, se_ratio, image_size):
        super().__init__()
        self.stride, self.expand_ratio, self.input_filters, self.output_filters = s, e, cin, cout
        oup = cin * e
        if e != 1:
            self._expand_conv = Conv2dStaticSamePadding(cin, oup, 1, image_size=image_size, bias=False)
            self._bn0 = nn.BatchNorm2d(oup, momentum=BN_MOM, eps=BN_EPS)
        self._depthwise_conv = Conv2dStaticSamePadding(oup, oup, k, stride=s, image_size=image_size, groups=oup, bias=False)
        self._bn1 = nn.BatchNorm2d(oup, momentum=BN_MOM, eps=BN_EPS)
        nsq = max(1, int(cin * se_ratio))
        self._se_reduce = Conv2dStaticSamePadding(oup, nsq, 1, image_size=1)
        self._se_expand = Conv2dStaticSamePadding(nsq, oup, 1, image_size=1)
        self._project_conv = Conv2dStaticSamePadding(oup, cout, 1, image_size=math.ceil(image_size / s), bias=False)
        self._bn2 = nn.BatchNorm2d(cout, momentum=BN_MOM, eps=BN_EPS)

    gate_in_weights = True     # False: the reference's op order (gate applied to the activation, model.py:110), as a separate pass

    def forward(self, inputs, drop_connect_rate=None):
        """One autograd node per block (SF.block_node: the ops below recorded on a tape, backward walks it in reverse) -- same kernels in the same order, less
        host time per block of the eager (data-parallel) step."""
        ps = self.__dict__.get('_block_params')
        if ps is None:
            ps = self.__dict__['_block_params'] = [p for p in self.parameters()]          # parameters are moved / loaded in place: the objects stay
        return SF.block_node(self._forward_ops, inputs, (drop_connect_rate,), ps)

    def _forward_ops(self, inputs, drop_connect_rate=None):
        x = skip_in = inputs
        skip = self.stride == 1 and self.input_filters == self.output_filters
        if self.expand_ratio != 1:
            # (`inputs` has two consumers when there is a skip connection.  Routing the second through the GEMM op's input alias -- its gradient added
            # inside the expansion's dX GEMM, SF.conv1x1(pass_input=True), what the Inception modules do -- was measured here, r04_d: the residual keeps
            # those dX GEMMs off the wave-specialised tiles, +0.40 ms/step of GEMM time against 0.24 ms of accumulation adds saved.  Not used.)
            x = SF.bn_act(self._expand_conv(x), self._bn0, SF.ACT_SWISH)
        se = (self._se_reduce.weight, self._se_reduce.bias, self._se_expand.weight, self._se_expand.bias)
        # drop_connect (utils.py:129-154, per-sample stochastic depth) and the skip add ride on the last BatchNorm pass: the sample's keep scale is
        # drawn inside the kernels from the library's Philox stream (H3: the reference's torch.rand stream cannot be matched anyway)
        rate = float(drop_connect_rate) if (skip and drop_connect_rate and self.training) else 0.0
        if MBConvBlock.gate_in_weights:
            # squeeze-excite gate folded into the projection weights (per-sample weights; the gated tensor is never written)
            y, Wb = SF.bn_act_gate_weights(self._depthwise_conv(x), self._bn1, SF.ACT_SWISH, *se, self._project_conv.weight)
            x = SF.conv1x1_per_sample(y, Wb)
        else:
            x = self._project_conv(SF.bn_act_se(self._depthwise_conv(x), self._bn1, SF.ACT_SWISH, *se))
        return SF.bn_act(x, self._bn2, SF.ACT_NONE, resid=skip_in if skip else None, drop_connect=rate)


class EfficientNet(nn.Module):
    def __init__(self, model_name='efficientnet-b4', stem_stride=2, drop_connect_rate=0.2, num_classes=1000):
        super().__init__()
        width, depth, res, _ = _PARAMS[model_name]
        self.drop_connect_rate = drop_connect_rate
        self.stem_stride = stem_stride
        c0 = round_filters(32, width)
        self._conv_stem = Conv2dStaticSamePadding(3, c0, 3, stride=stem_stride, image_size=res, bias=False)
        self._bn0 = nn.BatchNorm2d(c0, momentum=BN_MOM, eps=BN_EPS)
        size = math.ceil(res / 2)                               # N6: bookkeeping assumes a stride-2 stem
        self._blocks = nn.ModuleList([])
        self.endpoint_seg_indices = [0, 1, 2, 4]
        self.endpoint_blk_indices = []
        for i, (r, k, s, e, cin, cout, se) in enumerate(_BASE):
            cin, cout, r = round_filters(cin, width), round_filters(cout, width), round_repeats(r, depth)
            self._blocks.append(MBConvBlock(k, s, e, cin, cout, se, size))
            size = math.ceil(size / s)
            for _ in range(r - 1):
                self._blocks.append(MBConvBlock(k, 1, e, cout, cout, se, size))
            if i in self.endpoint_seg_indices:
                self.endpoint_blk_indices.append(len(self._blocks))
        chead = round_filters(1280, width)
        self._conv_head = Conv2dStaticSamePadding(cout, chead, 1, image_size=size, bias=False)
        self._bn1 = nn.BatchNorm2d(chead, momentum=BN_MOM, eps=BN_EPS)
        self._fc = nn.Linear(chead, num_classes)               # kept for checkpoint compatibility; never used (N3)

    @classmethod
    def from_name(cls, model_name, stem_stride=2, **kw):
        return cls(model_name, stem_stride=stem_stride, **kw)

    @classmethod
    def from_pretrained(cls, model_name, weights_path=None, advprop=False, ignore_missing_keys=False, stem_stride=2, **kw):
        """reference efficientnet/model.py:357-398: build the model, then load the published lukemelas checkpoint.  There is no
        network here, so the file must already be on disk: `weights_path`, or `<$SEGX_PRETRAINED_DIR>/<published file name>`
        (efficientnet/utils.py:570-596: e.g. adv-efficientnet-b4-44fb3a87.pth for advprop b4)."""
        model = cls.from_name(model_name, stem_stride=stem_stride, **kw)
        load_pretrained_weights(model, model_name, weights_path=weights_path, load_fc=True, advprop=advprop,
                                ignore_missing_keys=ignore_missing_keys)
        return model

    def extract_endpoints(self, inputs):
        endpoints = {}
        x = SF.bn_act(self._conv_stem(inputs), self._bn0, SF.ACT_SWISH)
        prev_x = x
        nblk = len(self._blocks)
        for idx, block in enumerate(self._blocks):
            rate = self.drop_connect_rate * float(idx) / nblk if self.drop_connect_rate else None
            x = block(x, drop_connect_rate=rate)
            if idx in self.endpoint_blk_indices:
                endpoints['reduction_%d' % (len(endpoints) + 1)] = prev_x
            prev_x = x
        x = SF.bn_act(self._conv_head(x), self._bn1, SF.ACT_SWISH)
        endpoints['reduction_%d' % (len(endpoints) + 1)] = x
        return endpoints


# published checkpoint file names (reference efficientnet/utils.py:570-596)
PRETRAINED_FILES = {False: {'efficientnet-b0': 'efficientnet-b0-355c32eb.pth', 'efficientnet-b1': 'efficientnet-b1-f1951068.pth',
                            'efficientnet-b2': 'efficientnet-b2-8bb594d6.pth', 'efficientnet-b3': 'efficientnet-b3-5fb5a3c3.pth',
                            'efficientnet-b4': 'efficientnet-b4-6ed6700e.pth', 'efficientnet-b5': 'efficientnet-b5-b6417697.pth',
                            'efficientnet-b6': 'efficientnet-b6-c76e70fd.pth', 'efficientnet-b7': 'efficientnet-b7-dcc49843.pth'},
                    True: {'efficientnet-b0': 'adv-efficientnet-b0-b64d5a18.pth', 'efficientnet-b1': 'adv-efficientnet-b1-0f3ce85a.pth',
                           'efficientnet-b2': 'adv-efficientnet-b2-6e9d97e5.pth', 'efficientnet-b3': 'adv-efficientnet-b3-cdd7c0f4.pth',
                           'efficientnet-b4': 'adv-efficientnet-b4-44fb3a87.pth', 'efficientnet-b5': 'adv-efficientnet-b5-86493f6b.pth',
                           'efficientnet-b6': 'adv-efficientnet-b6-ac80338e.pth', 'efficientnet-b7': 'adv-efficientnet-b7-4652b6dd.pth',
                           'efficientnet-b8': 'adv-efficientnet-b8-22a8fe65.pth'}}


def pretrained_path(model_name, advprop=False):
    import os
    d = os.environ.get('SEGX_PRETRAINED_DIR')
    if not d:
        raise RuntimeError('pretrained %s weights: no network access here -- download %s and pass weights_path=..., or put it '
                           'under $SEGX_PRETRAINED_DIR (or build with use_pretrained=False)' % (model_name, PRETRAINED_FILES[advprop][model_name]))
    path = os.path.join(d, PRETRAINED_FILES[advprop][model_name])
    if not os.path.exists(path):
        raise RuntimeError('pretrained weights not found: ' + path)
    return path


def load_pretrained_weights(model, model_name, weights_path=None, load_fc=True, advprop=False, ignore_missing_keys=False):
    """reference efficientnet/utils.py:601-636 with a local file instead of model_zoo.load_url; same key checks."""
    state_dict = torch.load(weights_path if isinstance(weights_path, str) else pretrained_path(model_name, advprop), map_location='cpu')
    if not load_fc:
        state_dict.pop('_fc.weight'); state_dict.pop('_fc.bias')
    ret = model.load_state_dict(state_dict, strict=False)
    if not ignore_missing_keys:
        expect = set() if load_fc else {'_fc.weight', '_fc.bias'}
        assert set(ret.missing_keys) == expect, 'Missing keys when loading pretrained weights: %s' % ret.missing_keys
    assert not ret.unexpected_keys, 'Unexpected keys when loading pretrained weights: %s' % ret.unexpected_keys
