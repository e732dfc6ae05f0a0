"""Inception-v1 I3D feature extractor -- host-side mirror of /root/reference/code/networks/aj_i3d/aj_i3d.py.

Same module / parameter names (`Conv3d_1a_7x7.conv3d.weight`, `Mixed_4b.b1b.bn.running_mean`, ...), the
dynamic TF-'same' zero padding of N7 (front = pad // 2; aj_i3d.py:8-30, 68-90) and `do_pool1=False`
(`MaxPool3d_2a_3x3` = Identity, :206-210).

Kernel status: everything runs on libsegx -- the 38 1x1x1 convolutions on the MFMA GEMM, the 7x7x7 stem and the 19
3x3x3 convolutions as implicit GEMM on the same MFMA engine (conv3d.hip: forward, backward-data, backward-weight),
BatchNorm3d+ReLU as one fused kernel (backbone.hip), the 'same' max-pools in conv3d.hip; the backward-data of the stride-2 stem
(a transposed convolution onto 3 channels) is the residue-class gather kernel of conv3d.hip -- and is not needed at all when Segtran3d
composes its input bridge into the stem filters (the default).  No ATen / MIOpen / rocBLAS arithmetic anywhere.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as SF


class MaxPool3dSamePadding(nn.MaxPool3d):
    def forward(self, x, pass_input=False):
        return SF.maxpool3d_same(x, self.kernel_size, self.stride, pass_input)  # libsegx: zero 'same' padding + max-pool (pass_input: -> (y, alias of x))


class Unit3D(nn.Module):
    def __init__(self, in_channels, output_channels, kernel_shape=(1, 1, 1), stride=(1, 1, 1), activation_fn=F.relu,
                 use_batch_norm=True, use_bias=False, name='unit_3d'):
        super().__init__()
        self._kernel_shape, self._stride = tuple(kernel_shape), tuple(stride)
        self._activation_fn, self._use_batch_norm = activation_fn, use_batch_norm
        self.name = name
        self.conv3d = nn.Conv3d(in_channels, output_channels, self._kernel_shape, self._stride, padding=0, bias=use_bias)
        if use_batch_norm:
            self.bn = nn.BatchNorm3d(output_channels, eps=0.001, momentum=0.01)
        self.pointwise = self._kernel_shape == (1, 1, 1) and self._stride == (1, 1, 1)

    def forward(self, x, conv_out=None):
        """conv_out: the convolution's output computed by the caller (Segtran3d composes its input bridge into the stem's filters)."""
        if conv_out is not None:
            x = conv_out
        elif self.pointwise:
            x = SF.conv1x1(x, self.conv3d.weight, self.conv3d.bias)            # libsegx MFMA GEMM
        else:
            assert self.conv3d.bias is None
            x = SF.conv3d_same(x, self.conv3d.weight, self._stride)              # libsegx implicit-GEMM MFMA convolution
        if self._use_batch_norm:                                               # fused BatchNorm3d (+ReLU), libsegx
            return SF.bn_act(x, self.bn, SF.ACT_RELU if self._activation_fn is F.relu else SF.ACT_NONE)
        if self._activation_fn is not None:
            x = self._activation_fn(x)
        return x


class InceptionModule(nn.Module):
    def __init__(self, in_channels, out_channels, name):
        super().__init__()
        o = out_channels
        self.b0 = Unit3D(in_channels, o[0], name=name + '/Branch_0/Conv3d_0a_1x1')
        self.b1a = Unit3D(in_channels, o[1], name=name + '/Branch_1/Conv3d_0a_1x1')
        self.b1b = Unit3D(o[1], o[2], (3, 3, 3), name=name + '/Branch_1/Conv3d_0b_3x3')
        self.b2a = Unit3D(in_channels, o[3], name=name + '/Branch_2/Conv3d_0a_1x1')
        self.b2b = Unit3D(o[3], o[4], (3, 3, 3), name=name + '/Branch_2/Conv3d_0b_3x3')
        self.b3a = MaxPool3dSamePadding(kernel_size=(3, 3, 3), stride=(1, 1, 1), padding=0)
        self.b3b = Unit3D(in_channels, o[5], name=name + '/Branch_3/Conv3d_0b_1x1')
        self.name = name

    fuse_reductions = True     # False: the reference's op order (four independent branches)
    cat_in_place = True        # False: torch.cat assembles the module's output (tests: the two forms agree bit for bit)

    def forward(self, x):
        if InceptionModule.fuse_reductions:
            # The two 1x1x1 reductions b1a | b2a (16 .. 192 filters each: skinny GEMMs on their own) as ONE pointwise convolution with the
            # concatenated filters and ONE BatchNorm + ReLU over the concatenated channels (per-channel: exactly the two layers; one synchronised
            # exchange instead of two), the 3x3x3 convolutions reading their channel slices of that tensor in place (SF.conv3d_slices): x is read
            # once instead of twice, its gradient arrives as one tensor instead of two.
            # r03: branch 0's own 1x1x1 convolution joins them (its filters LAST, so the slices of the 3x3x3 convolutions still start at channel 0):
            # the three pointwise convolutions of the module's input are ONE GEMM, their three BatchNorm layers one statistics / apply pass and one
            # synchronised exchange; branch 0's output is the tail slice of that tensor.
            # r04: no accumulation kernels of autograd's around the module -- (a) x has two consumers (this convolution, the pooling branch): the second
            # goes through the alias the GEMM op returns, its gradient is added inside the dX GEMM; (b) branch 0's slice of t leaves through
            # conv3d_slices (its gradient is written into t's gradient there); (c) the concatenation's gradient is read IN PLACE by the BatchNorm
            # backward kernels of the four branches (channel slices with a batch stride: no contiguous copies).
            w120 = torch.cat([self.b1a.conv3d.weight, self.b2a.conv3d.weight, self.b0.conv3d.weight], dim=0)
            if InceptionModule.cat_in_place and self.b3b.conv3d.bias is None:
                # r06: the module as ONE autograd node (SF.block_node: the ops recorded on a tape, backward walks it in reverse; same kernels, same order).  The
                # concatenations of the fused layers' parameters are ATen work and stay outside the node.
                bns = (self.b1a.bn, self.b2a.bn, self.b0.bn)
                bw, bb = torch.cat([m.weight for m in bns]), torch.cat([m.bias for m in bns])
                ps = [w120, bw, bb, self.b1b.conv3d.weight, self.b2b.conv3d.weight, self.b3b.conv3d.weight, self.b1b.bn.weight, self.b1b.bn.bias,
                      self.b2b.bn.weight, self.b2b.bn.bias, self.b3b.bn.weight, self.b3b.bn.bias]
                return SF.block_node(self._fused_ops, x, (w120, bw, bb), ps)
            t, x2 = SF.conv1x1(x, w120, pass_input=True)
            t = SF.bn_act_multi(t, [self.b1a.bn, self.b2a.bn, self.b0.bn], SF.ACT_RELU)
            y1, y2, y0 = SF.conv3d_slices(t, self.b1b.conv3d.weight, self.b2b.conv3d.weight, with_tail=True)
            if InceptionModule.cat_in_place:
                # r05: the three BatchNorm + ReLU passes that end branches 1..3 write their channels straight into the concatenation (SF.bn_act_cat): torch.cat
                # copied every branch once more (four strided copy kernels per module, 0.6-0.8 ms of a cfg4 / cfg5 step); only branch 0's slice is still copied
                y3 = SF.conv1x1(self.b3a(x2), self.b3b.conv3d.weight, self.b3b.conv3d.bias)
                return SF.bn_act_cat(y0, [(y1, self.b1b.bn), (y2, self.b2b.bn), (y3, self.b3b.bn)], SF.ACT_RELU)
            return torch.cat([y0, SF.bn_act(y1, self.b1b.bn, SF.ACT_RELU), SF.bn_act(y2, self.b2b.bn, SF.ACT_RELU), self.b3b(self.b3a(x2))], dim=1)
        return torch.cat([self.b0(x), self.b1b(self.b1a(x)), self.b2b(self.b2a(x)), self.b3b(self.b3a(x))], dim=1)


    def _fused_ops(self, x, w120, bw, bb):
        """the fused form of forward() with the parameter concatenations handed in (see there for what each step replaces)"""
        t, x2 = SF.conv1x1(x, w120, pass_input=True)
        t = SF.bn_act_multi(t, [self.b1a.bn, self.b2a.bn, self.b0.bn], SF.ACT_RELU, wb=(bw, bb))
        y1, y2, y0 = SF.conv3d_slices(t, self.b1b.conv3d.weight, self.b2b.conv3d.weight, with_tail=True)
        y3 = SF.conv1x1(self.b3a(x2), self.b3b.conv3d.weight, None)
        return SF.bn_act_cat(y0, [(y1, self.b1b.bn), (y2, self.b2b.bn), (y3, self.b3b.bn)], SF.ACT_RELU)


class InceptionI3d(nn.Module):
    VALID_ENDPOINTS = ('Conv3d_1a_7x7', 'MaxPool3d_2a_3x3', 'Conv3d_2b_1x1', 'Conv3d_2c_3x3', 'MaxPool3d_3a_3x3',
                       'Mixed_3b', 'Mixed_3c', 'MaxPool3d_4a_3x3', 'Mixed_4b', 'Mixed_4c', 'Mixed_4d', 'Mixed_4e',
                       'Mixed_4f', 'MaxPool3d_5a_2x2', 'Mixed_5b', 'Mixed_5c', 'Logits', 'Predictions')

    def __init__(self, num_classes=400, in_channels=3, do_pool1=True, name='inception_i3d'):
        super().__init__()
        ep = {}
        ep['Conv3d_1a_7x7'] = Unit3D(in_channels, 64, (7, 7, 7), (2, 2, 2), name=name + 'Conv3d_1a_7x7')
        ep['MaxPool3d_2a_3x3'] = MaxPool3dSamePadding((1, 3, 3), (1, 2, 2), padding=0) if do_pool1 else nn.Identity()
        ep['Conv3d_2b_1x1'] = Unit3D(64, 64, name=name + 'Conv3d_2b_1x1')
        ep['Conv3d_2c_3x3'] = Unit3D(64, 192, (3, 3, 3), name=name + 'Conv3d_2c_3x3')
        ep['MaxPool3d_3a_3x3'] = MaxPool3dSamePadding((1, 3, 3), (1, 2, 2), padding=0)
        ep['Mixed_3b'] = InceptionModule(192, [64, 96, 128, 16, 32, 32], name + 'Mixed_3b')
        ep['Mixed_3c'] = InceptionModule(256, [128, 128, 192, 32, 96, 64], name + 'Mixed_3c')
        ep['MaxPool3d_4a_3x3'] = MaxPool3dSamePadding((3, 3, 3), (2, 2, 2), padding=0)
        ep['Mixed_4b'] = InceptionModule(480, [192, 96, 208, 16, 48, 64], name + 'Mixed_4b')
        ep['Mixed_4c'] = InceptionModule(512, [160, 112, 224, 24, 64, 64], name + 'Mixed_4c')
        ep['Mixed_4d'] = InceptionModule(512, [128, 128, 256, 24, 64, 64], name + 'Mixed_4d')
        ep['Mixed_4e'] = InceptionModule(512, [112, 144, 288, 32, 64, 64], name + 'Mixed_4e')
        ep['Mixed_4f'] = InceptionModule(528, [256, 160, 320, 32, 128, 128], name + 'Mixed_4f')
        ep['MaxPool3d_5a_2x2'] = MaxPool3dSamePadding((2, 2, 2), (2, 2, 2), padding=0)
        ep['Mixed_5b'] = InceptionModule(832, [256, 160, 320, 32, 128, 128], name + 'Mixed_5b')
        ep['Mixed_5c'] = InceptionModule(832, [384, 192, 384, 48, 128, 128], name + 'Mixed_5c')
        self.end_points = ep
        # classification head: parameters kept for checkpoint compatibility (never used on the segtran path);
        # registered before the endpoints, as in the reference (aj_i3d.py:279-286), so parameter order matches.
        self.logits = Unit3D(1024, num_classes, activation_fn=None, use_batch_norm=False, use_bias=True, name='logits')
        for k, m in ep.items():
            self.add_module(k, m)

    def extract_features(self, x, stem_conv_out=None):
        """stem_conv_out: output of the (bias-free) stem convolution computed by the caller; `x` is then not read."""
        feat, prev = {}, None
        for name in self.VALID_ENDPOINTS:
            if name in self.end_points:
                m = self._modules[name]
                if name == 'Conv3d_1a_7x7' and stem_conv_out is not None:
                    x = m(None, conv_out=stem_conv_out)
                elif isinstance(m, MaxPool3dSamePadding) and prev in self.pyramid_endpoints and tuple(m.stride) != (1, 1, 1):
                    # the pooled tensor is also an endpoint the feature pyramid reads (segtran3d.py:436-441): the pyramid gets an alias, so that its gradient
                    # reaches the pool's backward kernel and is added there (SF._MaxPool3d) -- no accumulation kernel over two full-size tensors
                    x, feat[prev] = m(x, pass_input=True)
                else:
                    x = m(x)
                feat[name] = x
                prev = name
        return feat

    pyramid_endpoints = ('Conv3d_2c_3x3', 'Mixed_3c', 'Mixed_4f')      # feats[1..3] of Segtran3d (each is followed by a strided pool)
