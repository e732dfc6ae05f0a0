"""Domain discriminator + gradient reversal of the few-shot domain-adaptation recipe (mirror of reference code/networks/discriminator.py:14-86
and networks/revgrad.py; used by train2d.py --adv feat|mask, :876-926, 1259-1284).  Same constructor, `model.<i>` / `tail.<i>` state_dict keys
as the reference's nn.Sequential (the GradientReversal layer sits at index 0 when do_revgrad, shifting every index by one, as there).  All
arithmetic on libsegx: the five 4 x 4 stride-2 convolutions on the implicit-GEMM tile engine (depth-1 volumes), BatchNorm + LeakyReLU(0.2) as one
pass (`act` 3 of segx_bn_act_*), the global average pool as a plane reduction."""
import torch
import torch.nn as nn

from .. import functional as SF


class _RevGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = float(alpha)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -g * ctx.alpha, None


class GradientReversal(nn.Module):
    """identity forward, gradient * (-alpha) backward (revgrad.py:6-33)"""

    def __init__(self, alpha=1.0):
        super().__init__()
        self._alpha = float(alpha)

    def forward(self, x):
        return _RevGrad.apply(x, self._alpha)


class Discriminator(nn.Module):
    def __init__(self, num_in_chan, num_classes=2, do_avgpool=True, do_revgrad=True, num_base_chan=32):
        super().__init__()
        self.num_in_chan, self.num_base_chan, self.num_classes = num_in_chan, num_base_chan, num_classes
        self.use_BN, self.use_LeakyReLU, self.do_revgrad, self.do_avgpool = True, True, do_revgrad, do_avgpool
        c = num_base_chan
        layers = []
        for cin, cout, last in ((num_in_chan, c, False), (c, 2 * c, False), (2 * c, 4 * c, False), (4 * c, 8 * c, False), (8 * c, num_classes, True)):
            layers.append(nn.Conv2d(cin, cout, kernel_size=4, stride=2, padding=1, bias=False))
            if not last:
                layers += [nn.BatchNorm2d(cout), nn.LeakyReLU(0.2)]          # parameter containers at the reference's indices; fused below
        if do_revgrad:
            layers.insert(0, GradientReversal())
        self.model = nn.Sequential(*layers)
        self.tail = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten()) if do_avgpool else None

    def forward(self, x):
        mods = list(self.model)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, GradientReversal):
                x = m(x); i += 1
            elif isinstance(m, nn.Conv2d):
                x = SF.conv2d_dense(x, m.weight, 2, (1, 1, 1, 1))
                if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
                    x = SF.bn_act(x, mods[i + 1], SF.ACT_LEAKY); i += 3
                else:
                    i += 1
            else:
                raise RuntimeError('unexpected layer %r' % m)
        if self.tail is None:
            # the reference builds nn.Linear(h * w, num_classes) on the fly here (discriminator.py:70-83: a fresh, untrained layer per
            # instance whose input width depends on the image size); only the do_avgpool=True form (what train2d.py constructs) is mirrored
            raise NotImplementedError('Discriminator(do_avgpool=False)')
        B, C = x.shape[:2]
        return x.reshape(B, C, -1).mean(dim=2)                 # AdaptiveAvgPool2d(1) + Flatten on a [B, classes, h, w] map
