"""U-Net host of the Polyformer layer (mirror of reference code/networks/unet2d/unet_model.py:8-55 and unet_parts.py:9-78):
the encoder / decoder the few-shot domain-adaptation recipe inserts `Polyformer(feat_dim=64)` into, right before the 1x1 class projection.

Same constructor, attributes and `state_dict` keys as the reference (`inc.double_conv.0.weight`, `down1.maxpool_conv.1.double_conv.4.running_var`,
`up1.conv.double_conv.0.bias`, `outc.conv.weight`, `polyformer.polyformer_layers.0...`), so a reference checkpoint loads as is.  All arithmetic
runs on libsegx: the 3x3 convolutions on the implicit-GEMM tile engine (depth-1 volumes), BatchNorm + ReLU as one pass, the 2x2 max-pool on the
pooling kernels, the x2 bilinear up-sampling with align_corners=True on the one-axis resampling kernels, the class projection as a pointwise GEMM.
torch.cat / F.pad only move memory.  bilinear=False: the 2 x 2 stride-2 transposed convolutions run as pointwise GEMMs + a re-arrangement pass."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as SF
from .polyformer import Polyformer


class DoubleConv(nn.Module):
    """(conv3x3 + bias -> BatchNorm -> ReLU) twice (unet_parts.py:9-26).  `double_conv` holds the parameters at the reference's indices
    (0, 1, 3, 4); positions 2 and 5 (ReLU there) are fused into the BatchNorm pass here."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        mid = mid_channels or out_channels
        self.double_conv = nn.Sequential(nn.Conv2d(in_channels, mid, kernel_size=3, padding=1), nn.BatchNorm2d(mid), nn.Identity(),
                                         nn.Conv2d(mid, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.Identity())

    def forward(self, x):
        for i in (0, 3):
            conv, bn = self.double_conv[i], self.double_conv[i + 1]
            x = SF.bn_act(SF.conv2d_bias(x, conv.weight, conv.bias, pad=1), bn, SF.ACT_RELU)
        return x


class Down(nn.Module):
    """MaxPool2d(2) then DoubleConv (unet_parts.py:29-41); the convolutions sit at index 1 of `maxpool_conv` as in the reference."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.Identity(), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.maxpool_conv[1](SF.maxpool2d(x, 2))


class Up(nn.Module):
    """x2 up-sampling of the deeper map -- bilinear (align_corners=True) or, with bilinear=False, nn.ConvTranspose2d(k 2, s 2) --, zero-pad to
    the skip's size, concat [skip, up], DoubleConv (unet_parts.py:44-70)."""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        self.bilinear = bilinear
        if bilinear:
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:                                      # unet_parts.py:52-54: transposed convolution (same `up.weight` / `up.bias` keys), full-width DoubleConv
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        if self.bilinear:
            x1 = SF.interp_linear(x1, (2 * x1.shape[2], 2 * x1.shape[3]), align_corners=True)
        else:
            x1 = SF.conv_transpose2x2(x1, self.up.weight, self.up.bias)
        dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
        if dy or dx:
            x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([x2, x1], dim=1))


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        return SF.conv1x1(x, self.conv.weight, self.conv.bias)


class UNet(nn.Module):
    def __init__(self, n_channels, num_classes, bilinear=True, polyformer_args=None):
        super().__init__()
        self.n_channels, self.num_classes, self.bilinear = n_channels, num_classes, bilinear
        factor = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels, 64)
        self.down1, self.down2, self.down3 = Down(64, 128), Down(128, 256), Down(256, 512)
        self.down4 = Down(512, 1024 // factor)
        self.up1, self.up2 = Up(1024, 512 // factor, bilinear), Up(512, 256 // factor, bilinear)
        self.up3, self.up4 = Up(256, 128 // factor, bilinear), Up(128, 64, bilinear)
        self.outc = OutConv(64, num_classes)
        self.use_polyformer = polyformer_args is not None and polyformer_args.polyformer_mode is not None
        if self.use_polyformer:
            self.polyformer = Polyformer(feat_dim=64, args=polyformer_args)
        self.num_vis_layers = 3 + self.use_polyformer

    def forward(self, x):
        SF.defer_bn_ticks()
        try:
            return self._forward(x)
        finally:
            SF.flush_bn_ticks()                   # one multi-tensor `num_batches_tracked += 1` for the 18 BatchNorm layers

    def _forward(self, x):
        self.feature_maps = []
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        self.feature_maps.append(x5)
        y = self.up2(self.up1(x5, x4), x3)
        self.feature_maps.append(y)
        y = self.up4(self.up3(y, x2), x1)              # up3: 128 + 128 channels in
        self.feature_maps.append(y)
        if self.use_polyformer:
            y = self.polyformer(y)
            self.feature_maps.append(y)
        return self.outc(y)
