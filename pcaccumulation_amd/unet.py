"""Dense 2-D encoder-decoder and the small conv / linear heads: host mirror of models/unet.py.

Attribute names fix the state_dict contract: `down_convs.{i}.{conv1,conv2}`,
`up_convs.{i}.{upconv,conv1,conv2}`, `conv_final`, `seg_head.{0,1,3}`.  On the GPU the convolutions run on the
package's own MFMA kernels (pcaccumulation_amd.ops: 3x3 / 3x3x3, the 2x2 transposed convolutions, the 32 -> 2 head) fed
channels-last tensors; the library (MIOpen through PyTorch-ROCm) is the fallback for shapes they do not take and the CPU
path.  DESIGN.md sections 3, 15, 16 and 18: why the C <= 64 full-resolution layers are HBM-bound and what is fused around them.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from . import ops


def conv3x3(in_channels, out_channels):
    return nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)


class DownConv(nn.Module):
    """models/unet.py:45-71: two 3x3 conv + ReLU, optional 2x2 max-pool; returns (pooled, before_pool)."""

    def __init__(self, in_channels, out_channels, pooling=True):
        super(DownConv, self).__init__()
        self.in_channels, self.out_channels, self.pooling = in_channels, out_channels, pooling
        self.conv1 = conv3x3(in_channels, out_channels)
        self.conv2 = conv3x3(out_channels, out_channels)
        if pooling:
            self.pool = nn.MaxPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        pair = ops.conv_pair_fusable(x, self.conv1, self.conv2)   # bf16: conv1's ReLU backward in conv2's data-gradient epilogue
        x = ops.conv3x3(x, self.conv1, relu=True, premasked=pair)
        if self.pooling and ops.conv3x3_native(x, self.conv2) in ('bf16', 'split', 'mixed') and self.out_channels % 8 == 0:
            return ops.conv3x3_relu_pool(x, self.conv2, input_relu=pair)   # second conv + ReLU + pool, their backward in one pass (csrc/pool.hip)
        x = ops.conv3x3(x, self.conv2, relu=True, input_relu=pair)
        return (ops.carry_amax(x, self.pool(x)) if self.pooling else x), x          # window maxima of x: x's bound holds (fp32x3 scales)


class UpConv(nn.Module):
    """models/unet.py:74-113: 2x2 transpose-conv upsample, concat with the skip, two 3x3 conv + ReLU."""

    def __init__(self, in_channels, out_channels, merge_mode='concat', up_mode='transpose'):
        super(UpConv, self).__init__()
        assert up_mode == 'transpose', 'only the transpose-conv decoder the released weights use'
        self.in_channels, self.out_channels, self.merge_mode, self.up_mode = in_channels, out_channels, merge_mode, up_mode
        self.upconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
        self.conv1 = conv3x3(2 * out_channels if merge_mode == 'concat' else out_channels, out_channels)
        self.conv2 = conv3x3(out_channels, out_channels)

    def forward(self, from_down, from_up):
        if self.merge_mode == 'concat' and os.environ.get('PCACC_UPCONV_CAT', '0') != '1':
            up = ops.upconv2x2(from_up, self.upconv)                  # fp32x3 / mixed: the 1-tap split kernels; bf16: csrc/upconv_bf16.hip; else the library
            probe = up.new_empty((0, self.conv1.in_channels) + tuple(up.shape[2:]))
            pair = ops.conv_pair_fusable(probe, self.conv1, self.conv2)
            x = ops.conv3x3_cat(up, from_down, self.conv1, relu=True, premasked=pair)      # mixed: conv1 reads the two fp32 maps in place, no fp32 concatenation
            return ops.conv3x3(x, self.conv2, relu=True, input_relu=pair)
        if self.merge_mode == 'concat':
            x = ops.upconv_cat(from_up, from_down, self.upconv)        # opt-in experiment: the transposed convolution writes into the concatenation buffers
        else:
            x = ops.upconv2x2(from_up, self.upconv) + from_down
        pair = ops.conv_pair_fusable(x, self.conv1, self.conv2)
        return ops.conv3x3(ops.conv3x3(x, self.conv1, relu=True, premasked=pair), self.conv2, relu=True, input_relu=pair)


class UNet(nn.Module):
    """models/unet.py:116-233 (depth 5, 32..512 channels in the default config)."""

    def __init__(self, in_channels=3, depth=5, start_filts=64, up_mode='transpose', merge_mode='concat', **kwargs):
        super(UNet, self).__init__()
        if up_mode not in ('transpose', 'upsample') or merge_mode not in ('concat', 'add'):
            raise ValueError('unsupported up_mode / merge_mode: %s / %s' % (up_mode, merge_mode))
        self.up_mode, self.merge_mode = up_mode, merge_mode
        self.in_channels, self.start_filts, self.depth = in_channels, start_filts, depth
        downs, ups = [], []
        outs = in_channels
        for i in range(depth):
            ins = in_channels if i == 0 else outs
            outs = start_filts * (2 ** i)
            downs.append(DownConv(ins, outs, pooling=i < depth - 1))
        for i in range(depth - 1):
            ins = outs
            outs = ins // 2
            ups.append(UpConv(ins, outs, up_mode=up_mode, merge_mode=merge_mode))
        self.down_convs = nn.ModuleList(downs)
        self.up_convs = nn.ModuleList(ups)
        self.conv_final = conv3x3(outs, in_channels)
        self.reset_params()

    @staticmethod
    def weight_init(m):
        if isinstance(m, nn.Conv2d):                 # ConvTranspose2d keeps its default init (unet.py:210-219)
            init.xavier_normal_(m.weight)
            init.constant_(m.bias, 0)

    def reset_params(self):
        for m in self.modules():
            self.weight_init(m)

    def forward(self, x):
        skips = []
        with ops.stage('unet_enc'):
            for module in self.down_convs:
                x, before_pool = module(x)
                skips.append(before_pool)
        with ops.stage('unet_dec'):
            for i, module in enumerate(self.up_convs):
                x = module(skips[-(i + 2)], x)
        # 'mixed' mode: the bf16 gradient graph covers the encoder / decoder body and ends here.  The last convolution and the two heads behind it
        # keep fp32 gradients and fp32x3 products: each head starts conv -> BatchNorm, whose backward returns a zero-mean gradient -- the bias
        # gradients of those first convolutions and of conv_final are sums that cancel (to rounding / to a boundary term), and formed from
        # bf16-rounded gradient maps they were 50 % - 100 x off (0.0020 against 0.0013, 5e-4 against 4e-6 on c3); a BatchNorm fed a
        # bf16-rounded gradient keeps 2^-9 of what it subtracts as noise.  Everything from here to the losses stays fp32.
        with ops.stage('unet_dec'):
            return ops.conv3x3(ops.exit_mixed(x), self.conv_final)


class SegHead1D(nn.Module):
    """models/unet.py:235-256: Linear, BatchNorm1d, ReLU, Linear on [N, C] rows."""

    def __init__(self, in_channel, out_channel, bias=True):
        super(SegHead1D, self).__init__()
        mid = max(in_channel, out_channel)
        self.seg_head = nn.Sequential(nn.Linear(in_channel, mid, bias=bias), nn.BatchNorm1d(mid), nn.ReLU(),
                                      nn.Linear(mid, out_channel, bias=bias))

    def forward(self, feats):
        return self.seg_head(feats)


class SegHead2D(nn.Module):
    """models/unet.py:259-277: conv, BatchNorm2d, ReLU, conv on [B, C, H, W]."""

    def __init__(self, in_channel, out_channel, kernel_size=3, stride=1, padding=1, bias=True, groups=1):
        super(SegHead2D, self).__init__()
        mid = max(in_channel, out_channel)
        self.seg_head = nn.Sequential(
            nn.Conv2d(in_channel, mid, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias, groups=groups),
            nn.BatchNorm2d(mid), nn.ReLU(),
            nn.Conv2d(mid, out_channel, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias, groups=groups))

    def forward(self, feats):
        conv0, bn, act, conv1 = self.seg_head
        fused = isinstance(act, nn.ReLU)                       # normalisation and ReLU in one pass each way (csrc/bn.hip)
        h = self.hidden(feats)
        return ops.conv3x3(h, conv1)                           # c_out = 2: the streamed head kernels (csrc/head_conv.hip)

    def hidden(self, feats):
        """relu(bn(conv0(feats))): everything in front of the head's last convolution."""
        conv0, bn, act, conv1 = self.seg_head
        fused = isinstance(act, nn.ReLU)
        h = ops.batch_norm_nchw(ops.exit_mixed(ops.conv3x3(feats, conv0)), bn, relu=fused)     # (mixed mode: fp32 here already, see UNet.forward)
        return h if fused else act(h)

    def sparse_rows(self, feats):
        """The head's output as a table of rows evaluated on demand (ops.SparseConvRows): for a consumer that reads a few thousand cells of the map
        (the ego head's key points), when the last convolution qualifies; None otherwise."""
        h = self.hidden(feats)
        conv1 = self.seg_head[3]
        return ops.SparseConvRows(h, conv1) if ops.sparse_conv_available(h, conv1) else ops.nchw_as_rows(ops.exit_mixed(ops.conv3x3(h, conv1)))
