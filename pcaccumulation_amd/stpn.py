"""Spatio-temporal motion head: host mirror of models/stpn.py (state_dict keys `init_conv.{0,2,4,6}`,
`down_convs`, `up_convs`, `positional_encoding.{0,2}`, `final_proj.0`, `mos_seg`, `offset_head`)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .unet import DownConv, UpConv, SegHead1D


class STPN(nn.Module):
    def __init__(self, height_feat_size=32):
        super(STPN, self).__init__()
        n_filters = [32, 64, 128, 128, 256]
        # four 3x3x3 convolutions mixing the T axis (models/stpn.py:13-22)
        layers = []
        cin = height_feat_size
        for _ in range(4):
            layers += [nn.Conv3d(cin, n_filters[0], kernel_size=(3, 3, 3), stride=(1, 1, 1), padding=(1, 1, 1)), nn.ReLU()]
            cin = n_filters[0]
        self.init_conv = nn.Sequential(*layers)
        # small U-Net (models/stpn.py:24-43): widths max(64, w)
        downs, ins = [], height_feat_size
        for idx, width in enumerate(n_filters):
            width = max(64, width)
            downs.append(DownConv(ins, width, pooling=idx < len(n_filters) - 1))
            ins = width
        ups, ins = [], n_filters[-1]
        for width in n_filters[-2::-1]:
            width = max(64, width)
            ups.append(UpConv(ins, width, merge_mode='concat'))
            ins = width
        self.down_convs = nn.ModuleList(downs)
        self.up_convs = nn.ModuleList(ups)
        self.positional_encoding = nn.Sequential(nn.Linear(3, 32, bias=True), nn.ReLU(), nn.Linear(32, 64, bias=True), nn.ReLU())
        self.final_proj = nn.Sequential(nn.Linear(128, 128, bias=True), nn.ReLU())
        self.mos_seg = SegHead1D(128, 2)
        self.offset_head = SegHead1D(128, 2)

    def safe_guard_offset(self, offset, min=-20, max=20):
        """models/stpn.py:61-65: NaN -> 0, Inf -> 0, clamp to +-20."""
        if offset.is_cuda:                                          # the same function in two passes instead of seven (NaN, +Inf, -Inf -> 0, then the clamp)
            return torch.clamp(torch.nan_to_num(offset, nan=0.0, posinf=0.0, neginf=0.0), min, max)
        offset = torch.where(torch.isnan(offset), torch.zeros_like(offset), offset)
        offset = torch.where(torch.isinf(offset), torch.zeros_like(offset), offset)
        return torch.clamp(offset, min, max)

    def temporal_convs(self, x):
        """The four Conv3d(3x3x3)+ReLU of models/stpn.py:13-22 and the max over T (:83), evaluated as 2-D
        convolutions: out[t] = sum_kt W[:, :, kt] * in[t + kt - 1], i.e. a 3x3 Conv2d over the channel-stacked
        frames (t-1, t, t+1) with the Conv3d weight viewed as [O, 3*C, 3, 3].  Same parameters (`init_conv.{0,2,4,6}`
        keep their Conv3d shapes in the state_dict), same arithmetic up to summation order; what changes is that
        the work lands on the channels-last implicit-GEMM conv kernels instead of the batched-GEMM Conv3d
        backward-weights path, which took 48 ms per layer on MI355X (profiles/r01_*)."""
        B, C, T, H, W = x.shape
        rows = ops.carry_amax(x, x.permute(0, 2, 3, 4, 1).contiguous())            # [B,T,H,W,C]; free for warp output
        convs = [l for l in self.init_conv if isinstance(l, nn.Conv3d)]
        native = [ops.conv3x3_available(torch.empty((0, l.in_channels, H, W), dtype=rows.dtype, device=rows.device), l.weight)
                  if l.kernel_size == (3, 3, 3) and l.padding == (1, 1, 1) else None for l in convs]
        # bf16 / fp32x3 chain: a layer's ReLU backward rides in the NEXT layer's data-gradient epilogue (ops.conv_pair_fusable's scheme along the chain)
        chain = (all(n == 'bf16' for n in native) or all(n == 'split' for n in native) or all(n == 'mixed' for n in native)) and os.environ.get('PCACC_CONV_PAIR', '1') != '0'
        for k, layer in enumerate(convs):
            cin = layer.in_channels
            if native[k]:
                # bf16 / fp32x3 on the GPU: the MFMA kernels read frames t-1, t, t+1 in place (no channel-stacked copy)
                y = ops.conv3x3_rows(ops.carry_amax(rows, rows.view(B * T, H, W, cin)), layer.weight, layer.bias, frames=T, relu=True,
                                     premasked=chain and k + 1 < len(convs), input_relu=chain and k > 0)
                rows = ops.carry_amax(y, y.view(B, T, H, W, layer.out_channels))
                continue
            stacked = ops._stack_frames(rows.view(B * T, H, W, cin), T)                # [B*T,H,W,3C]
            w2 = layer.weight.permute(0, 2, 1, 3, 4).reshape(layer.out_channels, 3 * cin, 3, 3)
            y = F.relu(F.conv2d(stacked.permute(0, 3, 1, 2), w2, layer.bias, padding=1))
            rows = y.permute(0, 2, 3, 1).contiguous().view(B, T, H, W, layer.out_channels)
        out = ops.carry_amax(rows, ops.frames_max(rows))                               # maxima over frames: the stack's bound holds
        return ops.carry_amax(out, out.permute(0, 3, 1, 2))                            # [B,C,H,W], channels_last

    def backbone(self, x):
        """[B, C, T, H, W] -> [B, 64, H, W]: temporal conv stack, max over T, U-Net (models/stpn.py:82-92)."""
        with ops.stage('stpn_temporal'):
            x = self.temporal_convs(x)
        skips = []
        with ops.stage('stpn_unet'):
            for module in self.down_convs:
                x, before_pool = module(x)
                skips.append(before_pool)
            for i, module in enumerate(self.up_convs):
                x = module(skips[-(i + 2)], x)
        return x

    @staticmethod
    def point_mlp(seq, x):
        """nn.Sequential of (Linear, ReLU)* evaluated with the fused row-linear kernels (ReLU folded into the store)."""
        mods = list(seq)
        i = 0
        pd = ops.point_dtype() if x.is_cuda else x.dtype          # bf16 rows in the bf16 compute mode (GPU only)
        # 'mixed' mode: an fp32 input without a shadow (the 3-feature positions) starts a chain of bf16 shadows of fp32 rows
        head = ops.mixed_mode() and x.is_cuda and x.dtype == torch.float32 and x.shape[0] >= ops.MIN_ROWS_FUSED_LINEAR
        while i < len(mods):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = ops.linear_rows(x, mods[i], post_relu=relu, out_dtype=pd, mixed=head and i == 0)
            i += 2 if relu else 1
        return x

    @staticmethod
    def point_head(head, x):
        """SegHead1D = Linear, BatchNorm1d, ReLU, Linear (models/unet.py:240-245): the two Linear layers are fused row
        kernels, BatchNorm1d (batch statistics over the K points in train mode, trap 16) four streaming passes (ops.batch_norm_rows)."""
        lin0, bn, _, lin1 = head.seg_head
        x = ops.exit_mixed(x)             # 'mixed' mode: from the losses back to (and including) the layer in front of a BatchNorm everything is fp32 (unet.UNet.forward)
        return ops.linear_rows(ops.batch_norm_rows(ops.linear_rows(x, lin0), bn), lin1, pre_relu=True, out_dtype=torch.float32)

    def forward(self, x, points, time_indice, pc_range):
        """x [B,C,T,H,W]; points [K,3]; time_indice [K,2] -> (mos logits [K,2], offset [K,2], map [B,64,H,W])."""
        x = self.backbone(x)
        batch_idx = time_indice[:, 0].to(torch.int32).contiguous()
        ungridded = ops.bilinear_gather(x, points, batch_idx, abs(pc_range[0]), abs(pc_range[1]))
        pos = self.point_mlp(self.positional_encoding, points / abs(pc_range[0]))
        enc = self.point_mlp(self.final_proj, ops.cat_rows(pos, ungridded))
        classes = self.point_head(self.mos_seg, enc)
        offset = self.safe_guard_offset(self.point_head(self.offset_head, enc))
        return classes, offset, x
