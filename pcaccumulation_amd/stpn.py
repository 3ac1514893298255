"""Spatio-temporal motion head: host mirror of models/stpn.py (state_dict keys `init_conv.{0,2,4,6}`,
`down_convs`, `up_convs`, `positional_encoding.{0,2}`, `final_proj.0`, `mos_seg`, `offset_head`)."""
import torch
import torch.nn as nn

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
        offset = torch.where(torch.isnan(offset), torch.zeros_like(offset), offset)
        offset = torch.where(torch.isinf(offset), torch.zeros_like(offset), offset)
        return torch.clamp(offset, min, max)

    def backbone(self, x):
        """[B, C, T, H, W] -> [B, 64, H, W]: Conv3d stack, max over T, U-Net (models/stpn.py:82-92)."""
        x = self.init_conv(x)
        x = torch.max(x, dim=2)[0]
        skips = []
        for module in self.down_convs:
            x, before_pool = module(x)
            skips.append(before_pool)
        for i, module in enumerate(self.up_convs):
            x = module(skips[-(i + 2)], x)
        return x

    def forward(self, x, points, time_indice, pc_range):
        """x [B,C,T,H,W]; points [K,3]; time_indice [K,2] -> (mos logits [K,2], offset [K,2], map [B,64,H,W])."""
        x = self.backbone(x)
        batch_idx = time_indice[:, 0].to(torch.int32).contiguous()
        ungridded = ops.bilinear_gather(x, points, batch_idx, abs(pc_range[0]), abs(pc_range[1]))
        pos = self.positional_encoding(points / abs(pc_range[0]))
        enc = self.final_proj(torch.cat([pos, ungridded.to(pos.dtype)], dim=-1))
        classes = self.mos_seg(enc)
        offset = self.safe_guard_offset(self.offset_head(enc))
        return classes, offset, x
