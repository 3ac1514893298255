"""Pillar encoder and pillar <-> BEV movement: host mirror of models/pillar_encoder.py.

Same class / function names, constructor arguments and state_dict keys as the reference
(`fc_pos`, `fc_c`, `blocks.{i}.{fc_0,fc_1,shortcut}`); the per-pillar pooling,
scatter and bilinear gather run in the HIP kernels behind pcaccumulation_amd.ops.
"""
import torch
from torch import nn

from . import native, ops
from .ops import PillarIndex


class ResnetBlockFC(nn.Module):
    """models/pillar_encoder.py:13-55."""

    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        if size_out is None:
            size_out = size_in
        if size_h is None:
            size_h = min(size_in, size_out)
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.actvn = nn.ReLU()
        self.shortcut = None if size_in == size_out else nn.Linear(size_in, size_out, bias=False)
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x):
        # fc_1(relu(fc_0(relu(x)))) + shortcut(x): one kernel for the encoder's block shape on bf16 rows (csrc/pfn_block.hip), else
        # three fused row-linear launches (ReLU on load, residual on store); rows keep x's element type
        if ops.pfn_block_available(self, x):
            return ops.pfn_block(self, x)
        net = ops.linear_rows(x, self.fc_0, pre_relu=True)
        x_s = ops.linear_rows(x, self.shortcut) if self.shortcut is not None else x
        return ops.linear_rows(net, self.fc_1, pre_relu=True, residual=x_s)

    def forward_pooled(self, x, pooled, pidx):
        """forward(cat(x, pooled[point_to_voxel_map])) (models/pillar_encoder.py:116-118) with the gather and the concatenation
        folded into the two layers that read them (bf16 rows on the GPU); the same arithmetic otherwise."""
        if ops.pfn_block_available(self, x, pooled):
            return ops.pfn_block(self, x, pooled, pidx)
        if self.shortcut is None or not ops.linear_rows_cat_available(x, pooled, self.fc_0):
            return self.forward(torch.cat([x, ops.broadcast_to_points(pooled, pidx)], dim=1))
        net = ops.linear_rows_cat(x, pooled, pidx, self.fc_0, pre_relu=True)
        x_s = ops.linear_rows_cat(x, pooled, pidx, self.shortcut)
        return ops.linear_rows(net, self.fc_1, pre_relu=True, residual=x_s)


class PillarFeatureNet(nn.Module):
    """models/pillar_encoder.py:59-122."""

    def __init__(self, cfg):
        super(PillarFeatureNet, self).__init__()
        num_input_features = cfg['num_input_features']
        num_filters = cfg['num_filters']
        voxel_size = cfg['voxel_size']
        pc_range = cfg['pc_range']
        depth = cfg['depth']
        self.scale = abs(pc_range[0])
        self.n_frames = cfg['n_sweeps']
        self.name = "PillarFeatureNet"
        self.fc_pos = nn.Linear(num_input_features, 2 * num_filters)
        self.fc_c = nn.Linear(num_filters, num_filters)
        self.actvn = nn.ReLU()
        self.blocks = nn.ModuleList([ResnetBlockFC(2 * num_filters, num_filters) for _ in range(depth)])
        self.vx = voxel_size[0]
        self.vy = voxel_size[1]
        self.x_offset = self.vx / 2 + pc_range[0]
        self.y_offset = self.vy / 2 + pc_range[1]

    def point_features(self, raw_points, pidx, coordinates, pillar_mean, time_indice, pillar_major=False):
        """The 9 inputs of pillar_encoder.py:98-110, in the same arithmetic (f64 coordinates, f32 points).  pillar_major: rows in the order of pidx.order
        (forward() then takes pidx.pillar_major() as its index)."""
        coords = coordinates.contiguous()
        if coords.dtype not in (torch.float64, torch.int32):
            coords = coords.to(torch.float64)
        ti = time_indice.contiguous()
        if ti.dtype != torch.float64:
            ti = ti.to(torch.float64)
        return native.pfn_features(raw_points.contiguous(), pidx.p2v, pillar_mean.contiguous(), coords, ti, float(self.vx),
                                   float(self.vy), float(self.x_offset), float(self.y_offset), float(self.scale),
                                   float(self.n_frames), **({'order': pidx.order} if pillar_major else {}))

    def forward(self, raw_points, point_to_voxel_map, coordinates, pillar_mean, time_indice, pidx=None, keep_dtype=False, features=None, canvas=False):
        """keep_dtype: return the pooled rows in the element type of the point rows (bf16 in the bf16 compute mode; MotionNet feeds
        them to the canvas fill as they are) instead of the reference's float32.  features: the 9 inputs per point when the caller
        built them ahead of time (point_features depends on the batch only, MotionNet.prepare_inputs).  canvas=True (MotionNet): the result is the pair
        (tensor, is_canvas) -- where the last pooling can write the BEV canvas itself (ops.segment_max_canvas: 'mixed' mode) the tensor is the [n_cells, C]
        canvas and no pooled-row table exists; otherwise the pooled rows as always."""
        if pidx is None:                                                  # reference call signature
            pidx = PillarIndex.from_point_map(point_to_voxel_map, coordinates.shape[0])
        if features is None:
            features = self.point_features(raw_points, pidx, coordinates, pillar_mean, time_indice)
        pd = ops.point_dtype() if features.is_cuda else features.dtype     # bf16 rows in the bf16 compute mode (GPU only)
        net = ops.linear_rows(features, self.fc_pos, out_dtype=pd, mixed=ops.mixed_mode())     # 'mixed' mode: the chain continues as bf16 shadows of fp32 rows
        net = self.blocks[0](net)
        for block in self.blocks[1:]:
            if ops.pfn_pool_block_available(block, net, pidx):
                net = ops.pfn_block(block, net, None, pidx, pool=True)       # max-pool, broadcast, concatenation and block: one autograd node
            else:
                net = block.forward_pooled(net, ops.carry_amax(net, ops.segment_max(net, pidx)), pidx)      # maxima of net's rows: net's bound holds
        feats = ops.linear_rows(net, self.fc_c)
        if canvas and ops.segment_max_canvas_available(feats, pidx):
            return ops.segment_max_canvas(feats, pidx), True
        pooled = ops.carry_amax(feats, ops.segment_max(feats, pidx))
        pooled = pooled if keep_dtype else pooled.float()
        return (pooled, False) if canvas else pooled


def _index_for(coords, batch_size, input_shape):
    return PillarIndex(coords, None, int(batch_size), [int(v) for v in input_shape])


def scatter_point_pillar(voxel_features, coords, batch_size, input_shape, pidx=None):
    """models/pillar_encoder.py:125-174.  Returns logical [B, C, nt, ny, nx] like the reference; the
    memory behind it is channels-last ([B, nt, ny, nx, C])."""
    if pidx is None:
        pidx = _index_for(coords, batch_size, input_shape)
    src = voxel_features
    out_dtype = src.dtype if src.dtype in (torch.float32, torch.bfloat16) else torch.float32
    canvas = ops.pillar_scatter(src, pidx, out_dtype)
    out = ops.canvas_as_reference(canvas, pidx)
    return out if src.dtype == out.dtype else out.to(src.dtype)


def inverse_scatter_point_pillar(voxel_features, coords, batch_size, input_shape, pidx=None):
    """models/pillar_encoder.py:177-204: [B, C, nt, ny, nx] -> [M, C] (rows in pillar order)."""
    if pidx is None:
        pidx = _index_for(coords, batch_size, input_shape)
    b, c = voxel_features.shape[0], voxel_features.shape[1]
    rows = voxel_features.permute(0, 2, 3, 4, 1).contiguous().view(-1, c)
    if rows.element_size() * c % 4:
        raise ValueError('inverse_scatter_point_pillar: row size must be a multiple of 4 bytes')
    return ops.gather_rows(rows, pidx.cell)


def _batch_index(time_indice):
    return time_indice[:, 0].to(torch.int32).contiguous()


def ungrid(feats, points, pc_range, time_indice):
    """models/pillar_encoder.py:231-267.  feats [B,C,H,W], points [N,3], time_indice [N,2] -> [N,C].
    Like the reference it normalises points[:, :2] IN PLACE (callers pass clones; appendix C trap 2).
    Output rows follow the input order, which equals the reference's batch-grouped order because the
    batch column of a collated batch is sorted (trap 3)."""
    out = ops.bilinear_gather(feats, points.detach().clone(), _batch_index(time_indice), abs(pc_range[0]), abs(pc_range[1]))
    points[:, 0] = points[:, 0] / abs(pc_range[0])
    points[:, 1] = points[:, 1] / abs(pc_range[1])
    return out


def temporal_ungrid(feats, points, pc_range, time_indice):
    """models/pillar_encoder.py:206-228.  feats [B,T,C,H,W] -> [N,C]; one launch over all frames."""
    b, t, c, h, w = feats.shape
    idx = (time_indice[:, 0] * t + time_indice[:, 1]).to(torch.int32).contiguous()
    return ops.bilinear_gather(feats.reshape(b * t, c, h, w), points, idx, abs(pc_range[0]), abs(pc_range[1]))
