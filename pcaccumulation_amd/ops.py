"""Torch-level operators of the hot path: thin autograd wrappers over the C ABI (native.py).

Layout convention: BEV canvases and feature maps are channels-last in HBM.  A canvas of C
channels is the 2-D tensor [n_cells, C] with n_cells = B*T*ny*nx in (b, t, y, x) order; viewed as
[B*T, C, ny, nx] it is a torch channels_last tensor, as [B, C, T, ny, nx] a channels_last_3d one.
"""
import weakref

import os

import torch

from . import native


class PillarIndex(object):
    """Everything derived from (coordinates, point_to_voxel_map) that the forward reuses:
    linear cell index per pillar, dense cell -> pillar table, point -> pillar CSR, and the per-frame
    pillar lists in ascending cell order.  Built once per forward (the reference recomputes the index
    arithmetic inside every scatter call, models/pillar_encoder.py:144-170, 193-203)."""

    def __init__(self, coordinates, point_to_voxel_map, batch_size, input_shape, cell_order=False):
        """cell_order=True renumbers the pillars in ascending cell order for everything built here (`coordinates`, `p2v`, `cell`,
        `cell2pillar`, the CSR; `perm[new id] = id in the caller's numbering`).  Precondition: every pillar owns one in-range cell (what
        the voxeliser guarantees; duplicate or out-of-range rows in `coordinates` would leave ids without a rank) -- MotionNet.forward
        checks the occupied-cell count against m at its one host sync and raises.  The voxeliser numbers pillars in first-touch
        order of the points (libs/voxel_generator.py:41-58), i.e. randomly with respect to the canvas, which turns every
        pillar <-> canvas transfer (pillar scatter, its backward, the inverse scatter) into random row accesses; in cell
        order they stream.  Nothing per-pillar leaves MotionNet.forward, so the renumbering is invisible outside."""
        nx, ny, nz, nt = (int(v) for v in input_shape[:4])
        self.nx, self.ny, self.nt, self.batch_size = nx, ny, nt, int(batch_size)
        self.m = int(coordinates.shape[0])
        self.n_cells = self.batch_size * nt * ny * nx
        self.cells_per_frame = ny * nx
        coords = coordinates.contiguous()
        if coords.dtype not in (torch.float64, torch.int32):
            coords = coords.to(torch.float64)
        self.cell, self.cell2pillar = native.cell_index(coords, nx, ny, nt, self.batch_size)
        self.coordinates, self.perm, self._frames = coordinates, None, None
        p2v = point_to_voxel_map
        if p2v is not None:
            p2v = (p2v[:, 0] if p2v.dim() == 2 else p2v).to(torch.int32)
        if cell_order and self.m > 0:
            sp, offs = native.frame_pillars(self.cell2pillar, self.cells_per_frame, self.m)     # caller's ids in cell order
            spl = sp.long()
            rank = torch.full((self.m,), -1, dtype=torch.int32, device=sp.device)   # pillars that own no cell keep -1 (see check below)
            rank[spl] = torch.arange(self.m, dtype=torch.int32, device=sp.device)
            self.perm = sp
            self.cell = self.cell[spl].contiguous()                          # ascending
            self.cell2pillar = torch.where(self.cell2pillar >= 0, rank[self.cell2pillar.clamp(min=0).long()], self.cell2pillar)
            self.coordinates = coordinates[spl]
            self._frames = (torch.arange(self.m, dtype=torch.int32, device=sp.device), offs)
            if p2v is not None:
                # points of a pillar without a rank (malformed `coordinates`: MotionNet.forward raises at its host sync, but the kernels
                # queued before that would index with -1) are parked on pillar 0: wrong values for a batch that is rejected anyway, no
                # out-of-bounds access
                p2v = rank.clamp(min=0)[p2v.long()]
        if p2v is not None:
            self.p2v = p2v.contiguous()
            self.n = int(self.p2v.shape[0])
            self.seg_offsets, self.order = native.csr_build(self.p2v, self.m)

    @classmethod
    def from_point_map(cls, point_to_voxel_map, m):
        """CSR only (no canvas geometry): enough for the per-pillar reductions of the pillar encoder."""
        self = cls.__new__(cls)
        p2v = point_to_voxel_map[:, 0] if point_to_voxel_map.dim() == 2 else point_to_voxel_map
        self.m = int(m)
        self.p2v = p2v.to(torch.int32).contiguous()
        self.n = int(self.p2v.shape[0])
        self.seg_offsets, self.order = native.csr_build(self.p2v, self.m)
        self._frames = None
        return self

    def pillar_major(self):
        """[r6] The same index for rows that are stored in the order of `self.order` (ascending pillar, ascending point inside a pillar -- the reference's own
        [M, max_points, C] layout, libs/voxel_generator.py:41-58): `p2v` ascending, `order` the identity, everything per-pillar shared.  The pillar encoder
        runs on such rows (MotionNet.prepare_inputs builds its nine inputs per point in this order): its poolings read consecutive rows and its
        pillar -> point broadcasts consecutive table rows instead of gathering them at random; a stable order keeps every tie where it was."""
        if getattr(self, '_pillar_major', None) is None:
            import copy
            pm = copy.copy(self)
            pm.p2v = native.gather_rows(self.p2v.view(-1, 1), self.order).view(-1)
            pm.order = torch.arange(self.n, dtype=torch.int32, device=self.p2v.device)
            pm._pillar_major = pm
            self._pillar_major = pm
        return self._pillar_major

    def frame_pillars(self):
        """(sorted_pillars [M] i32, frame_offsets [B*T+1] i32): occupied pillars per frame in cell order."""
        if self._frames is None:
            self._frames = native.frame_pillars(self.cell2pillar, self.cells_per_frame, self.m)
        return self._frames


# ---------------------------------------------------------------------------------------------------
_SEGMAX_DUAL = os.environ.get('PCACC_SEGMAX_DUAL', '1') != '0'      # A/B switch: '0' = the pooled rows' bf16 shadow by a conversion pass of its own (before round 5)


class _SegmentMax(torch.autograd.Function):
    """scatter(net, p2v, dim=0, reduce='max') -- models/pillar_encoder.py:116,120."""

    @staticmethod
    def forward(ctx, src, pidx):
        src = src.contiguous()
        mixed = _MIXED and src.dtype == torch.bfloat16 and twin(src, required=False) is not None
        if mixed:                                                  # shadow rows: maxima AND winners from the fp32 twin (a bf16 copy ties close values)
            s32 = twin(src)
            if _SEGMAX_DUAL and s32.dtype == torch.float32 and s32.is_contiguous() and not native._seg_two_level(s32.shape[0], pidx.m):
                out, out16, arg = native.segment_max_dual(s32, pidx.seg_offsets, pidx.order, pidx.m)   # the shadow from the same store, not a pass of its own
                out = shadow(carry_amax(s32, out), out16)
            else:
                out, arg = native.segment_max(s32, pidx.seg_offsets, pidx.order, pidx.m)
                out = shadow(carry_amax(s32, out))
        else:
            out, arg = native.segment_max(src if src.dtype in (torch.float32, torch.bfloat16) else src.float(), pidx.seg_offsets,
                                          pidx.order, pidx.m)
        ctx.pidx = pidx
        ctx.src_dtype = src.dtype
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        pidx = ctx.pidx
        out_dtype = ctx.src_dtype if ctx.src_dtype in (torch.float32, torch.bfloat16) else torch.float32
        grad_out = grad_out.contiguous()
        g = native.segment_max_backward(grad_out, arg, pidx.p2v, pidx.n, out_dtype=out_dtype)
        if _SPLIT and g.dtype == torch.float32 and grad_out.dtype == torch.float32 and g.numel() >= 8 * grad_out.numel():
            set_amax_tag(g, amax_of(grad_out))                     # g = grad_out's elements at the winners, zeros elsewhere: its bound holds (a pass over the
        return (g if g.dtype == ctx.src_dtype else g.to(ctx.src_dtype)), None      # few pooled rows instead of one over every point row)


def segment_max(src, pidx):
    return _SegmentMax.apply(src, pidx)[0]


class _SegmentMaxCanvas(torch.autograd.Function):
    """[r6] scatter(net, p2v, 'max') and scatter_point_pillar (models/pillar_encoder.py:119-122 -> :125-174) as ONE pass in the 'mixed' mode: the pooling
    walks the canvas cells and writes the fp32 canvas and its bf16 shadow itself (pcacc_segment_max_canvas); no [M, C] pooled-row table.  Backward: the point
    rows' gradient straight from the canvas gradient through the pillars' cell numbers (no gathered [M, C] copy)."""

    @staticmethod
    def forward(ctx, src, pidx):
        s32 = twin(src)
        canvas32, canvas16, arg = native.segment_max_canvas(s32, pidx.seg_offsets, pidx.order, pidx.m, pidx.cell2pillar)
        carry_amax(s32, canvas32)                                  # maxima of src's rows, or zeros: its bound holds
        if _POISON:
            canvas16.fill_(float('nan'))
        ctx.pidx = pidx
        ctx.save_for_backward(arg)
        return shadow(canvas32, canvas16)

    @staticmethod
    def backward(ctx, grad_canvas):
        (arg,) = ctx.saved_tensors
        pidx = ctx.pidx
        g = grad_canvas.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        return native.segment_max_canvas_backward(g, arg, pidx.p2v, pidx.cell, pidx.n, out_dtype=torch.bfloat16), None


_FUSED_CANVAS = os.environ.get('PCACC_FUSED_CANVAS', '1') != '0'      # A/B switch: '0' = pooling, then the two canvas fills (rounds 1-5)


def segment_max_canvas_available(src, pidx):
    """'mixed' mode on the GPU, bf16 shadow rows with a contiguous fp32 twin, a full pillar index (cell table), short segments."""
    if not (_FUSED_CANVAS and _MIXED and src.is_cuda and src.dtype == torch.bfloat16 and src.dim() == 2 and src.shape[1] % 4 == 0 and src.is_contiguous()
            and getattr(pidx, 'cell2pillar', None) is not None and getattr(pidx, 'cell', None) is not None and pidx.m > 0
            and not native._seg_two_level(src.shape[0], pidx.m)):
        return False
    t = twin(src, required=False)
    return t is not None and t.dtype == torch.float32 and t.is_contiguous()


def segment_max_canvas(src, pidx):
    """-> the canvas [n_cells, C] (a bf16 shadow with its fp32 twin registered) of the per-pillar maxima of `src`."""
    return _SegmentMaxCanvas.apply(src.contiguous(), pidx)


class _BroadcastToPoints(torch.autograd.Function):
    """pillar_feats[point_to_voxel_map] -- the gather after each pooling (pillar_encoder.py:116)."""

    @staticmethod
    def forward(ctx, pillar_feats, pidx):
        ctx.pidx = pidx
        return native.gather_rows(pillar_feats.contiguous(), pidx.p2v)

    @staticmethod
    def backward(ctx, grad):
        pidx = ctx.pidx
        grad = grad.contiguous()
        return native.segment_sum(grad if grad.dtype in (torch.float32, torch.bfloat16) else grad.float(), pidx.seg_offsets,
                                  pidx.order, pidx.m), None


def broadcast_to_points(pillar_feats, pidx):
    return _BroadcastToPoints.apply(pillar_feats, pidx)


def segment_mean3_maxlabel(points, labels, pidx):
    """models/motionnet.py:159-160 (no gradient: inputs are data)."""
    lab = None
    if labels is not None:
        lab = labels.reshape(-1).to(torch.int64).contiguous()
    mean, mlab = native.segment_mean3_maxlabel(points.contiguous(), lab, pidx.seg_offsets, pidx.order, pidx.m)
    return mean, mlab


class _PillarScatter(torch.autograd.Function):
    """scatter_point_pillar (models/pillar_encoder.py:125-174) into a channels-last canvas [n_cells, C].  bf16 rows go into a bf16
    canvas as they are (bf16 compute mode: no fp32 [M,C] table between the pillar encoder and the canvas); everything else is read
    as fp32.  The gradient comes back in the element type of `feats`."""

    @staticmethod
    def forward(ctx, feats, pidx, out_dtype):
        ctx.pidx, ctx.in_dtype = pidx, feats.dtype
        feats = feats.contiguous()
        if _MIXED and feats.dtype == torch.bfloat16 and feats.shape[1] % 8 == 0 and twin(feats, required=False) is not None:
            # shadow rows -> shadow canvas: the fp32 canvas from the twin rows, the bf16 canvas from the shadow rows (two streaming fills instead
            # of a fill and a cast pass); the gradient comes back as a bf16 canvas and leaves as bf16 rows
            f32 = twin(feats)
            canvas32 = carry_amax(f32, native.pillar_scatter(f32, pidx.cell2pillar, torch.float32))
            canvas16 = native.pillar_scatter(feats, pidx.cell2pillar, torch.bfloat16)
            return shadow(canvas32, canvas16)
        if not (feats.dtype == torch.bfloat16 and out_dtype == torch.bfloat16 and feats.shape[1] % 8 == 0):
            feats = feats.float()
        return native.pillar_scatter(feats, pidx.cell2pillar, out_dtype)

    @staticmethod
    def backward(ctx, grad_canvas):
        g = native.gather_rows(grad_canvas.contiguous(), ctx.pidx.cell)
        return (g if g.dtype == ctx.in_dtype else g.to(ctx.in_dtype)), None, None


def pillar_scatter(feats, pidx, out_dtype=torch.float32):
    return _PillarScatter.apply(feats, pidx, out_dtype)


def canvas_as_nchw(canvas, pidx):
    """[n_cells, C] -> logical [B*T, C, ny, nx] (channels_last strides, zero copy)."""
    c = canvas.shape[1]
    return canvas.view(pidx.batch_size * pidx.nt, pidx.ny, pidx.nx, c).permute(0, 3, 1, 2)


def canvas_as_reference(canvas, pidx):
    """[n_cells, C] -> logical [B, C, nt, ny, nx], the reference's return layout (zero copy)."""
    c = canvas.shape[1]
    return canvas.view(pidx.batch_size, pidx.nt, pidx.ny, pidx.nx, c).permute(0, 4, 1, 2, 3)


def nchw_as_rows(x):
    """logical [N, C, H, W] -> [N*H*W, C] rows; free when x is channels_last."""
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).contiguous().view(n * h * w, c)


class _BilinearGather(torch.autograd.Function):
    """ungrid (models/pillar_encoder.py:231-267) as a direct 4-tap gather on a channels-last map."""

    @staticmethod
    def forward(ctx, fmap, points, map_idx, x_scale, y_scale):
        fm = fmap.permute(0, 2, 3, 1).contiguous()                      # [N,H,W,C]; no copy if channels_last
        if fm.dtype not in (torch.float32, torch.bfloat16):
            fm = fm.float()
        ctx.shape = tuple(fm.shape)
        ctx.in_dtype = fmap.dtype
        ctx.scales = (float(x_scale), float(y_scale))
        ctx.save_for_backward(points, map_idx)
        return native.bilinear_gather(twin_or_self(fm), points, map_idx, float(x_scale), float(y_scale))

    @staticmethod
    def backward(ctx, grad_out):
        points, map_idx = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        if grad_out.dtype not in (torch.float32, torch.bfloat16):
            grad_out = grad_out.float()
        out_dtype = ctx.in_dtype if ctx.in_dtype in (torch.float32, torch.bfloat16) else torch.float32
        # sorted, atomic-free reduction (deterministic); writes the map in its own element type
        g = native.bilinear_gather_backward_sorted(grad_out, ctx.shape, points, map_idx, *ctx.scales, out_dtype=out_dtype)
        g = g.permute(0, 3, 1, 2)
        return (g if g.dtype == ctx.in_dtype else g.to(ctx.in_dtype)), None, None, None, None


def bilinear_gather(fmap, points, map_idx, x_scale, y_scale):
    """fmap logical [N,C,H,W]; points [K,3] f32; map_idx [K] i32 -> [K,C] f32 (input order)."""
    return _BilinearGather.apply(fmap, points.contiguous().float(), map_idx.to(torch.int32).contiguous(), x_scale, y_scale)


def bev_warp(bev_cl, inv_pose, x_reso, y_reso, x_min, y_min):
    """bev_cl [B,T,H,W,C] (detached); inv_pose [B,T,4,4] f32 -> warped [B,T,H,W,C] (motionnet.py:82-114)."""
    return native.bev_warp(bev_cl.contiguous(), inv_pose.contiguous().float(), float(x_reso), float(y_reso),
                           float(x_min), float(y_min))


def rigid_transform(points, frame_idx, tsfm):
    """points [N,3] f32, frame_idx [N] i32 = b*T+t, tsfm [B,T,4,4] -> [N,3] (motionnet.py:117-135)."""
    return native.rigid_transform(points.contiguous().float(), frame_idx.to(torch.int32).contiguous(),
                                  tsfm.reshape(-1, 16).contiguous().float())


def gather_rows(src2d, idx):
    return native.gather_rows(src2d.contiguous(), idx.to(torch.int32).contiguous())


# [r6] A training step is bit-reproducible: every sum over rows runs in a fixed order or in integers (CSR ascending inside segments of any length, piece sums
# added in piece order, partial-slot reductions in slot order, the offset centres and the few-row sums of the TubeNet in 64-bit fixed point).  The
# neighbourhood sums of the sparse ego-head convolution's backward used Tensor.index_add_ (fp32 atomics in arrival order): with this switch on (default) they
# take a CSR + the atomic-free segment sum instead.  PCACC_DETERMINISTIC=0 restores index_add_ (A/B only).
DETERMINISTIC = os.environ.get('PCACC_DETERMINISTIC', '1') != '0'


class ScatterPlan(object):
    """An index vector prepared for several scatter() calls: int32 copy now, CSR (for 'max') and counts (for 'mean') on demand."""

    def __init__(self, index, dim_size=None):
        idx = index.reshape(-1)
        self.m = int(dim_size) if dim_size is not None else (int(idx.max()) + 1 if idx.numel() else 0)
        self.p2v = idx.to(torch.int32).contiguous()
        self.n = int(self.p2v.shape[0])
        self._csr = None
        self._count = None

    def _build(self):
        if self._csr is None:
            self._csr = native.csr_build(self.p2v, self.m)
        return self._csr

    @property
    def seg_offsets(self):
        return self._build()[0]

    @property
    def order(self):
        return self._build()[1]

    def small(self, c):
        return 0 < self.m * c <= 8192

    def count(self):
        if self._count is None:
            if self.small(1):
                ones = torch.ones((self.n, 1), dtype=torch.float32, device=self.p2v.device)
                self._count = native.scatter_sum_small(ones, self.p2v, self.m)[:, 0]
            else:
                self._count = (self.seg_offsets[1:] - self.seg_offsets[:-1]).to(torch.float32)
        return self._count


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, plan):
        ctx.plan = plan
        return native.segment_sum(src, plan.seg_offsets, plan.order, plan.m)

    @staticmethod
    def backward(ctx, grad):
        return native.gather_rows(grad.contiguous(), ctx.plan.p2v), None


class _ScatterSumSmall(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, plan):
        ctx.plan = plan
        return native.scatter_sum_small(src, plan.p2v, plan.m)

    @staticmethod
    def backward(ctx, grad):
        return native.gather_rows(grad.contiguous(), ctx.plan.p2v), None


def _pad4(x):
    """[N,C] f32 with C padded up to the next width the segment kernels are instantiated for (4 * 2^k floats per row)."""
    c = x.shape[1]
    width = 4
    while width < c:
        width *= 2
    if width != c:
        x = torch.nn.functional.pad(x, (0, width - c))
    return x.contiguous(), c


def scatter(src, index, dim=0, dim_size=None, reduce='sum', plan=None):
    """torch_scatter.scatter(src, index, dim=0, dim_size, reduce) as the per-instance TubeNet and the offset
    loss use it (models/tpointnet.py:227-284, models/alignnet.py:133-134, libs/loss.py:216): few output rows
    (K*T ~ 100), many inputs per row.  torch's scatter_reduce resolves that with one atomic per element, which on
    ROCm is a compare-and-swap loop for fp64 add and float max -- minutes at 800 k points on 21 rows.  Here the
    index is sorted once into a CSR (ScatterPlan) and each reduction is an atomic-free segmented loop.
    Values are reduced in fp32 and returned in src's dtype; empty segments are 0; 'max' routes the gradient to
    the lowest index attaining the maximum."""
    assert dim == 0 and reduce in ('sum', 'mean', 'max')
    if plan is None:
        plan = ScatterPlan(index, dim_size)
    shape_tail = tuple(src.shape[1:])
    x = src.reshape(src.shape[0], -1)
    c = x.shape[1]
    if reduce != 'max' and plan.small(c):
        out = _ScatterSumSmall.apply(x.to(torch.float32).contiguous(), plan)          # few rows: LDS-privatised, no CSR
    elif reduce == 'max' and x.dtype == torch.bfloat16:
        xp, c = _pad4(x)                                                              # bf16 rows go in as they are (max is exact)
        out = _SegmentMax.apply(xp, plan)[0]
        if _MIXED and twin(out, required=False) is not None:                          # shadow rows: the pooled rows leave the bf16 graph as their fp32 twin
            return exit_mixed(out)[:, :c].reshape((plan.m,) + shape_tail)             # (fp32: NOT rounded back to the shadow's element type)
    else:
        x32, c = _pad4(x.to(torch.float32))
        out = _SegmentMax.apply(x32, plan)[0] if reduce == 'max' else _SegmentSum.apply(x32, plan)
    if reduce == 'mean':
        out = out / plan.count().clamp(min=1.0)[:, None]
    out = out[:, :c].reshape((plan.m,) + shape_tail)
    return out.to(src.dtype) if out.dtype != src.dtype else out


# ---------------------------------------------------------------------------------------------------
MIN_ROWS_FUSED_LINEAR = 2048          # below this a library GEMM launch is as good; above it the layer is an HBM stream


class _RowsLinear(torch.autograd.Function):
    """nn.Linear on [rows, k] with optional ReLU on the input, ReLU on the output and a residual add, one HBM pass
    (pcacc_rows_linear*); backward = the same kernel with w^T (masks replay the ReLUs) + the MFMA weight-gradient kernel.
    Rows may be fp32 or bf16 (bf16 compute mode): the output takes `out_dtype`, gradients take the dtype of what they
    are the gradient of, weights and their gradients stay fp32.  fp32x3 mode (ops.set_split) with all-fp32 rows and widths in
    {32, 64, 128}: the split-fp16 matrix-core kernels of csrc/mlp_split.hip, absolute maxima taken once per tensor."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, pre_relu, post_relu, out_dtype):
        x = x.contiguous()
        w = weight.contiguous()
        res = residual.contiguous() if residual is not None else None
        k, n = w.shape[1], w.shape[0]
        split = (_SPLIT and x.dtype == torch.float32 and out_dtype == torch.float32 and native.rows_split_supported(k, n)
                 and w.dtype == torch.float32 and (res is None or res.dtype == torch.float32))
        x_amax = None
        if split:
            x_amax = amax_of(x)
            y, y_amax = native.rows_linear_split(x, x_amax, w, bias, res, pre_relu, post_relu, want_amax=True)
            set_amax_tag(y, y_amax)
        else:
            y = native.rows_linear(x, w, bias, res, pre_relu, post_relu, out_dtype=out_dtype)
        ctx.flags = (pre_relu, post_relu, bias is not None, residual is not None, res.dtype if res is not None else None, split)
        ctx.save_for_backward(x, w, y if post_relu else None, x_amax)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y, x_amax = ctx.saved_tensors
        pre_relu, post_relu, has_bias, has_res, res_dtype, split = ctx.flags
        gy = dense(gy)
        gx = gw = gb = gres = None
        if split:
            if gy.dtype != torch.float32:
                gy = gy.float()
            g_amax = amax_of(gy)
            if ctx.needs_input_grad[0]:
                gx, gx_amax = native.rows_linear_split(gy, g_amax, w.t().contiguous(), None, None, False, False, in_mask=y,
                                                       out_mask=x if pre_relu else None, want_amax=True)
                set_amax_tag(gx, gx_amax)
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                gw, gb = native.rows_wgrad_split(gy, g_amax, x, x_amax, dy_mask=y, x_relu=pre_relu, split=True)
                gb = gb if has_bias else None
        else:
            if ctx.needs_input_grad[0]:
                gx = native.rows_linear(gy, w.t().contiguous(), None, None, False, False, in_mask=y, out_mask=x if pre_relu else None,
                                        out_dtype=x.dtype)
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                gw, gb = native.rows_wgrad(gy, x, dy_mask=y, x_relu=pre_relu, split=True)     # contiguous: autograd adopts them without a copy
                gb = gb if has_bias else None
        if has_res and ctx.needs_input_grad[3]:
            gres = (gy if y is None else gy * (y > 0)).to(res_dtype)
            if split:
                carry_amax(gy, gres)                           # gy itself, or gy with some elements zeroed
        return gx, gw, gb, gres, None, None, None


class _RowsLinearMixed(_RowsLinear):
    """'mixed' mode: _RowsLinear's forward in fp32 (on the twin of a shadow input, or on an fp32 input that carries no gradient: the
    9-feature rows), the result registered as the twin of the bf16 shadow autograd sees; the backward is _RowsLinear's on bf16 rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, pre_relu, post_relu, out_dtype):
        x = x.contiguous()
        w = weight.contiguous()
        x32 = twin(x) if x.dtype == torch.bfloat16 else x
        res = residual.contiguous() if residual is not None else None
        res32 = (twin(res) if res.dtype == torch.bfloat16 else res) if res is not None else None
        k, n = w.shape[1], w.shape[0]
        if native.rows_split_supported(k, n) and w.dtype == torch.float32:
            y32, y_amax, y16 = native.rows_linear_split(x32, amax_of(x32), w, bias, res32, pre_relu, post_relu, want_bf16=True)   # shadow from the same epilogue
            set_amax_tag(y32, y_amax)
            y = shadow(y32, y16)
        elif k <= 9 and n in (8, 16, 32, 64, 128) and w.dtype == torch.float32 and x32.dtype == torch.float32:
            y32, y_amax, y16 = native.rows_linear_few_dual(x32, w, bias, res32, pre_relu, post_relu)      # plain fp32 FMAs; shadow + maxima from the store phase
            set_amax_tag(y32, y_amax)
            y = shadow(y32, y16)
        else:
            y32 = native.rows_linear(x32, w, bias, res32, pre_relu, post_relu, out_dtype=torch.float32)
            y = shadow(y32)
        ctx.flags = (pre_relu, post_relu, bias is not None, residual is not None, res.dtype if res is not None else None, False)
        ctx.save_for_backward(x, w, y if post_relu else None, None)
        return y


class _RowsLinearCat(torch.autograd.Function):
    """_RowsLinear on x = cat(xa, pooled[p2v]) without materialising the gather or the concatenation (bf16 rows, or fp32 rows in the
    fp32x3 mode, on the GPU; pcacc_rows_linear_cat_bf16 / _cat_split): the PFN blocks' input (models/pillar_encoder.py:116-118).
    Backward: the data gradient leaves the kernel already split; the pooled half is summed over each pillar's points (CSR segment sum)."""

    @staticmethod
    def forward(ctx, xa, pooled, pidx, weight, bias, residual, pre_relu, post_relu):
        xa, pooled, w = xa.contiguous(), pooled.contiguous(), weight.contiguous()
        res = residual.contiguous() if residual is not None else None
        split = xa.dtype == torch.float32
        amax = (None, None)
        if split:
            amax = (amax_of(xa), amax_of(pooled))
            y, y_amax = native.rows_linear_cat_split(xa, amax[0], pooled, amax[1], pidx.p2v, w, bias, res, pre_relu, post_relu, want_amax=True)
            set_amax_tag(y, y_amax)
        else:
            y = native.rows_linear_cat(xa, pooled, pidx.p2v, w, bias, res, pre_relu, post_relu)
        ctx.pidx = pidx
        ctx.flags = (pre_relu, post_relu, bias is not None, residual is not None, split)
        ctx.save_for_backward(xa, pooled, w, y if post_relu else None, amax[0], amax[1])
        return y

    @staticmethod
    def backward(ctx, gy):
        xa, pooled, w, y, a_amax, p_amax = ctx.saved_tensors
        pre_relu, post_relu, has_bias, has_res, split = ctx.flags
        pidx = ctx.pidx
        gy = dense(gy)
        ga = gp = gw = gb = gres = None
        g_amax = amax_of(gy) if split else None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if split:
                ga, gb_rows, gab_amax = native.rows_linear_cat_backward_split(gy, g_amax, w.t().contiguous(), y, xa, pooled, pidx.p2v, pre_relu)
                set_amax_tag(ga, gab_amax)
            else:
                ga, gb_rows = native.rows_linear_cat_backward(gy, w.t().contiguous(), y, xa, pooled, pidx.p2v, pre_relu)
            if ctx.needs_input_grad[1]:
                gp = native.segment_sum(gb_rows, pidx.seg_offsets, pidx.order, pidx.m).to(pooled.dtype)
        if ctx.needs_input_grad[3] or (has_bias and ctx.needs_input_grad[4]):
            if split:
                gw, gb = native.rows_wgrad_cat_split(gy, g_amax, xa, a_amax, pooled, p_amax, pidx.p2v, dy_mask=y, x_relu=pre_relu, split=True)
            else:
                gw, gb = native.rows_wgrad_cat(gy, xa, pooled, pidx.p2v, dy_mask=y, x_relu=pre_relu, split=True)
            gb = gb if has_bias else None
        if has_res and ctx.needs_input_grad[5]:
            gres = gy if y is None else gy * (y > 0)
            if split:
                carry_amax(gy, gres)
        return ga, gp, None, gw, gb, gres, None, None


def linear_rows_cat_available(xa, pooled, layer):
    """bf16 point rows (or fp32 rows in the fp32x3 mode) on the GPU with widths the matrix-core kernels take; otherwise callers concatenate
    (parity mode, CPU)."""
    return (xa.is_cuda and xa.dtype == pooled.dtype and (xa.dtype == torch.bfloat16 or (_SPLIT and xa.dtype == torch.float32 and layer.weight.dtype == torch.float32))
            and xa.shape[0] >= MIN_ROWS_FUSED_LINEAR
            and layer.in_features == xa.shape[1] + pooled.shape[1]
            and native.rows_linear_cat_supported(xa.shape[1], pooled.shape[1], layer.out_features) and not torch.is_autocast_enabled())


def linear_rows_cat(xa, pooled, pidx, layer, pre_relu=False, post_relu=False, residual=None):
    """`layer(relu?(cat(xa, pooled[pidx.p2v])))` (+ residual, relu?) -- see _RowsLinearCat."""
    return _RowsLinearCat.apply(xa, pooled, pidx, layer.weight, layer.bias, residual, pre_relu, post_relu)


class _PfnBlock(torch.autograd.Function):
    """ResnetBlockFC(64, 32) on bf16 point rows as one kernel each way (csrc/pfn_block.hip); x = xa [rows,64], or
    cat(xa [rows,32], pooled[pidx.p2v]) without materialising gather or concatenation.  The pooled half of the data gradient is
    summed over each pillar's points (CSR segment sum), as in _RowsLinearCat.
    pool=True: pooled = segment_max(xa) is taken here too (models/pillar_encoder.py:116-118 as one autograd node): xa's gradient
    through the max-pool is added into its direct gradient in one pass (pcacc_segment_max_backward_acc) instead of a dense
    max-pool gradient plus autograd's sum of the two."""

    @staticmethod
    def forward(ctx, xa, pooled, pidx, w0, b0, ws, w1, b1, pool=False):
        xa = xa.contiguous()
        arg = None
        if pool:
            pooled, arg = native.segment_max(xa, pidx.seg_offsets, pidx.order, pidx.m)
        pooled = pooled.contiguous() if pooled is not None else None
        w0, ws, w1 = w0.contiguous(), ws.contiguous(), w1.contiguous()
        out, hr = native.pfn_block_forward(xa, pooled, pidx.p2v if pooled is not None else None, w0, b0, ws, w1, b1)
        ctx.pidx = pidx
        ctx.save_for_backward(xa, pooled, hr, w0, ws, w1, arg)
        ctx.has_bias = (b0 is not None, b1 is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        xa, pooled, hr, w0, ws, w1, arg = ctx.saved_tensors
        pidx = ctx.pidx
        gxa, gxb, gp = native.pfn_block_backward(xa, pooled, pidx.p2v if pooled is not None else None, hr, g.contiguous(), w0, ws, w1)
        gpool = None
        if pooled is not None and (arg is not None or ctx.needs_input_grad[1]):
            gpool = native.segment_sum(gxb, pidx.seg_offsets, pidx.order, pidx.m)
            if arg is not None:
                native.segment_max_backward_acc(gpool, arg, pidx.p2v, gxa)
                gpool = None
            else:
                gpool = gpool.to(pooled.dtype)
        part = lambda name: gp[native.PFN_BLOCK_SLICES[name][0]:native.PFN_BLOCK_SLICES[name][1]].view(native.PFN_BLOCK_SLICES[name][2])
        return (gxa, gpool, None, part('w0'), part('b0') if ctx.has_bias[0] else None, part('ws'), part('w1'),
                part('b1') if ctx.has_bias[1] else None, None)


class _PfnBlockSplit(torch.autograd.Function):
    """_PfnBlock on fp32 rows in the fp32x3 mode (csrc/pfn_block_split.hip): one kernel forward; backward = one data-gradient kernel
    (reads d(out) and two sign masks) + the three weight gradients on the row weight-gradient kernels."""

    @staticmethod
    def forward(ctx, xa, pooled, pidx, w0, b0, ws, w1, b1, pool=False):
        xa = xa.contiguous()
        a_amax = amax_of(xa)
        arg = None
        if pool:
            pooled, arg = native.segment_max(xa, pidx.seg_offsets, pidx.order, pidx.m)
            p_amax = a_amax                                                   # maxima of xa's rows: its bound holds
        else:
            pooled = pooled.contiguous() if pooled is not None else None
            p_amax = amax_of(pooled) if pooled is not None else None
        w0, ws, w1 = w0.contiguous(), ws.contiguous(), w1.contiguous()
        out, hr, xmask, hmask, out_amax, hr_amax = native.pfn_block_split_forward(xa, a_amax, pooled, p_amax, pidx.p2v if pooled is not None else None,
                                                                                  w0, b0, ws, w1, b1)
        set_amax_tag(out, out_amax)
        ctx.pidx = pidx
        ctx.save_for_backward(xa, pooled, hr, xmask, hmask, w0, ws, w1, a_amax, p_amax, hr_amax, arg)
        ctx.has_bias = (b0 is not None, b1 is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        xa, pooled, hr, xmask, hmask, w0, ws, w1, a_amax, p_amax, hr_amax, arg = ctx.saved_tensors
        pidx = ctx.pidx
        g = dense(g)
        g_amax = amax_of(g)
        gxa, gxb, dh, gx_amax, dh_amax = native.pfn_block_split_dgrad(g, g_amax, xmask, hmask, w0, ws, w1, pooled is not None)
        gpool = None
        if pooled is not None and (arg is not None or ctx.needs_input_grad[1]):
            gpool = native.segment_sum(gxb, pidx.seg_offsets, pidx.order, pidx.m)
            if arg is not None:
                gx_amax = native.segment_max_backward_acc(gpool, arg, pidx.p2v, gxa, want_amax=True)
                gpool = None
            else:
                gpool = gpool.to(pooled.dtype)
        set_amax_tag(gxa, gx_amax)
        gw1, gb1 = native.rows_wgrad_split(g, g_amax, hr, hr_amax, split=True)
        if pooled is not None:
            gws, _ = native.rows_wgrad_cat_split(g, g_amax, xa, a_amax, pooled, p_amax, pidx.p2v, split=True)
            gw0, gb0 = native.rows_wgrad_cat_split(dh, dh_amax, xa, a_amax, pooled, p_amax, pidx.p2v, x_relu=True, split=True)
        else:
            gws, _ = native.rows_wgrad_split(g, g_amax, xa, a_amax, split=True)
            gw0, gb0 = native.rows_wgrad_split(dh, dh_amax, xa, a_amax, x_relu=True, split=True)
        return gxa, gpool, None, gw0, gb0 if ctx.has_bias[0] else None, gws, gw1, gb1 if ctx.has_bias[1] else None, None


class _PfnBlockMixed(_PfnBlock):
    """'mixed' mode: _PfnBlockSplit's forward on the twins of shadow rows, _PfnBlock's bf16 backward (one kernel for the data gradients and the
    three weight gradients).  The pooling's winners come from the fp32 rows."""

    @staticmethod
    def forward(ctx, xa, pooled, pidx, w0, b0, ws, w1, b1, pool=False):
        xa = xa.contiguous()
        xa32 = twin(xa)
        a_amax = amax_of(xa32)
        arg = None
        if pool:
            pooled32, arg = native.segment_max(xa32, pidx.seg_offsets, pidx.order, pidx.m)
            p_amax = a_amax
            pooled = pooled32.to(torch.bfloat16) if not _POISON else torch.full_like(pooled32, float('nan'), dtype=torch.bfloat16)
        elif pooled is not None:
            pooled = pooled.contiguous()
            pooled32 = twin(pooled)
            p_amax = amax_of(pooled32)
        else:
            pooled32 = p_amax = None
        w0, ws, w1 = w0.contiguous(), ws.contiguous(), w1.contiguous()
        out32, out_amax, out16, hr = native.pfn_block_split_forward_dual(xa32, a_amax, pooled32, p_amax, pidx.p2v if pooled32 is not None else None,
                                                                         w0, b0, ws, w1, b1)     # shadows of out and relu(h) from the same epilogue
        set_amax_tag(out32, out_amax)
        if _POISON:
            hr.fill_(float('nan'))
        ctx.pidx = pidx
        ctx.save_for_backward(xa, pooled, hr, w0, ws, w1, arg)
        ctx.has_bias = (b0 is not None, b1 is not None)
        return shadow(out32, out16)


def pfn_block_available(block, x, pooled=None):
    """The fused block takes bf16 rows (or fp32 rows in the fp32x3 mode) on the GPU, the encoder's widths (64 -> 32 -> 32 with a shortcut)
    and fp32 parameters."""
    width = x.shape[1] + (pooled.shape[1] if pooled is not None else 0)
    dtype_ok = x.dtype == torch.bfloat16 or (_SPLIT and x.dtype == torch.float32)
    return (x.is_cuda and dtype_ok and (pooled is None or (pooled.dtype == x.dtype and x.shape[1] == 32 and pooled.shape[1] == 32))
            and x.shape[0] >= MIN_ROWS_FUSED_LINEAR and width == 64 and block.size_in == 64 and block.size_h == 32 and block.size_out == 32
            and block.shortcut is not None and block.fc_0.weight.dtype == torch.float32 and not torch.is_autocast_enabled())


def pfn_block(block, x, pooled=None, pidx=None, pool=False):
    """block(x) or block(cat(x, pooled[pidx.p2v])) for a pillar_encoder.ResnetBlockFC -- see _PfnBlock / _PfnBlockSplit.
    pool=True: pooled = segment_max(x, pidx), taken inside the same autograd node."""
    fn = _PfnBlockSplit if x.dtype == torch.float32 else (_PfnBlockMixed if _MIXED else _PfnBlock)
    return fn.apply(x, pooled, pidx, block.fc_0.weight, block.fc_0.bias, block.shortcut.weight, block.fc_1.weight, block.fc_1.bias, pool)


def pfn_pool_block_available(block, x, pidx):
    """block(cat(x, segment_max(x)[p2v])) as one node: the fused block's conditions with a [m, 32] pooled half in the rows' type
    (short segments: native.segment_max keeps the type)."""
    return (x.shape[1] == 32 and x.is_cuda and not native._seg_two_level(x.shape[0], pidx.m) and os.environ.get('PCACC_PFN_POOL_NODE', '1') != '0'
            and pfn_block_available(block, x, torch.empty((0, 32), dtype=x.dtype, device=x.device)))


# ---- stages of the forward, for the precision-map experiment (tools/r06_precision_map.py + the -DPCACC_X3_EXPERIMENT build of the library) -------------
# MotionNet.forward brackets its stages with `with ops.stage(name):`; without an experiment configured (always, in production) that is a shared null context.
import contextlib as _contextlib
_NO_STAGE = _contextlib.nullcontext()
_STAGE_WORDS = None                     # {stage name: experiment word (csrc/common.h)}; set by the experiment driver only


class _Stage(object):
    def __init__(self, word):
        self.word = int(word)

    def __enter__(self):
        native.x3_experiment(self.word)

    def __exit__(self, *exc):
        native.x3_experiment(0)
        return False


def stage(name):
    if _STAGE_WORDS is None:
        return _NO_STAGE
    return _Stage(_STAGE_WORDS.get(name, 0))


def set_stage_words(words):
    """Experiment driver: {stage: word} or None.  Needs the experiment build of the library (PCACC_LIB=build/x3exp/libpcacc_hip.so)."""
    global _STAGE_WORDS
    _STAGE_WORDS = dict(words) if words is not None else None
    if words is not None:
        native.x3_experiment(0)


_POINT_DTYPE = torch.float32


def set_point_dtype(dtype):
    """Element type in which the per-point MLP chains keep their activations (MotionNet sets it from cfg misc.compute_dtype):
    float32 (default, the parity mode) or bfloat16 (half the HBM traffic, products on the bf16 matrix cores)."""
    global _POINT_DTYPE
    assert dtype in (torch.float32, torch.bfloat16)
    _POINT_DTYPE = dtype


def point_dtype():
    return _POINT_DTYPE


_SPLIT = False


def set_split(flag):
    """fp32x3 compute mode (MotionNet sets it from cfg misc.compute_dtype == 'fp32x3'): fp32 tensors everywhere, the dense 3x3 stacks and
    the wide per-point linear layers form their products on the bf16 matrix cores from hi / lo halves (csrc/conv_split.hip) instead
    of calling the library's fp32 convolutions / the fp32 vector kernels."""
    global _SPLIT
    _SPLIT = bool(flag)


def split_mode():
    return _SPLIT


# ---- 'mixed' compute mode: fp32x3 forward values, bf16 gradient graph --------------------------------------------------------------------
# north_star's 1e-3 is a statement about FORWARD outputs (mos_iou, ego errors, EPE): they need the fp32-accurate products of the fp32x3
# kernels.  The backward does not: a data / weight gradient formed from bf16 operands with fp32 accumulation carries a relative rounding of
# 2^-9 per element and layer, random in sign -- nothing a gradient norm or an Adam step resolves -- and moves half the bytes.  Autograd ties a
# gradient's type to the type of the forward tensor it belongs to, so inside a mixed SEGMENT (U-Net + heads; STPN temporal stack + U-Net) the
# tensors autograd sees are bf16 SHADOWS of the fp32 activations; every operator of a segment computes its forward from the fp32 TWIN of its
# input shadow (fp32x3 kernels), registers the twin of its output and saves shadows for a bf16 backward (the bf16 mode's kernels).  Forward
# values never come from a shadow: PCACC_MIXED_POISON=1 fills every shadow with NaN in the forward (tests/test_mixed.py).
_MIXED = False
_TWINS = {}            # storage address of a shadow -> (shadow, fp32 twin); same sizes and strides, both at storage offset 0; strong references
_POISON = os.environ.get('PCACC_MIXED_POISON') == '1'


def set_mixed(flag):
    global _MIXED
    _MIXED = bool(flag)
    if not _MIXED:
        _TWINS.clear()


def mixed_mode():
    return _MIXED


class bf16_rows(object):
    """Context: the row operators issued inside run as in the 'bf16' compute mode (bf16 rows, one-term bf16 matrix-core products, fp32 accumulation) although the
    model runs 'mixed' -- the 'mixed2' mode's point heads (MotionNet._stpn_heads).  The precision map of round 6 (profiles/r06_precision_map.txt: every stage
    of the forward reduced alone, six fixtures) shows the STPN's per-point layers to be the one stage whose outputs stay within north_star's 1e-3 with bf16
    products: the fg / moving decision of a point is not a near-tie, and nothing behind these layers amplifies their rounding (the TubeNet reads the STPN's MAP,
    not the point heads).  The flags are read when an operator's forward is ISSUED; each operator's backward follows what its forward saved."""

    def __enter__(self):
        self.saved = (_MIXED, _SPLIT, _POINT_DTYPE)
        set_split(False)
        globals()['_MIXED'] = False                           # not set_mixed(False): the segment's twins are still needed by the operators after the context
        set_point_dtype(torch.bfloat16)
        return self

    def __exit__(self, *exc):
        globals()['_MIXED'] = self.saved[0]
        set_split(self.saved[1])
        set_point_dtype(self.saved[2])
        return False


def set_poison(flag):
    """Test switch: every shadow made from now on holds NaN instead of the rounded twin (forward values must not change; the backward is garbage)."""
    global _POISON
    _POISON = bool(flag)


def twins_clear():
    """Start of a forward: the pairs of the previous one are no longer needed (its backward is done)."""
    _TWINS.clear()


def shadow(y32, y16=None):
    """The bf16 shadow of a dense fp32 tensor (same sizes, same strides), registered as its twin's key.  y16: a shadow somebody else produced
    (torch.cat of shadows, a kernel's second output)."""
    if y32.storage_offset() != 0 or y32.dtype != torch.float32:
        raise native.NativeError('shadow: the twin must be a float32 tensor at storage offset 0')
    if y16 is None:
        y16 = torch.empty_strided(y32.size(), y32.stride(), dtype=torch.bfloat16, device=y32.device)
        if _POISON:
            y16.fill_(float('nan'))
        else:
            y16.copy_(y32)
    elif tuple(y16.stride()) != tuple(y32.stride()) or tuple(y16.size()) != tuple(y32.size()) or y16.storage_offset() != 0:
        raise native.NativeError('shadow: layouts differ')
    elif _POISON and not y16.requires_grad:
        y16.fill_(float('nan'))
    _TWINS[y16.untyped_storage().data_ptr()] = (y16, y32)
    return y16


def twin(t16, required=True):
    """The fp32 twin of a shadow or of any view of one (same view of the twin's storage)."""
    e = _TWINS.get(t16.untyped_storage().data_ptr()) if t16.dtype == torch.bfloat16 else None
    if e is None:
        if required:
            raise native.NativeError('mixed mode: a bf16 tensor without an fp32 twin reached a segment operator (shape %s): its forward values '
                                     'would come from bf16 data' % (tuple(t16.shape),))
        return None
    base = e[1]
    if tuple(t16.size()) == tuple(base.size()) and tuple(t16.stride()) == tuple(base.stride()) and t16.storage_offset() == 0:
        return base
    return torch.as_strided(base, t16.size(), t16.stride(), t16.storage_offset())


def twin_or_self(t):
    """fp32 view of `t` for a forward-only consumer (BEV warp, bilinear gathers, the ego head): the twin of a shadow, `t` itself otherwise."""
    if _MIXED and t.dtype == torch.bfloat16:
        return twin(t)
    return t


class _EnterMixed(torch.autograd.Function):
    """fp32 tensor -> its bf16 shadow: the head of a mixed segment.  The gradient arrives in bf16 and is handed on in fp32."""

    @staticmethod
    def forward(ctx, x):
        return shadow(carry_amax(x, x.detach()))

    @staticmethod
    def backward(ctx, g):
        return g.float()


class _ExitMixed(torch.autograd.Function):
    """shadow -> its fp32 twin as an autograd-visible tensor: the tail of a mixed segment (the fp32 consumer's gradient enters in bf16)."""

    @staticmethod
    def forward(ctx, x16):
        return twin(x16).view_as(x16)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16)


def cat_rows(a, b):
    """cat([a, b in a's element type], dim=-1) for point rows; in the 'mixed' mode (a = bf16 shadow, b = fp32 rows) b enters the shadow graph
    and the twins are concatenated too (models/stpn.py:96-97: positional code | gathered map features)."""
    if _MIXED and a.dtype == torch.bfloat16 and twin(a, required=False) is not None:
        return cat_maps((a, enter_mixed(b.contiguous()) if b.dtype == torch.float32 else b), dim=-1)
    return torch.cat([a, b.to(a.dtype)], dim=-1)


class _OnTwin(torch.autograd.Function):
    """Any differentiable fp32 function of one map, inside a mixed segment, without a kernel pair of its own (BatchNorm2d in eval mode, maps too
    small for the streaming kernels): forward = fn(twin) in fp32; backward = `replay` (default: fn) re-evaluated on the twin under autograd in
    fp32 (the gradient arrives and leaves in bf16).  Correct and slow -- the segment's hot operators do not come through here."""

    @staticmethod
    def forward(ctx, x16, fn, replay, *params):
        ctx.replay, ctx.x32, ctx.params = (replay or fn), twin(x16), params
        y = fn(ctx.x32)
        return shadow(y.contiguous() if y.storage_offset() else y)

    @staticmethod
    def backward(ctx, gy):
        with torch.enable_grad():
            x = ctx.x32.detach().requires_grad_(True)
            y = ctx.replay(x)
            leaves = [x] + [p for p in ctx.params if p.requires_grad]
            grads = torch.autograd.grad(y, leaves, gy.float(), allow_unused=True)
        it = iter(grads[1:])
        return (grads[0].to(torch.bfloat16), None, None) + tuple(next(it) if p.requires_grad else None for p in ctx.params)


def on_twin(x16, fn, params=(), replay=None):
    """fn(twin of x16) as a shadow; `params`: the tensors fn closes over that may need gradients; `replay`: what the backward differentiates
    when fn has side effects (a training-mode BatchNorm's running statistics)."""
    return _OnTwin.apply(x16, fn, replay, *params)


def enter_mixed(x):
    return _EnterMixed.apply(x) if _MIXED and x.is_cuda and x.dtype == torch.float32 else x


def bev_warp_enter_mixed(bev_cl, inv_pose, x_reso, y_reso, x_min, y_min, amax_from=None):
    """enter_mixed(bev_warp(...)) for a map that carries no gradient (models/motionnet.py:205-209 detaches it): in the 'mixed' mode the warp kernel writes
    the bf16 shadow beside its fp32 result (pcacc_bev_warp_dual) instead of a conversion pass over the 212 MB map; everywhere else the two calls."""
    src = twin_or_self(bev_cl)
    if _MIXED and src.is_cuda and src.dtype == torch.float32 and not src.requires_grad and src.is_contiguous() and src.shape[-1] % 4 == 0 and not _POISON:
        w32, w16 = native.bev_warp_dual(src, inv_pose.contiguous().float(), float(x_reso), float(y_reso), float(x_min), float(y_min))
        carry_amax(amax_from if amax_from is not None else src, w32)      # convex combinations of the map's cells (or zero): its bound holds
        return shadow(w32, w16)
    warped = bev_warp(src, inv_pose, x_reso, y_reso, x_min, y_min)
    carry_amax(amax_from if amax_from is not None else src, warped)
    return enter_mixed(warped)


def exit_mixed(x):
    return _ExitMixed.apply(x) if _MIXED and x.dtype == torch.bfloat16 else x


class _Cat2(torch.autograd.Function):
    """cat((a, b), -1) of two contiguous row tensors on the library's own copy kernel (pcacc_cat2_rows); the gradient goes back as the two channel
    slices of the incoming gradient (views: the encoder's pool tail reads its slice in place)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[-1]
        return native.cat2_rows(a, b)

    @staticmethod
    def backward(ctx, g):
        return g[..., :ctx.ca], g[..., ctx.ca:]


def _cat2(a, b, dim):
    """torch.cat((a, b), dim) -- through pcacc_cat2_rows when both are channels-last maps concatenated along the channels (dim 1 of NCHW) or plain
    rows concatenated along their last dimension, on the GPU, with 16-byte row pieces."""
    if a.is_cuda and a.dtype == b.dtype and a.dtype in (torch.float32, torch.bfloat16) and a.dim() == b.dim() \
            and os.environ.get('PCACC_OWN_CAT', '1') != '0':
        ar = None
        if a.dim() == 4 and dim == 1:
            ar, br = a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)
            back = lambda y: y.permute(0, 3, 1, 2)
        elif dim in (-1, a.dim() - 1):
            ar, br, back = a, b, (lambda y: y)
        if ar is not None and ar.is_contiguous() and br.is_contiguous() and tuple(ar.shape[:-1]) == tuple(br.shape[:-1]) \
                and (ar.shape[-1] * a.element_size()) % 16 == 0 and (br.shape[-1] * a.element_size()) % 16 == 0 and ar.numel() > 0 \
                and ar.data_ptr() % 16 == 0 and br.data_ptr() % 16 == 0:
            return back(_Cat2.apply(ar, br))
    return torch.cat((a, b), dim)


def cat_maps(tensors, dim=1):
    """torch.cat for maps inside a mixed segment (shadows: the twins are concatenated too and registered); the plain concatenation otherwise."""
    a, b = tensors
    y = _cat2(a, b, dim)
    if _MIXED and y.dtype == torch.bfloat16:
        tw = [twin(t) for t in tensors]
        y32 = merge_amax(_cat2(tw[0], tw[1], dim), *tw)
        if y32._base is not None and y32._base.numel() == y32.numel():        # a permuted view of the whole concatenation: the consumer's view of the
            carry_amax(y32, y32._base)                                        # twin is rebuilt from the storage and finds the tag on the base only
        if _POISON:
            with torch.no_grad():
                y.fill_(float('nan'))
        shadow(y32, y)
        return y
    return merge_amax(y, *tensors)


# ---- absolute maxima of the fp32 tensors the fp32x3 kernels split (their power-of-two scales) ----------------------------------------
# The kernels that produce a tensor collect its maximum in their store phase; the array rides on the tensor OBJECT as `_pcacc_amax` together
# with the tensor's version counter (an in-place write invalidates it) and is picked up by the next split kernel that reads the tensor --
# forward chains (layer to layer) and backward chains (a data gradient handed on by autograd) alike.  Anything else costs one
# pcacc_absmax256 pass, remembered the same way (a tensor read by two layers is measured once).  Upper bounds are as good as the exact
# maximum up to a bit of precision per factor two, so pure re-arrangements and maxima (views, pooling) may pass the tag on.
def amax_tag(t):
    a = getattr(t, '_pcacc_amax', None)
    if a is not None and a[1] == t._version:
        return a[0]
    b = t._base                                               # a view (autograd's PermuteBackward / ViewBackward hand on views of the tensor a
    if b is not None:                                         # kernel produced): the base's bound holds for any part of it; one version counter
        a = getattr(b, '_pcacc_amax', None)
        if a is not None and a[1] == b._version:
            return a[0]
    return None


def dense(t):
    """t.contiguous(), the tag kept when that is a copy."""
    c = t.contiguous()
    return c if c is t else carry_amax(t, c)


def set_amax_tag(t, parts):
    if parts is not None:
        t._pcacc_amax = (parts, t._version)
    return t


def carry_amax(src, dst):
    """dst holds a subset / re-arrangement / maxima of src's elements: src's bound holds for it."""
    return set_amax_tag(dst, amax_tag(src))


def merge_amax(dst, *srcs):
    """dst is made of the elements of `srcs` (a concatenation): the element-wise maximum of their arrays bounds it (one 256-element launch
    instead of a pass over dst).  No tag when a source has none."""
    tags = [amax_tag(t) for t in srcs]
    if _SPLIT and tags and all(t is not None for t in tags):
        m = tags[0]
        for t in tags[1:]:
            m = torch.maximum(m, t)
        set_amax_tag(dst, m)
    return dst


def amax_of(t):
    a = amax_tag(t)
    if a is None:
        a = native.absmax256(t)
        set_amax_tag(t, a)
    return a


def linear_rows(x, layer, pre_relu=False, post_relu=False, residual=None, out_dtype=None, mixed=False):
    """`layer(relu?(x))` (+ residual, relu?) for an nn.Linear `layer` on a 2-D `x`.  Large row counts with a supported
    feature width go through the fused HIP kernels; anything else is the library GEMM with the same semantics.
    out_dtype: element type of the result (default: that of x); bf16 rows are only taken on the GPU."""
    k, n = layer.in_features, layer.out_features
    out_dtype = out_dtype or x.dtype
    fused = (x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[0] >= MIN_ROWS_FUSED_LINEAR
             and native.rows_linear_supported(k, n) and not torch.is_autocast_enabled())
    if _MIXED and x.is_cuda and (x.dtype == torch.bfloat16 or mixed):
        # a shadow (or, with mixed=True, the fp32 head of a chain that is to continue as shadows): fp32 forward, bf16 backward
        if not fused:
            raise native.NativeError('mixed mode: a row layer outside the fused kernels (%s, %d -> %d)' % (tuple(x.shape), k, n))
        return _RowsLinearMixed.apply(x, layer.weight, layer.bias, residual, pre_relu, post_relu, torch.bfloat16)
    if fused:
        return _RowsLinear.apply(x, layer.weight, layer.bias, residual, pre_relu, post_relu, out_dtype)
    h = torch.relu(x) if pre_relu else x
    y = torch.nn.functional.linear(h.to(layer.weight.dtype), layer.weight, layer.bias)
    if residual is not None:
        y = y + residual
    y = torch.relu(y) if post_relu else y
    return y.to(out_dtype)


class _TransformByIndex(torch.autograd.Function):
    """out[i] = R[idx[i]] @ p[i] + t[idx[i]] for a small table of 4x4 transforms (reconstruct_sequence /
    ego_motion_compensation, toolbox/register_utils.py:59-93).  The reference gathers one 4x4 matrix per point and runs
    a batched 3x3 matmul with batch = number of points (0.63 ms per call at 800 k points on MI355X); this is one
    streaming launch.  Gradients: points via the transposed rotations, the table via a per-row outer product reduced
    over the index (CSR segment sum)."""

    @staticmethod
    def forward(ctx, points, idx, tsfm):
        tsfm = tsfm.detach().clone()              # callers update the pose table in place afterwards (tpointnet.py:291-296)
        ctx.save_for_backward(points, idx, tsfm)
        return native.rigid_transform(points, idx, tsfm.reshape(-1, 16))

    @staticmethod
    def backward(ctx, g):
        points, idx, tsfm = ctx.saved_tensors
        g = g.contiguous()
        gp = gt = None
        if ctx.needs_input_grad[0]:
            rt = torch.zeros_like(tsfm)
            rt[:, :3, :3] = tsfm[:, :3, :3].transpose(1, 2)
            gp = native.rigid_transform(g, idx, rt.reshape(-1, 16).contiguous())
        if ctx.needs_input_grad[2]:
            hom = torch.cat([points, torch.ones_like(points[:, :1])], dim=1)                 # [N,4]
            outer = (g[:, :, None] * hom[:, None, :]).reshape(-1, 12)
            rows = scatter(outer, idx, dim=0, dim_size=tsfm.shape[0], reduce='sum')          # [K,12]
            gt = torch.zeros_like(tsfm)
            gt[:, :3, :] = rows.view(-1, 3, 4)
        return gp, None, gt


def transform_by_index(points, idx, tsfm):
    """points [N,3], idx [N] (any integer / float dtype holding integers), tsfm [K,4,4] -> [N,3] in points.dtype."""
    out = _TransformByIndex.apply(points.contiguous().float(), idx.to(torch.int32).contiguous(), tsfm.contiguous().float())
    return out if points.dtype == torch.float32 else out.to(points.dtype)


# ---- TubeNet slot algebra (csrc/tube.hip) -------------------------------------------------------------------------------------------
def _plan_sum(x, plan):
    """Per-slot sums of [N,c] f32 rows (c = 4 or 16: a width the segment kernels are instantiated for)."""
    if plan.small(x.shape[1]):
        return native.scatter_sum_small(x, plan.p2v, plan.m)
    return native.segment_sum(x, plan.seg_offsets, plan.order, plan.m)


def tube_rows(xyz, plan, slot_centre, n_frames):
    """[N,4] f32 rows of the TubeNet's positional embedding: point minus the centroid of its instance in the anchor frame, and
    t / n_frames (models/tpointnet.py:246-251).  No gradient: AlignNet hands the points over detached (models/alignnet.py:239)."""
    return native.tube_rows(xyz.detach().contiguous().float(), plan.p2v, slot_centre.detach().contiguous().float(), n_frames)


class _TubeCode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geo, motion, frame, n_frames):
        ctx.dims = (geo.shape[0], n_frames, geo.shape[1])
        return native.tube_code(geo.contiguous(), motion.contiguous(), frame.contiguous(), n_frames)

    @staticmethod
    def backward(ctx, grad):
        g_geo, g_motion, g_frame = native.tube_code_backward(grad.contiguous(), *ctx.dims)
        return g_geo, g_motion, g_frame, None


def tube_code(geo, motion, frame, n_frames):
    """Regressor input rows [K*T, 4c]: (geometry code, motion code) of the instance, frame code of the slot, frame code of the
    instance's anchor frame (models/tpointnet.py:259-262) -- one gather instead of three repeat_interleave and a cat."""
    return _TubeCode.apply(geo.float(), motion.float(), frame.float(), int(n_frames))


class _TubePose(torch.autograd.Function):
    """Everything of TPointNet.forward behind the regressor, and the bookkeeping of AlignNet's loop (csrc/tube.hip)."""

    @staticmethod
    def forward(ctx, pose_vec, rows, plan, remaining, total, slot_centre, weights, n_frames):
        pose_vec = pose_vec.contiguous().float()
        remaining = remaining.detach().contiguous().float().view(-1, 4, 4)
        total = total.detach().contiguous().float().view(-1, 4, 4) if total is not None else None
        pose_c, gt_c, step, rem_out, total_out, loss_rt, wsum = native.tube_pose_forward(pose_vec, remaining, total, slot_centre, weights, n_frames)
        count = plan.count()
        l12 = native.tube_finish(_plan_sum(native.tube_gap_forward(rows, plan.p2v, pose_c, gt_c), plan), count, weights, wsum)
        ctx.save_for_backward(pose_vec, rows, remaining, slot_centre, weights, pose_c, gt_c, count, wsum)
        ctx.plan, ctx.n_frames = plan, n_frames
        ctx.mark_non_differentiable(step, rem_out, total_out)
        return l12[0], l12[1], loss_rt[0], loss_rt[1], step, rem_out, total_out

    @staticmethod
    def backward(ctx, g_l1, g_l2, g_rot, g_trans, *unused):
        pose_vec, rows, remaining, slot_centre, weights, pose_c, gt_c, count, wsum = ctx.saved_tensors
        plan = ctx.plan
        c = lambda g, dt: g.contiguous().to(dt) if g is not None else None
        g16 = native.tube_gap_backward(rows, plan.p2v, pose_c, gt_c, weights, count, wsum, c(g_l1, torch.float32), c(g_l2, torch.float32))
        g_vec = native.tube_pose_backward(pose_vec, remaining, slot_centre, weights, wsum, _plan_sum(g16, plan), c(g_rot, torch.float64),
                                          c(g_trans, torch.float64), ctx.n_frames)
        return g_vec, None, None, None, None, None, None, None


def tube_pose(pose_vec, rows, plan, remaining, total, slot_centre, weights, n_frames):
    """pose_vec [K*T,7] -> (l1_loss, l2_loss, rot_loss, trans_loss, step [K*T,4,4], remaining' [K*T,4,4], total' [K*T,4,4]):
    models/tpointnet.py:264-296 and models/alignnet.py:257-263.  Only pose_vec receives a gradient (the reference's other inputs
    are detached or constant); the three pose tables come back without one -- no loss term reads them (libs/loss.py:253-263)."""
    return _TubePose.apply(pose_vec, rows, plan, remaining, total, slot_centre.detach().contiguous().float(), weights.detach().contiguous().float(),
                           int(n_frames))


_PREPARED = {}
_WEIGHT_EPOCH = 0


def weights_may_have_changed():
    """Forget every prepared (packed / split) copy of a weight.  A parameter's version counter does not see every writer: the fused
    optimizers (torch.optim.Adam(fused=True): `_fused_adam_`) update parameters WITHOUT incrementing it, so a copy keyed on
    (object, version, address) alone outlives an optimizer step.  MotionNet calls this at the start of every training-mode forward and on
    every train() / eval() switch, DataParallelStep after its optimizer step; call it yourself after writing weights through anything
    else that bypasses the counter (raw pointers, another fused optimizer outside these paths)."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


_ALL_OPTIMIZERS_WATCHED = False


def watch_all_optimizers():
    """Register -- once per process -- a post-step hook on EVERY torch optimizer (torch.optim.optimizer.register_optimizer_step_post_hook) that
    calls weights_may_have_changed(): whichever optimizer object writes the parameters, now or later, the prepared copies are dropped."""
    global _ALL_OPTIMIZERS_WATCHED
    if not _ALL_OPTIMIZERS_WATCHED:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(lambda *a, **k: weights_may_have_changed())
        _ALL_OPTIMIZERS_WATCHED = True


def _weight_key(weight):
    return (weight._version, weight.data_ptr(), _WEIGHT_EPOCH)


def _weight_ref(cache, key, weight):
    """A weak reference that removes the weight's entry (and frees its device copies) when the weight dies."""
    return weakref.ref(weight, lambda _r, c=cache, k=key: c.pop(k, None))


# ---- every prepared form a step needs, in one launch --------------------------------------------------------------------------------
# The per-weight caches below miss once per weight and optimizer step: ~90 host calls and launches of 10-20 us kernels at the start of a
# 'mixed' training step.  The first miss after the weights changed (a new weight epoch) instead re-prepares EVERY form the process has
# asked for so far (weights still alive, fp32, on that device) with one launch into buffers that stay allocated
# (native.prepare_weights_batch), and fills all three caches; a miss for any other reason (a version change inside an epoch, a weight
# seen for the first time) takes the per-weight path as before.  PCACC_BATCH_PREPARE=0 switches it off.
_BATCH_ON = os.environ.get('PCACC_BATCH_PREPARE', '1') != '0'
_BATCH_SEEN = {}            # (id(weight), kind) -> weak reference; kind 0 = split 3x3, 1 = split transposed 2x2, 2 = bf16 3x3, 3 = bf16 transposed 2x2
_BATCH_STATE = {}           # device index -> {'sig', 'jobs', 'n', 'blocks', 'forms': [(ref, kind, fwd, bwd)], 'epoch'}


def _batch_note(weight, kind):
    k = (id(weight), kind)
    if _BATCH_ON and k not in _BATCH_SEEN and weight.is_cuda and weight.dtype == torch.float32:
        _BATCH_SEEN[k] = weakref.ref(weight, lambda _r, k=k: _BATCH_SEEN.pop(k, None))


def _batch_build(dev, live):
    import numpy as np
    jobs = np.zeros((len(live), 16), dtype=np.int64)
    forms, b0 = [], 0
    for row, (ref, w, kind) in zip(jobs, live):
        st = w.stride()
        if kind == 1:
            a, b, kt = w.shape[0], w.shape[1], 1                                # c_in, c_up
            fwd = (torch.empty((2, 4 * b, a), dtype=torch.float16, device=dev), torch.empty((4 * b,), dtype=torch.float32, device=dev))
            bwd = (torch.empty((2, a, 4 * b), dtype=torch.float16, device=dev), torch.empty((a,), dtype=torch.float32, device=dev))
            strides, blocks = (st[0], st[1], 0, st[2], st[3]), 4 * b + a
        elif kind == 3:
            a, b, kt = w.shape[0], w.shape[1], 1                                # c_in, c_up
            fwd = torch.empty((4 * b, a), dtype=torch.bfloat16, device=dev)
            bwd = torch.empty((a, 4 * b), dtype=torch.bfloat16, device=dev)
            strides, blocks = (st[0], st[1], 0, st[2], st[3]), min(512, max(1, (8 * a * b + 2047) // 2048))
        else:
            a, b, kt = w.shape[0], w.shape[1], (3 if w.dim() == 5 else 1)       # c_out, c_in
            strides = tuple(st) if kt == 3 else (st[0], st[1], 0, st[2], st[3])
            if kind == 0:
                fwd = (torch.empty((2, kt * 9, a, b), dtype=torch.float16, device=dev), torch.empty((a,), dtype=torch.float32, device=dev))
                bwd = (torch.empty((2, kt * 9, b, a), dtype=torch.float16, device=dev), torch.empty((b,), dtype=torch.float32, device=dev))
                blocks = a + b
            else:
                fwd = torch.empty((kt * 9, a, b), dtype=torch.bfloat16, device=dev)
                bwd = torch.empty((kt * 9, b, a), dtype=torch.bfloat16, device=dev)
                blocks = min(512, max(1, (2 * kt * 9 * a * b + 2047) // 2048))
        ptrs = (fwd[0].data_ptr(), fwd[1].data_ptr(), bwd[0].data_ptr(), bwd[1].data_ptr()) if kind < 2 else (fwd.data_ptr(), 0, bwd.data_ptr(), 0)
        row[:] = (w.data_ptr(),) + ptrs + strides + (a, b, kt, kind, b0, blocks)
        b0 += blocks
        forms.append((ref, kind, fwd, bwd))
    return {'jobs': torch.from_numpy(jobs).to(dev), 'n': len(live), 'blocks': b0, 'forms': forms}


def _batch_refresh(dev):
    """-> True if every known form on `dev` was (re)prepared for the current weight epoch by this call."""
    state = _BATCH_STATE.get(dev.index)
    if state is not None and state['epoch'] == _WEIGHT_EPOCH:
        return False
    live = []
    for (_, kind), ref in list(_BATCH_SEEN.items()):
        w = ref()
        if w is not None and w.is_cuda and w.device == dev and w.dtype == torch.float32:
            live.append((ref, w, kind))
    if len(live) < 2:
        return False
    sig = tuple((id(w), kind, w.data_ptr(), w.stride()) for _, w, kind in live)
    if state is None or state['sig'] != sig:
        state = _batch_build(dev, live)
        state['sig'] = sig
        _BATCH_STATE[dev.index] = state
    native.prepare_weights_batch(state['jobs'], state['n'], state['blocks'])
    state['epoch'] = _WEIGHT_EPOCH
    caches = (_PREPARED_SPLIT, _PREPARED_UP, _PREPARED, _PREPARED_UPB)
    for ref, kind, fwd, bwd in state['forms']:
        w = ref()
        if w is not None:
            caches[kind][id(w)] = (_weight_ref(caches[kind], id(w), w), _weight_key(w), fwd, bwd)
    return True


def _batch_hit(cache, weight, kind):
    """The prepared forms of `weight` out of a batch refresh, or None (the caller then prepares this weight alone)."""
    if not (_BATCH_ON and (id(weight), kind) in _BATCH_SEEN and _batch_refresh(weight.device)):
        return None
    hit = cache.get(id(weight))
    if hit is not None and hit[0]() is weight and hit[1] == _weight_key(weight):
        return hit[2], hit[3]
    return None


def prepared_conv_weights(weight):
    """(forward form, data-gradient form) of a 3x3 / 3x3x3 weight for the MFMA kernels, prepared once per weight VERSION and weight
    epoch (weights_may_have_changed): the forward of a training step and its backward share one launch, evaluation passes reuse the forms
    until the optimizer (or a load_state_dict) writes the parameter again."""
    key = id(weight)
    hit = _PREPARED.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == _weight_key(weight):
        return hit[2], hit[3]
    got = _batch_hit(_PREPARED, weight, 2)
    if got is not None:
        return got
    _batch_note(weight, 2)
    w = weight.detach()
    if w.dtype != torch.float32:
        w = w.float()
    fwd, bwd = native.conv3x3_prepare_weights_pair(w)
    if len(_PREPARED) > 4096:
        _PREPARED.clear()
    _PREPARED[key] = (_weight_ref(_PREPARED, key, weight), _weight_key(weight), fwd, bwd)
    return fwd, bwd


class _Conv3x3(torch.autograd.Function):
    """3x3 (kt=1) / 3x3x3 (kt=3) convolution + bias + ReLU on bf16 channels-last rows [n_img, H, W, C] through the MFMA
    implicit-GEMM kernel (csrc/conv.hip).  Backward: ReLU mask, data gradient with the same kernel on mirrored /
    transposed weights, weight and bias gradients with the library's backward-weights kernel (fp32 results)."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias, frames, relu, premasked=False, input_relu=False):
        y = native.conv3x3(x_rows, prepared_conv_weights(weight)[0], bias.detach().float() if bias is not None else None, frames, relu)
        # premasked: every consumer of y hands back a gradient that is already zero where y <= 0 (pool_skip below; a following layer
        # called with input_relu): no ReLU pass here.  input_relu: x_rows is a ReLU output whose producer was called with premasked --
        # the data gradient leaves masked where x_rows <= 0 (in the kernel's epilogue where it can, by a threshold pass otherwise)
        ctx.save_for_backward(x_rows, weight, y if relu and not premasked else None)
        ctx.meta = (frames, relu and not premasked, bias is not None, bool(input_relu))
        return y

    @staticmethod
    def backward(ctx, gy):
        x_rows, weight, y = ctx.saved_tensors
        frames, relu, has_bias, input_relu = ctx.meta
        gy = gy.contiguous()
        gx = gw = gb = None
        kt = 3 if weight.dim() == 5 else 1
        o, i = weight.shape[0], weight.shape[1]
        h, w = x_rows.shape[1], x_rows.shape[2]
        need_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        lib_dgrad = ctx.needs_input_grad[0] and kt == 1 and not conv3x3_preferred(o, i, h, w)      # data gradient: channels exchanged
        deep_w = need_w and kt == 1 and not native.conv3x3_wgrad_supported(i, o) and native.conv3x3_wgrad_deep_supported(h, w, i, o)
        deep_d = ctx.needs_input_grad[0] and kt == 1 and o > 64 and native.conv3x3_deep_supported(h, w, o, i)
        # ReLU backward: the deep (MFMA-bound) kernels zero the gradient where the forward output is <= 0 while they stage it
        # (mask = y: the second read is free there); every other consumer gets the masked gradient as a tensor of its own
        mask = y if relu and (deep_w or not need_w) and (deep_d or not ctx.needs_input_grad[0]) else None
        if relu and mask is None:
            gy = torch.ops.aten.threshold_backward(gy, y, 0)
        gx_masked = False
        if ctx.needs_input_grad[0] and not lib_dgrad:
            if input_relu and mask is None and native.conv3x3_outmask_supported(h, w, o, i, kt):
                gx = native.conv3x3(gy, prepared_conv_weights(weight)[1], None, frames, False, out_mask=x_rows)
                gx_masked = True
            else:
                gx = native.conv3x3(gy, prepared_conv_weights(weight)[1], None, frames, False, mask=mask)
        if need_w and native.conv3x3_wgrad_supported(i, o):
            # weight gradient on the matrix cores too (one launch per frame tap); bias gradient = a column sum of dY
            if kt == 3:
                parts = [native.conv3x3_wgrad(gy, x_rows, frames, dt) for dt in (-1, 0, 1)]
                gw = torch.stack([p[0].view(o, 3, 3, i) for p in parts], dim=1).permute(0, 4, 1, 2, 3)    # [o, i, kt, 3, 3]
                gb = parts[1][1]                                                           # dt = 0 visits every frame
            else:
                gw, gb = native.conv3x3_wgrad(gy, x_rows)
                gw = gw.view(o, 3, 3, i).permute(0, 3, 1, 2)
            gw = gw.to(weight.dtype)
            gb = gb.clone() if has_bias and ctx.needs_input_grad[2] else None
            need_w = False
        elif deep_w:
            gw, gb = native.conv3x3_wgrad_deep(gy, x_rows, mask=mask)                      # deep layers: 64 x 64 weight blocks, strips
            gw = gw.view(o, 3, 3, i).permute(0, 3, 1, 2).to(weight.dtype)
            gb = gb if has_bias and ctx.needs_input_grad[2] else None
            need_w = False
        if need_w or lib_dgrad:
            xin = _stack_frames(x_rows, frames) if kt == 3 else x_rows
            w2 = weight.detach().permute(0, 2, 1, 3, 4).reshape(o, 3 * i, 3, 3) if kt == 3 else weight.detach()
            gx2, gw2, gb2 = torch.ops.aten.convolution_backward(
                gy.permute(0, 3, 1, 2), xin.permute(0, 3, 1, 2), w2.to(torch.bfloat16).contiguous(memory_format=torch.channels_last),
                [o] if has_bias else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [lib_dgrad, need_w, has_bias and need_w])
            if lib_dgrad:
                gx = gx2.permute(0, 2, 3, 1).contiguous()
            if need_w:
                gw = gw2.float()
                if kt == 3:
                    gw = gw.reshape(o, 3, i, 3, 3).permute(0, 2, 1, 3, 4)
                gw = gw.reshape(weight.shape)
                gb = gb2.float() if has_bias else None
        if input_relu and gx is not None and not gx_masked:                               # the producer relies on it: mask here if the kernel did not
            gx = torch.ops.aten.threshold_backward(gx, x_rows, 0)
        return gx, gw, gb, None, None, None, None


_PREPARED_SPLIT = {}


def prepared_conv_weights_split(weight):
    """(forward form, data-gradient form) of a 3x3 / 3x3x3 weight as fp16 hi / lo planes + row scales (fp32x3 mode), once per weight version."""
    key = id(weight)
    hit = _PREPARED_SPLIT.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == _weight_key(weight):
        return hit[2], hit[3]
    got = _batch_hit(_PREPARED_SPLIT, weight, 0)
    if got is not None:
        return got
    _batch_note(weight, 0)
    w = weight.detach()
    if w.dtype != torch.float32:
        w = w.float()
    fwd, bwd = native.conv3x3_split_prepare_weights(w)
    if len(_PREPARED_SPLIT) > 4096:
        _PREPARED_SPLIT.clear()
    _PREPARED_SPLIT[key] = (_weight_ref(_PREPARED_SPLIT, key, weight), _weight_key(weight), fwd, bwd)
    return fwd, bwd


class _Conv3x3Split(torch.autograd.Function):
    """3x3 (kt=1) / 3x3x3 (kt=3) convolution + bias + ReLU on fp32 channels-last rows [n_img, H, W, C] at fp32 accuracy on the 16-bit
    matrix cores (csrc/conv_split.hip).  Backward: data gradient = the same kernel on mirrored / transposed weights, weight and bias
    gradients by the split weight-gradient kernel; both read the gradient through the ReLU mask (the forward output) while staging.
    The absolute maxima the kernels scale their operands by are taken once per tensor (x in forward, reused by the weight gradient;
    the incoming gradient once for both backward kernels)."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias, frames, relu, premasked=False, input_relu=False):
        x_amax = amax_of(x_rows)
        y, y_amax = native.conv3x3_split(x_rows, prepared_conv_weights_split(weight)[0], bias.detach().float() if bias is not None else None, frames,
                                         relu, amax=x_amax, want_amax=True)
        set_amax_tag(y, y_amax)
        # premasked: every consumer of y hands back a gradient that is already zero where y <= 0 (_PoolSkip, a following layer called with
        # input_relu): no mask reads in the backward.  input_relu: x_rows is a ReLU output whose producer was called with premasked -- the
        # data gradient is stored masked where x_rows <= 0 (one read of x in the epilogue instead of two mask reads in the producer's backward)
        ctx.save_for_backward(x_rows, weight, y if relu and not premasked else None, x_amax)
        ctx.meta = (frames, bias is not None, bool(input_relu))
        return y

    @staticmethod
    def backward(ctx, gy):
        x_rows, weight, y, x_amax = ctx.saved_tensors
        frames, has_bias, input_relu = ctx.meta
        gy = dense(gy)
        if gy.dtype != torch.float32:
            gy = gy.float()
        gx = gw = gb = None
        kt = 3 if weight.dim() == 5 else 1
        o, i = weight.shape[0], weight.shape[1]
        g_amax = amax_of(gy)                                   # of the unmasked gradient: an upper bound is all the scale needs
        if ctx.needs_input_grad[0]:
            gx, gx_amax = native.conv3x3_split(gy, prepared_conv_weights_split(weight)[1], None, frames, False, mask=y, amax=g_amax, want_amax=True,
                                               out_mask=x_rows if input_relu else None)
            set_amax_tag(gx, gx_amax)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            if kt == 3:
                parts = [native.conv3x3_wgrad_split(gy, x_rows, frames, dt, mask=y, dy_amax=g_amax, x_amax=x_amax) for dt in (-1, 0, 1)]
                gw = torch.stack([p[0].view(o, 3, 3, i) for p in parts], dim=1).permute(0, 4, 1, 2, 3)    # [o, i, kt, 3, 3]
                gb = parts[1][1]                                                           # dt = 0 visits every frame
            else:
                gw, gb = native.conv3x3_wgrad_split(gy, x_rows, mask=y, dy_amax=g_amax, x_amax=x_amax)
                gw = gw.view(o, 3, 3, i).permute(0, 3, 1, 2)
            gw = gw.to(weight.dtype)
            gb = gb if has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None, None


class _Conv3x3Mixed(_Conv3x3):
    """_Conv3x3Split's forward on the fp32 twin of a bf16 shadow, _Conv3x3's backward on the shadows ('mixed' compute mode): fp32-accurate
    forward values, bf16 data / weight gradients (fp32 accumulation, fp32 weight gradients)."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias, frames, relu, premasked=False, input_relu=False):
        x32 = twin(x_rows)
        if not x32.is_contiguous():
            raise native.NativeError('mixed conv3x3: the twin of the input rows must be contiguous')
        y32, y_amax, y16 = native.conv3x3_split(x32, prepared_conv_weights_split(weight)[0], bias.detach().float() if bias is not None else None, frames,
                                                relu, amax=amax_of(x32), want_amax=True, want_bf16=True)      # the shadow from the same epilogue
        set_amax_tag(y32, y_amax)
        y = shadow(y32, y16)
        ctx.save_for_backward(x_rows, weight, y if relu and not premasked else None)
        ctx.meta = (frames, relu and not premasked, bias is not None, bool(input_relu))
        return y


class _Conv3x3CatMixed(_Conv3x3):
    """conv1(cat(up, skip)) of a decoder stage (models/unet.py:101-113) in the 'mixed' mode: the forward reads the two fp32 twins in place
    (pcacc_conv3x3_split_cat -- the fp32 concatenation is never written), autograd sees the convolution of the bf16 concatenation `x_rows` (the
    shadow the bf16 backward reads; it has no twin of its own)."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias, relu, premasked, a32, b32):
        amax = torch.maximum(amax_of(a32), amax_of(b32))
        y32, y_amax, y16 = native.conv3x3_split_cat(a32, b32, amax, prepared_conv_weights_split(weight)[0], bias.detach().float() if bias is not None else None,
                                                    relu, want_bf16=True)
        set_amax_tag(y32, y_amax)
        y = shadow(y32, y16)
        ctx.save_for_backward(x_rows, weight, y if relu and not premasked else None)
        ctx.meta = (1, relu and not premasked, bias is not None, False)
        return y

    @staticmethod
    def backward(ctx, gy):
        return _Conv3x3.backward(ctx, gy)[:3] + (None, None, None, None)


def conv3x3_cat(a, b, conv, relu=False, premasked=False):
    """`relu?(conv(cat((a, b), 1)))` for a decoder stage.  'mixed' mode on the GPU: the fp32 forward reads the two twins in place, only the bf16
    shadows are concatenated (for the bf16 backward); everything else: cat_maps + conv3x3."""
    if (_MIXED and a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 4 and a.shape[1] % 32 == 0
            and b.shape[1] % 32 == 0 and os.environ.get('PCACC_CONV_CAT', '1') != '0' and conv3x3_native(a.new_empty((0, a.shape[1] + b.shape[1]) + tuple(a.shape[2:])), conv) == 'mixed'):
        ar, br = a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)
        a32, b32 = twin(ar), twin(br)
        if ar.is_contiguous() and br.is_contiguous() and a32.is_contiguous() and b32.is_contiguous():
            x16 = _cat2(ar, br, -1)                             # the shadow: what the weight gradient reads, whose gradient is split between a and b
            if _POISON:
                with torch.no_grad():
                    x16.fill_(float('nan'))
            return _Conv3x3CatMixed.apply(x16, conv.weight, conv.bias, bool(relu), bool(premasked), a32, b32).permute(0, 3, 1, 2)
    return conv3x3(cat_maps((a, b), 1), conv, relu=relu, premasked=premasked)


_PREPARED_UP = {}


def prepared_upconv_weights_split(weight):
    key = id(weight)
    hit = _PREPARED_UP.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == _weight_key(weight):
        return hit[2], hit[3]
    got = _batch_hit(_PREPARED_UP, weight, 1)
    if got is not None:
        return got
    _batch_note(weight, 1)
    fwd, bwd = native.upconv2x2_split_prepare_weights(weight.detach())
    if len(_PREPARED_UP) > 4096:
        _PREPARED_UP.clear()
    _PREPARED_UP[key] = (_weight_ref(_PREPARED_UP, key, weight), _weight_key(weight), fwd, bwd)
    return fwd, bwd


class _UpConv2x2Split(torch.autograd.Function):
    """nn.ConvTranspose2d(kernel 2, stride 2) on fp32 channels-last rows in the fp32x3 mode (csrc/conv_split.hip, 1-tap kernels):
    [n,h,w,c_in] -> [n,2h,2w,c_up]."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias):
        x_amax = amax_of(x_rows)
        y, y_amax = native.upconv2x2_split(x_rows, x_amax, prepared_upconv_weights_split(weight)[0], bias.detach() if bias is not None else None, 0)
        set_amax_tag(y, y_amax)
        ctx.save_for_backward(x_rows, weight, x_amax)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x_rows, weight, x_amax = ctx.saved_tensors
        gy = dense(gy)
        if gy.dtype != torch.float32:
            gy = gy.float()
        g_amax = amax_of(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx, gx_amax = native.upconv2x2_split(gy, g_amax, prepared_upconv_weights_split(weight)[1], None, 1)
            set_amax_tag(gx, gx_amax)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = native.upconv2x2_wgrad_split(gy, g_amax, x_rows, x_amax)
            gw = gw.to(weight.dtype)
            gb = gb if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


_PREPARED_UPB = {}


def prepared_upconv_weights_bf16(weight):
    """(forward form bf16 [4 c_up, c_in], data-gradient form bf16 [c_in, 4 c_up]) of a transposed 2 x 2 weight (csrc/upconv_bf16.hip)."""
    key = id(weight)
    hit = _PREPARED_UPB.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == _weight_key(weight):
        return hit[2], hit[3]
    got = _batch_hit(_PREPARED_UPB, weight, 3)
    if got is not None:
        return got
    _batch_note(weight, 3)
    fwd, bwd = native.upconv2x2_bf16_prepare_weights(weight.detach())
    if len(_PREPARED_UPB) > 4096:
        _PREPARED_UPB.clear()
    _PREPARED_UPB[key] = (_weight_ref(_PREPARED_UPB, key, weight), _weight_key(weight), fwd, bwd)
    return fwd, bwd


def _upconv_bf16_backward(ctx, gy, x_rows, weight):
    """Data, weight and bias gradient of the transposed 2 x 2 convolution from bf16 rows.  gy may be a channel slice of the decoder's concatenation
    gradient (read in place through its pixel pitch).  Own kernels (csrc/upconv_bf16.hip) where the channel counts allow, else the library."""
    c_in, c_up = weight.shape[0], weight.shape[1]
    if gy.dtype != torch.bfloat16:
        gy = gy.to(torch.bfloat16)
    if native.upconv2x2_bf16_supported(c_in, c_up) and weight.dtype == torch.float32 and os.environ.get('PCACC_UPCONV_BF16', '1') != '0':
        gx = native.upconv2x2_bf16(gy, prepared_upconv_weights_bf16(weight)[1], None, 1) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = native.upconv2x2_bf16_wgrad(gy, x_rows, want_bias=ctx.has_bias and ctx.needs_input_grad[2], like=weight)    # in the weight's layout
        return gx, gw, gb
    gy = gy.contiguous()
    w16 = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gx, gw, _ = torch.ops.aten.convolution_backward(
        gy.permute(0, 3, 1, 2), x_rows.permute(0, 3, 1, 2), w16, None, [2, 2], [0, 0], [1, 1], True, [0, 0], 1,
        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
    # bias gradient = column sums of the gradient rows, fp32 accumulation (the library's own bias path reduces the channels-last map over
    # three dimensions: 0.32 ms for the [20, 256, 36, 36] map alone)
    gb = gy.reshape(-1, gy.shape[-1]).sum(0, dtype=torch.float32) if ctx.has_bias and ctx.needs_input_grad[2] else None
    if gx is not None:
        gx = gx.permute(0, 2, 3, 1).contiguous()
    return gx, (gw.float().contiguous(memory_format=torch.channels_last) if gw is not None else None), gb


class _UpConv2x2Bf16(torch.autograd.Function):
    """nn.ConvTranspose2d(kernel 2, stride 2) on bf16 channels-last rows [n,h,w,c_in] -> [n,2h,2w,c_up] (bf16 compute mode): forward, data and
    weight gradient on the bf16 matrix cores (csrc/upconv_bf16.hip), fp32 accumulation, fp32 weight / bias gradients."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias):
        ctx.save_for_backward(x_rows, weight)
        ctx.has_bias = bias is not None
        return native.upconv2x2_bf16(x_rows, prepared_upconv_weights_bf16(weight)[0], bias.detach().float() if bias is not None else None, 0)

    @staticmethod
    def backward(ctx, gy):
        x_rows, weight = ctx.saved_tensors
        return _upconv_bf16_backward(ctx, gy, x_rows, weight)


class _UpConv2x2Mixed(torch.autograd.Function):
    """_UpConv2x2Split's forward on the twin, bf16 backward ('mixed' mode: csrc/upconv_bf16.hip)."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias):
        x32 = twin(x_rows)
        y32, y_amax, y16 = native.upconv2x2_split(x32, amax_of(x32), prepared_upconv_weights_split(weight)[0], bias.detach() if bias is not None else None, 0,
                                                  want_bf16=True)
        set_amax_tag(y32, y_amax)
        ctx.save_for_backward(x_rows, weight)
        ctx.has_bias = bias is not None
        return shadow(y32, y16)

    @staticmethod
    def backward(ctx, gy):
        x_rows, weight = ctx.saved_tensors
        return _upconv_bf16_backward(ctx, gy, x_rows, weight)


class _UpConvCatMixed(torch.autograd.Function):
    """cat(upconv(x), skip) of a decoder stage (models/unet.py:101-113) in the 'mixed' mode: the transposed convolution writes its fp32 result and
    the bf16 shadow straight into the first halves of the two concatenation buffers (pcacc_upconv2x2_split_dual with a pixel pitch); only the
    skip halves are copied.  Backward: the up-sampled half of the buffer's gradient goes through the transposed convolution's bf16 backward, the
    skip half is handed on as the channel slice it is (the encoder's pool tail reads it in place).  Opt-in (PCACC_UPCONV_CAT=1): measured neutral."""

    @staticmethod
    def forward(ctx, x_rows, skip_rows, weight, bias):
        x32, s32 = twin(x_rows), twin(skip_rows)
        n, h, w, _ = x_rows.shape
        c_up, c_skip = weight.shape[1], skip_rows.shape[3]
        if c_skip != c_up or tuple(skip_rows.shape[:3]) != (n, 2 * h, 2 * w):
            raise native.NativeError('upconv + concatenation: the skip map must have the up-sampled map\'s shape')
        buf32 = torch.empty((n, 2 * h, 2 * w, 2 * c_up), dtype=torch.float32, device=x_rows.device)
        buf16 = torch.empty((n, 2 * h, 2 * w, 2 * c_up), dtype=torch.bfloat16, device=x_rows.device)
        _, up_amax, _ = native.upconv2x2_split(x32, amax_of(x32), prepared_upconv_weights_split(weight)[0], bias.detach() if bias is not None else None, 0,
                                               want_bf16=True, into=(buf32, buf16))
        buf32[..., c_up:].copy_(s32)
        if _POISON:
            buf16.fill_(float('nan'))
        else:
            buf16[..., c_up:].copy_(skip_rows)
        set_amax_tag(buf32, torch.maximum(up_amax, amax_of(s32)))
        ctx.save_for_backward(x_rows, weight)
        ctx.has_bias, ctx.c_up = bias is not None, c_up
        return shadow(buf32, buf16)

    @staticmethod
    def backward(ctx, g):
        x_rows, weight = ctx.saved_tensors
        c_up = ctx.c_up
        gy = g[..., :c_up].contiguous()
        w16 = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx, gw, _ = torch.ops.aten.convolution_backward(
            gy.permute(0, 3, 1, 2), x_rows.permute(0, 3, 1, 2), w16, None, [2, 2], [0, 0], [1, 1], True, [0, 0], 1,
            [ctx.needs_input_grad[0], ctx.needs_input_grad[2], False])
        gb = gy.reshape(-1, c_up).sum(0, dtype=torch.float32) if ctx.has_bias and ctx.needs_input_grad[3] else None
        if gx is not None:
            gx = gx.permute(0, 2, 3, 1).contiguous()
        return (gx, g[..., c_up:] if ctx.needs_input_grad[1] else None,
                gw.float().contiguous(memory_format=torch.channels_last) if gw is not None else None, gb)


def upconv_cat(from_up, from_down, conv):
    """cat((conv(from_up), from_down), 1) for a decoder stage.  PCACC_UPCONV_CAT=1 ('mixed' mode): without the copy of the up-sampled half
    (_UpConvCatMixed) -- opt-in: measured neutral (36.2 / 36.2 / 39.1 ms without, 37.6 / 37.0 / 38.5 with: the strided copies of the skip halves cost
    what the saved half of the concatenation cost)."""
    if (_MIXED and from_up.is_cuda and from_up.dtype == torch.bfloat16 and from_down.dtype == torch.bfloat16
            and conv.kernel_size == (2, 2) and conv.stride == (2, 2) and conv.padding == (0, 0) and conv.output_padding == (0, 0) and conv.groups == 1
            and conv.dilation == (1, 1) and conv.weight.dtype == torch.float32 and conv.out_channels == from_down.shape[1]
            and os.environ.get('PCACC_UPCONV_CAT', '0') == '1'):
        xr, sr = from_up.permute(0, 2, 3, 1), from_down.permute(0, 2, 3, 1)
        if xr.is_contiguous() and sr.is_contiguous() and native.upconv2x2_split_supported(from_up.shape[-2], from_up.shape[-1], conv.in_channels, conv.out_channels):
            return _UpConvCatMixed.apply(xr, sr, conv.weight, conv.bias).permute(0, 3, 1, 2)
    return cat_maps((upconv2x2(from_up, conv), from_down), 1)


def upconv2x2(x, conv):
    """`conv(x)` for the decoders' nn.ConvTranspose2d(kernel 2, stride 2) (models/unet.py:22-30): fp32 channels-last maps in the fp32x3 mode
    run the split kernels, everything else the module (library)."""
    plain = (conv.kernel_size == (2, 2) and conv.stride == (2, 2) and conv.padding == (0, 0) and conv.output_padding == (0, 0) and conv.groups == 1
             and conv.dilation == (1, 1) and conv.weight.dtype == torch.float32)
    if _MIXED and x.is_cuda and x.dtype == torch.bfloat16:
        xr = x.permute(0, 2, 3, 1)
        if not (plain and xr.is_contiguous() and native.upconv2x2_split_supported(x.shape[-2], x.shape[-1], conv.in_channels, conv.out_channels)):
            raise native.NativeError('mixed mode: transposed convolution outside the split kernels (%s)' % (tuple(x.shape),))
        return _UpConv2x2Mixed.apply(xr, conv.weight, conv.bias).permute(0, 3, 1, 2)
    if (_SPLIT and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and plain
            and native.upconv2x2_split_supported(x.shape[-2], x.shape[-1], conv.in_channels, conv.out_channels)):
        xr = x.permute(0, 2, 3, 1)
        if xr.is_contiguous():
            set_amax_tag(xr, amax_of(x))
            y = _UpConv2x2Split.apply(xr, conv.weight, conv.bias)
            return carry_amax(y, y.permute(0, 3, 1, 2))
    if (x.is_cuda and plain and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16))
            and native.upconv2x2_bf16_supported(conv.in_channels, conv.out_channels) and os.environ.get('PCACC_UPCONV_BF16', '1') != '0'):
        xr = x.permute(0, 2, 3, 1)                              # bf16 compute mode: own kernels, no library call, no layout copies
        if xr.is_contiguous():
            return _UpConv2x2Bf16.apply(xr if xr.dtype == torch.bfloat16 else xr.to(torch.bfloat16), conv.weight, conv.bias).permute(0, 3, 1, 2)
    return conv(x)


class _HeadConv3x3(torch.autograd.Function):
    """Conv2d(c_in, c_out <= 4, 3, padding 1) on channels-last rows [n,H,W,c_in] (f32 or bf16) -> f32 logits (csrc/head_conv.hip): the fg / bg
    head's last layer (models/unet.py:259-277), exact fp32 arithmetic in every compute mode."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias):
        ctx.save_for_backward(x_rows, weight)
        ctx.has_bias = bias is not None
        return native.head_conv3x3_forward(twin_or_self(x_rows), weight.detach(), bias.detach() if bias is not None else None)

    @staticmethod
    def backward(ctx, gy):
        x_rows, weight = ctx.saved_tensors
        gy = gy.contiguous().float()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = native.head_conv3x3_dgrad(gy, weight.detach(), x_rows.shape[3], x_rows.dtype)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = native.head_conv3x3_wgrad(gy, x_rows, want_bias=ctx.has_bias)
            gw = gw.to(weight.dtype)
        return gx, gw, gb


def head_conv3x3_available(x, conv):
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32
            and native.head_conv3x3_supported(conv.in_channels, conv.out_channels) and x.permute(0, 2, 3, 1).is_contiguous())


def _stack_frames(rows, frames):
    """[B*T, H, W, C] -> [B*T, H, W, 3C]: frames t-1, t, t+1 side by side (zeros outside the sequence)."""
    n, h, w, c = rows.shape
    r = rows.view(n // frames, frames, h, w, c)
    prev = torch.nn.functional.pad(r[:, :-1], (0, 0, 0, 0, 0, 0, 1, 0))
    nxt = torch.nn.functional.pad(r[:, 1:], (0, 0, 0, 0, 0, 0, 0, 1))
    return torch.cat([prev, r, nxt], dim=-1).view(n, h, w, 3 * c)


def conv3x3_preferred(c_in, c_out, h=None, w=None):
    """Layer shapes the hand-written kernels take: everything with c_in <= 64 (weights resident in LDS, csrc/conv.hip -- the layers
    bound by HBM and LDS rather than by the matrix cores) and, given the image size, the deep layers on small images (c_in >= 128,
    strips of consecutive pixels with K-deep tiling, csrc/conv_deep.hip)."""
    if not native.conv3x3_supported(c_in, c_out):
        return False
    if c_in <= 64:
        return True
    return h is not None and native.conv3x3_deep_supported(h, w, c_in, c_out)


def conv3x3_available(x, weight):
    """'bf16' / 'split' when `x` (NCHW view) and the 3x3 weight take an MFMA path, else None (falsy): GPU, supported channel counts, and
    bf16 compute (bf16 kernels) or fp32 rows in the fp32x3 mode (split-bf16 kernels, csrc/conv_split.hip)."""
    if not x.is_cuda:
        return None
    if _MIXED and x.dtype == torch.bfloat16:
        # a shadow inside a mixed segment: forward on the split kernels, backward on the bf16 kernels -- both must take the layer
        if not (weight.dtype == torch.float32 and native.conv3x3_split_supported(x.shape[-2], x.shape[-1], weight.shape[1], weight.shape[0])):
            raise native.NativeError('mixed mode: 3x3 convolution outside the split kernels (%s, weight %s)' % (tuple(x.shape), tuple(weight.shape)))
        return 'mixed'
    if x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16):
        return 'bf16' if conv3x3_preferred(weight.shape[1], weight.shape[0], x.shape[-2], x.shape[-1]) else None
    if _SPLIT and x.dtype == torch.float32 and weight.dtype == torch.float32 \
            and native.conv3x3_split_supported(x.shape[-2], x.shape[-1], weight.shape[1], weight.shape[0]):
        return 'split'
    return None


def conv3x3_rows(x_rows, weight, bias, frames=1, relu=False, premasked=False, input_relu=False):
    """x_rows [n_img, H, W, C_in] -> [n_img, H, W, C_out] (bf16; f32 in the fp32x3 mode).  weight [O,I,3,3] (frames ignored) or [O,I,3,3,3]."""
    if _MIXED and x_rows.dtype == torch.bfloat16:
        return _Conv3x3Mixed.apply(x_rows.contiguous(), weight, bias, int(frames), bool(relu), bool(premasked), bool(input_relu))
    if _SPLIT and x_rows.dtype == torch.float32 and not torch.is_autocast_enabled():
        xc = x_rows.contiguous()
        if xc is not x_rows:
            carry_amax(x_rows, xc)
        return _Conv3x3Split.apply(xc, weight, bias, int(frames), bool(relu), bool(premasked), bool(input_relu))
    if x_rows.dtype != torch.bfloat16:
        x_rows = x_rows.to(torch.bfloat16)
    return _Conv3x3.apply(x_rows.contiguous(), weight, bias, int(frames), bool(relu), bool(premasked), bool(input_relu))


def conv3x3_native(x, conv):
    """'bf16' / 'split' when conv3x3(x, conv) takes the MFMA kernels (plain 3x3 / stride 1 / padding 1, see conv3x3_available), else None."""
    if conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1:
        return conv3x3_available(x, conv.weight)
    return None


def conv_pair_fusable(x, conv1, conv2):
    """conv1 -> ReLU -> conv2 (models/unet.py:45-71) where the first ReLU's backward can ride in conv2's data-gradient epilogue: both layers on
    the bf16 MFMA kernels, or both on the fp32x3 kernels.  The caller then passes premasked=True to conv1 and input_relu=True to conv2 -- conv1's output must have no
    other consumer."""
    mode = conv3x3_native(x, conv1)
    if mode not in ('bf16', 'split', 'mixed') or conv2.in_channels != conv1.out_channels or os.environ.get('PCACC_CONV_PAIR', '1') == '0':      # A/B switch
        return False
    probe = torch.empty((0, conv1.out_channels, x.shape[-2], x.shape[-1]), dtype=torch.float32 if mode == 'split' else torch.bfloat16, device=x.device)
    return conv3x3_native(probe, conv2) == mode


def conv3x3(x, conv, relu=False, premasked=False, input_relu=False):
    """`relu?(conv(x))` for an nn.Conv2d(3x3, stride 1, padding 1) on an NCHW tensor; channels-last bf16 inputs on the GPU go
    through the MFMA kernel, everything else through the library with the same semantics.  premasked / input_relu: see
    conv_pair_fusable (only with it)."""
    if not relu and head_conv3x3_available(x, conv):            # c_out <= 4: the streamed fp32 kernels, f32 logits out
        return _HeadConv3x3.apply(x.permute(0, 2, 3, 1), conv.weight, conv.bias).permute(0, 3, 1, 2)
    mode = conv3x3_native(x, conv)
    if mode:
        xr = x.permute(0, 2, 3, 1)
        if mode == 'split' and xr.is_contiguous():
            set_amax_tag(xr, amax_of(x))                       # measured (or inherited) on the caller's tensor: its next reader finds it there
        y = conv3x3_rows(xr, conv.weight, conv.bias, 1, relu, premasked=premasked, input_relu=input_relu)
        return carry_amax(y, y.permute(0, 3, 1, 2))
    assert not premasked and not input_relu
    y = conv(x)
    return torch.relu(y) if relu else y


class _PoolSkip(torch.autograd.Function):
    """y [n, H, W, C] bf16 (f32 in the fp32x3 mode) rows = a ReLU output -> (2x2 max-pool of y, y).  Backward: ONE pass that un-pools the
    first gradient, adds the second and zeroes the sum where y <= 0 (csrc/pool.hip) -- in place of max_pool2d_backward + add +
    threshold_backward; on f32 rows the pass also leaves the gradient's maximum for the split kernels that read it."""

    @staticmethod
    def forward(ctx, y_rows):
        if _MIXED and y_rows.dtype == torch.bfloat16:              # shadow: pooled twin from the fp32 twin, its shadow registered; the skip is a view
            y32 = twin(y_rows)
            ctx.save_for_backward(y32)                             # the forward's own values pick the windows' winners in the backward too
            ctx.mixed = True
            return shadow(carry_amax(y32, native.maxpool2x2(y32))), y_rows.view_as(y_rows)
        ctx.save_for_backward(y_rows)
        ctx.mixed = False
        pooled = native.maxpool2x2(y_rows)
        skip = y_rows.view_as(y_rows)
        if y_rows.dtype == torch.float32:
            carry_amax(y_rows, pooled)                          # window maxima of y: its bound holds
            carry_amax(y_rows, skip)
        return pooled, skip

    @staticmethod
    def backward(ctx, g_pool, g_skip):
        y_rows, = ctx.saved_tensors
        if ctx.mixed:                                              # y f32, gradients bf16 (pcacc_pool_skip_relu_backward_strided_y32)
            b16 = lambda g: g.to(torch.bfloat16) if g is not None else None
            return native.pool_skip_relu_backward(y_rows, b16(g_pool.contiguous() if g_pool is not None else None), b16(g_skip))
        c = lambda g: g.contiguous().to(y_rows.dtype) if g is not None else None
        # the skip gradient usually arrives as a channel slice of the decoder's concatenation gradient: the kernel reads it in place
        gs = g_skip.to(y_rows.dtype) if g_skip is not None else None
        if y_rows.dtype == torch.float32:
            g, g_amax = native.pool_skip_relu_backward(y_rows, c(g_pool), gs, want_amax=True)
            return set_amax_tag(g, g_amax)
        return native.pool_skip_relu_backward(y_rows, c(g_pool), gs)


def conv3x3_relu_pool(x, conv, input_relu=False):
    """(max_pool2d(y, 2), y) with y = relu(conv(x)) -- the tail of models/unet.py:60-71 -- when conv3x3_native(x, conv) and the
    channel count suits the pooling kernel; the ReLU's backward is folded into the pooling's (one pass, see _PoolSkip)."""
    y = conv3x3_rows(x.permute(0, 2, 3, 1), conv.weight, conv.bias, 1, True, premasked=True, input_relu=input_relu)
    pooled, skip = _PoolSkip.apply(y)
    return pooled.permute(0, 3, 1, 2), skip.permute(0, 3, 1, 2)


class _SparseConv3x3(torch.autograd.Function):
    """A 3x3 convolution (padding 1) evaluated ONLY at selected cells of a channels-last map: out[j] = bias + sum over the 3x3 neighbourhood of cell
    cells[j] of W[:, :, dy, dx] . h[neighbour] -- the rows a dense convolution followed by a row gather would give.  The ego head reads the last layer of
    its feature head at <= 1024 key points per frame (models/motionnet.py:196-201 computes the whole [B*T, 64, Ny, Nx] map, models/egomotion.py:156-166
    samples it): 32 768 of 1 658 880 cells of a 4-sequence step, i.e. 2.4 of 122 GFLOP, and the backward is two small GEMMs plus a scatter-add
    instead of a dense data / weight gradient.  fp32 throughout (plain library GEMMs: [K, 9 C] x [9 C, O])."""

    @staticmethod
    def forward(ctx, h_rows, weight, bias, cells):
        n, H, W, C = h_rows.shape
        cells = cells.reshape(-1).long()
        x, y, img = cells % W, (cells // W) % H, cells // (W * H)
        d = torch.arange(-1, 2, device=cells.device)
        yy, xx = (y[:, None, None] + d[None, :, None]).expand(-1, 3, 3), (x[:, None, None] + d[None, None, :]).expand(-1, 3, 3)
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        idx = torch.where(ok, (img[:, None, None] * H + yy) * W + xx, torch.full_like(yy, -1)).reshape(-1).to(torch.int32)     # [K * 9], -1 = zero padding
        patches = native.gather_rows(h_rows.reshape(-1, C), idx).view(-1, 9 * C)                                              # [K, 9 C] (tap-major)
        w2 = weight.detach().float().permute(2, 3, 1, 0).reshape(9 * C, -1)                                                     # [(dy, dx, ci), co]
        out = torch.addmm(bias.detach().float(), patches, w2) if bias is not None else patches @ w2
        ctx.save_for_backward(patches, w2, idx)
        ctx.meta = (tuple(h_rows.shape), tuple(weight.shape), bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        patches, w2, idx = ctx.saved_tensors
        (n, H, W, C), wshape, has_bias = ctx.meta
        g = g.contiguous().float()
        gh = gw = gb = None
        if ctx.needs_input_grad[0]:
            gp = (g @ w2.t()).view(-1, C)                                                      # [K * 9, C] rows of the neighbourhoods' gradients
            cells = n * H * W
            if DETERMINISTIC and C % 4 == 0:
                # overlapping neighbourhoods add up in ascending tap-row order: a CSR over the cells (taps outside the map go to an extra last cell that is
                # dropped) + the atomic-free segment sum -- index_add_ adds with atomics, in arrival order
                key = torch.where(idx >= 0, idx, cells).to(torch.int32)
                offs, order = native.csr_build(key, cells + 1)
                gh = native.segment_sum(gp.contiguous(), offs, order, cells + 1)[:cells].view(n, H, W, C)
            else:
                gh = torch.zeros((cells + 1, C), dtype=torch.float32, device=g.device)         # last row: where the taps outside the map add up (dropped)
                gh.index_add_(0, torch.where(idx >= 0, idx, cells).long(), gp)                 # overlapping neighbourhoods add up
                gh = gh[:cells].view(n, H, W, C)
        if ctx.needs_input_grad[1]:
            k = g.shape[0]
            if k % 16 == 0 and k >= 4096:
                # [9 C, K] x [K, O] with K = 32 768 key points: the library runs one long reduction per output tile (142 us); sixteen slices as one batched
                # product and a sum take ~30 us
                gw2 = torch.bmm(patches.view(16, k // 16, -1).transpose(1, 2), g.view(16, k // 16, -1)).sum(0)
            else:
                gw2 = patches.t() @ g
            gw = gw2.view(3, 3, wshape[1], wshape[0]).permute(3, 2, 0, 1).contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gh, gw, gb, None


class SparseConvRows(object):
    """The rows of `conv(h)` as a lazily evaluated table: `.at(cells)` computes the convolution at those cells only (_SparseConv3x3), `.dense()`
    the whole map as rows.  h: NCHW view of a channels-last fp32 map."""

    def __init__(self, h, conv):
        self.h, self.conv = h, conv
        self.is_cuda, self.device = h.is_cuda, h.device

    def at(self, cells):
        rows = self.h.permute(0, 2, 3, 1)
        out = _SparseConv3x3.apply(rows if rows.is_contiguous() else rows.contiguous(), self.conv.weight, self.conv.bias, cells)
        return out.view(tuple(cells.shape) + (out.shape[-1],))

    def dense(self):
        return nchw_as_rows(exit_mixed(conv3x3(self.h, self.conv)))


def sparse_conv_available(h, conv):
    return (h.is_cuda and h.dtype == torch.float32 and h.dim() == 4 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32 and h.shape[1] % 4 == 0
            and os.environ.get('PCACC_SPARSE_EGO', '1') != '0')


class _Sinkhorn(torch.autograd.Function):
    """Log-domain Sinkhorn with slack (models/egomotion.py:100-137) on [P,k,k] log-affinities: two launches per iteration forward,
    two backward (csrc/ego.hip); the backward replays the half-steps from the input and the recorded log-sum-exp vectors."""

    @staticmethod
    def forward(ctx, log_alpha, n_iters):
        log_alpha = log_alpha.contiguous().float()
        out, lr, lc = native.sinkhorn_forward(log_alpha, n_iters)
        ctx.save_for_backward(log_alpha, lr, lc)
        return out

    @staticmethod
    def backward(ctx, grad):
        log_alpha, lr, lc = ctx.saved_tensors
        return native.sinkhorn_backward(grad.contiguous().float(), log_alpha, lr, lc), None


class _EgoAffinity(torch.autograd.Function):
    """affinity = -(square_distance(fs, ft, normalised) - softplus(alpha)) / (exp(beta) + 0.02)  (models/egomotion.py:177-180) for P pairs:
    one tiled kernel forward; backward = one pass giving d(dot) and the two scalar gradients + two library GEMMs for the features."""

    @staticmethod
    def forward(ctx, feats_s, feats_t, params):
        fs, ft = feats_s.contiguous().float(), feats_t.contiguous().float()
        aff = native.ego_affinity_forward(fs, ft, params)
        ctx.save_for_backward(fs, ft, aff, params)
        return aff

    @staticmethod
    def backward(ctx, g):
        fs, ft, aff, params = ctx.saved_tensors
        gd, gp = native.ego_affinity_backward(g.contiguous().float(), aff, params)
        return torch.bmm(gd, ft), torch.bmm(gd.transpose(1, 2), fs), gp


def ego_affinity(feats_s, feats_t, softplus_alpha, denom):
    """feats [P,k,c] (rows L2-normalised by the caller), the two scalars as tensors -> [P,k,k]."""
    params = torch.stack([softplus_alpha.reshape(()), denom.reshape(())]).float()
    return _EgoAffinity.apply(feats_s, feats_t, params)


class _EgoPerm(torch.autograd.Function):
    """(perm, rowsum, weighted_t, colsum) of models/egomotion.py:173-184 + libs/outlier_loss.py from the Sinkhorn result: one pass each way
    plus a column-sum pass (the support mask is rebuilt from the coordinates inside the kernel; coordinates and thresholds carry no
    gradient).  With the two sum vectors the outlier loss no longer touches the [P,k,k] matrix: its gradient -- a constant per row and
    column -- reaches the backward kernel as two [P,k] vectors instead of a dense matrix built by expand + add."""

    @staticmethod
    def forward(ctx, log_perm, coor_s, coor_t, thr2):
        ct = coor_t.contiguous().float()
        perm, rowsum, wt, colsum = native.ego_perm_forward(log_perm.contiguous().float(), coor_s.contiguous().float(), ct, thr2.contiguous().float())
        ctx.save_for_backward(perm, ct, rowsum, wt)
        ctx.set_materialize_grads(False)
        return perm, rowsum.unsqueeze(2), wt, colsum

    @staticmethod
    def backward(ctx, g_perm, g_rowsum, g_wt, g_colsum):
        perm, ct, rowsum, wt = ctx.saved_tensors
        c = lambda t: t.contiguous().float() if t is not None else None
        g = native.ego_perm_backward(c(g_perm), c(g_rowsum.squeeze(2)) if g_rowsum is not None else None, c(g_wt), c(g_colsum), perm, ct, rowsum, wt)
        return g, None, None, None


def ego_perm(log_perm, coor_s, coor_t, thr2):
    """-> (perm [P,k,k], rowsum [P,k,1], weighted_t [P,k,3], colsum [P,k])."""
    return _EgoPerm.apply(log_perm, coor_s, coor_t, thr2)


class _KabschCov(torch.autograd.Function):
    """(cov, x1_mean, x2_mean) of the weighted Kabsch solve (toolbox/register_utils.py:263-291) for P pairs: one kernel each way instead of
    ~15 + ~35 small launches on [P,k,3] tensors.  x1 (pillar means) carries no gradient."""

    @staticmethod
    def forward(ctx, x1, x2, w):
        x1, x2, w = x1.contiguous().float(), x2.contiguous().float(), w.contiguous().float()
        cov, m1, m2, norm = native.kabsch_cov_forward(x1, x2, w)
        ctx.save_for_backward(x1, x2, w, m1, m2, norm)
        ctx.set_materialize_grads(False)
        return cov, m1.unsqueeze(1), m2.unsqueeze(1)

    @staticmethod
    def backward(ctx, g_cov, g_m1, g_m2):
        x1, x2, w, m1, m2, norm = ctx.saved_tensors
        c = lambda t: t.contiguous().float() if t is not None else None
        gx2, gw = native.kabsch_cov_backward(x1, x2, w, m1, m2, norm, c(g_cov), c(g_m1.squeeze(1)) if g_m1 is not None else None,
                                             c(g_m2.squeeze(1)) if g_m2 is not None else None)
        return None, gx2, gw


def kabsch_cov(x1, x2, w):
    """-> (cov [P,3,3], x1_mean [P,1,3], x2_mean [P,1,3])."""
    return _KabschCov.apply(x1, x2, w)


class _KabschRT(torch.autograd.Function):
    """rotation = V diag(1, 1, det(V U^T)) U^T and translation = x2_mean - R x1_mean (toolbox/register_utils.py:305-313) from the SVD factors
    and the weighted means: one thread per pair each way (the torch formulation's det() is an LU factorisation, forward and backward)."""

    @staticmethod
    def forward(ctx, u, v, m1, m2):
        u, v = u.contiguous().float(), v.contiguous().float()
        m1c, m2c = m1.reshape(-1, 3).contiguous().float(), m2.reshape(-1, 3).contiguous().float()
        rot, trans = native.kabsch_rt_forward(u, v, m1c, m2c)
        ctx.save_for_backward(u, v, m1c)
        ctx.shapes = (m1.shape, m2.shape)
        ctx.set_materialize_grads(False)
        return rot, trans.unsqueeze(2)

    @staticmethod
    def backward(ctx, g_rot, g_trans):
        u, v, m1c = ctx.saved_tensors
        c = lambda t: t.contiguous().float() if t is not None else None
        gu, gv, gm1, gm2 = native.kabsch_rt_backward(u, v, m1c, c(g_rot), c(g_trans.squeeze(2)) if g_trans is not None else None)
        return gu, gv, gm1.view(ctx.shapes[0]), gm2.view(ctx.shapes[1])


def kabsch_rt(u, v, x1_mean, x2_mean):
    """-> (rotation [n,3,3], translation [n,3,1])."""
    return _KabschRT.apply(u, v, x1_mean, x2_mean)


def sinkhorn(log_alpha, n_iters):
    return _Sinkhorn.apply(log_alpha, int(n_iters))


# ---------------------------------------------------------------------------------------------------
class _SegLoss(torch.autograd.Function):
    """L1 (csrc/loss.hip): weighted cross entropy + Lovasz-Softmax + IoU counters of libs/loss.py:110-137 on the selected rows of
    a two-class logit tensor, read in place ([n,2] rows or an NCHW head output).  Returns (terms [2] = cross entropy, Lovasz;
    metric [4,2] f64 = compute_iou's counters)."""

    @staticmethod
    def forward(ctx, logits, labels, rows, plane):
        n = rows.shape[0] if rows is not None else logits.numel() // 2
        terms, metric, lov, saved = native.seg_loss_forward(logits, plane, labels, rows, n)
        ctx.save_for_backward(logits, labels, rows, lov, saved)
        ctx.plane, ctx.n = plane, n
        ctx.mark_non_differentiable(metric)
        return terms, metric

    @staticmethod
    def backward(ctx, grad_terms, _):
        logits, labels, rows, lov, saved = ctx.saved_tensors
        g = grad_terms.contiguous().float()
        return native.seg_loss_backward(logits, ctx.plane, labels, rows, ctx.n, lov, saved, g[0:1], g[1:2]), None, None, None


def seg_loss(logits, labels, rows=None):
    """logits [n,2] or [..., 2, H, W] (f32 / bf16), labels with one entry per row (any integer type, flattened), rows = int64
    indices of the supervised rows or None for all.  -> (terms [2], metric [4,2])."""
    if logits.dim() >= 3 and not logits.is_contiguous() and logits.movedim(-3, -1).is_contiguous():
        logits = logits.movedim(-3, -1).reshape(-1, 2)                           # channels-last head output: already rows
    if logits.dim() >= 3:
        plane = logits.shape[-1] * logits.shape[-2]
        assert logits.shape[-3] == 2
    else:
        plane = 0
        assert logits.dim() == 2 and logits.shape[1] == 2
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    labels = labels.reshape(-1)
    if labels.dtype != torch.int64:
        labels = labels.long()
    if rows is not None and rows.dtype != torch.int64:
        rows = rows.long()
    return _SegLoss.apply(logits.contiguous(), labels.contiguous(), rows, plane)


class _OffsetLoss(torch.autograd.Function):
    """L2 (csrc/loss.hip): libs/loss.py:194-250 in three launches; gradient w.r.t. the estimated offsets only (the ground truth and
    the ego-compensated points carry none, models/motionnet.py:207-209)."""

    @staticmethod
    def forward(ctx, offset_est, points, time_indice, inst_labels, label_base, ego_motion, inst_motion, transformed_points, rows):
        out, gt = native.offset_loss_forward(points, time_indice, inst_labels, label_base, ego_motion, inst_motion, ego_motion.shape[1],
                                             transformed_points, offset_est, rows)
        ctx.save_for_backward(gt, offset_est, rows)
        ctx.mark_non_differentiable(gt)
        return out, gt

    @staticmethod
    def backward(ctx, grad_out, _):
        gt, est, rows = ctx.saved_tensors
        g = grad_out.contiguous().float()
        return (native.offset_loss_backward(gt, est, rows, g[0:1], g[1:2]),) + (None,) * 8


def offset_loss(offset_est, points, time_indice, inst_labels, label_base, ego_motion, inst_motion, transformed_points, rows):
    """-> (out [3] = L1 term, direction term, mean L2 error; offset_gt [m,2])."""
    f = lambda t: t.detach().float().contiguous()
    return _OffsetLoss.apply(offset_est.float().contiguous(), f(points), time_indice.long().contiguous(), inst_labels.long().contiguous(),
                             label_base, f(ego_motion), f(inst_motion), f(transformed_points), rows)


class _FramesMax(torch.autograd.Function):
    """A9: max over the frame axis of [S,T,...] rows in one streaming pass each way (csrc/canvas.hip)."""

    @staticmethod
    def forward(ctx, x):
        mixed = _MIXED and x.dtype == torch.bfloat16
        x32 = twin(x) if mixed else x
        out, arg = native.frames_max(x32)
        ctx.save_for_backward(arg)
        ctx.frames = x.shape[1]
        return shadow(carry_amax(x32, out)) if mixed else out

    @staticmethod
    def backward(ctx, grad):
        arg, = ctx.saved_tensors
        return native.frames_max_backward(grad.contiguous(), arg, ctx.frames)


def frames_max(x):
    """torch.max(x, dim=1)[0] for a contiguous f32 / bf16 [S,T,...] stack (models/stpn.py:83)."""
    return _FramesMax.apply(x.contiguous())


class _Svd3(torch.autograd.Function):
    """torch.svd for [n,3,3] matrices without the host-side status check (csrc/ego.hip: Jacobi in float64, closed-form backward)."""

    @staticmethod
    def forward(ctx, a):
        u, s, v = native.svd3(a.contiguous().float())
        ctx.save_for_backward(u, s, v)
        return u, s, v

    @staticmethod
    def backward(ctx, gu, gs, gv):
        u, s, v = ctx.saved_tensors
        c = lambda g: g.contiguous().float() if g is not None else None
        return native.svd3_backward(u, s, v, c(gu), c(gs), c(gv))


def svd3(a):
    """(u, s, v) with a = u diag(s) v^T, as torch.svd (toolbox/register_utils.py:293)."""
    return _Svd3.apply(a)


class _BatchNormRows(torch.autograd.Function):
    """Training-mode nn.BatchNorm1d on [rows, c] rows (models/unet.py:240-245) in four streaming passes (csrc/bn.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, momentum, running_mean, running_var, relu=False):
        x = x.contiguous()
        mixed = _MIXED and x.dtype == torch.bfloat16               # shadow rows: statistics and output from the fp32 twin, bf16 backward on the shadow
        if mixed and twin(x).dim() == 2 and twin(x).is_contiguous():
            y, y16, am, mean, invstd = native.bn_rows_forward_dual(twin(x), gamma, beta, eps, momentum, running_mean, running_var, relu=relu)
            set_amax_tag(y, am)
            out = shadow(y, y16)                                    # shadow and maxima from the normalisation's own store phase
        else:
            y, mean, invstd = native.bn_rows_forward(twin(x) if mixed else x, gamma, beta, eps, momentum, running_mean, running_var, relu=relu)
            out = shadow(y) if mixed else y
        ctx.save_for_backward(x, gamma, mean, invstd, beta if relu else None)
        ctx.relu = relu
        return out

    @staticmethod
    def backward(ctx, gy):
        x, gamma, mean, invstd, beta = ctx.saved_tensors
        if _SPLIT and x.dtype == torch.float32:                    # an fp32x3 layer sits in front: the maxima its backward scales by come from this store phase
            gx, gg, gb, am = native.bn_rows_backward(gy.contiguous().to(x.dtype), x, gamma, mean, invstd, relu_beta=beta, relu=ctx.relu, want_amax=True)
            set_amax_tag(gx, am)
        else:
            gx, gg, gb = native.bn_rows_backward(gy.contiguous().to(x.dtype), x, gamma, mean, invstd, relu_beta=beta, relu=ctx.relu)
        return gx, (gg if gamma is not None else None), (gb if gamma is not None else None), None, None, None, None, None


def batch_norm_rows(x, bn):
    """`bn(x)` for an nn.BatchNorm1d `bn` on 2-D rows.  Training mode on the GPU with a momentum and f32 affine parameters runs
    the fused passes (running statistics and num_batches_tracked updated as the module does); everything else is the module."""
    if not (bn.training and x.is_cuda and x.dim() == 2 and bn.momentum is not None and native.bn_rows_supported(x)
            and x.shape[0] >= MIN_ROWS_FUSED_LINEAR and not torch.is_autocast_enabled()
            and (bn.weight is None or bn.weight.dtype == torch.float32)):
        return bn(x)
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return _BatchNormRows.apply(x, bn.weight, bn.bias, float(bn.eps), float(bn.momentum), rm, rv)


def batch_norm_nchw(x, bn, relu=False):
    """`bn(x)` (relu: `relu(bn(x))`, one pass each way) for an nn.BatchNorm2d on a channels-last NCHW tensor (models/unet.py:259-277, the two SegHead2D): training mode on the GPU
    runs the streaming passes of csrc/bn.hip on the [N*H*W, C] rows the memory already is (f32 or bf16 rows, statistics in fp32 /
    float64 as the module's); everything else is the module (library)."""
    def module(t):
        return torch.relu(bn(t)) if relu else bn(t)

    def replay(t):                                                              # the same function without the running-statistics update
        y = torch.nn.functional.batch_norm(t, None if bn.training else bn.running_mean, None if bn.training else bn.running_var, bn.weight, bn.bias,
                                           bn.training, 0.0, bn.eps)
        return torch.relu(y) if relu else y

    def fallback():
        if _MIXED and x.dtype == torch.bfloat16:                                # a shadow: the module on its fp32 twin
            return on_twin(x, module, [p for p in (bn.weight, bn.bias) if p is not None], replay)
        return module(x)
    if not (bn.training and x.is_cuda and x.dim() == 4 and bn.momentum is not None and x.dtype in (torch.float32, torch.bfloat16)
            and (bn.weight is None or bn.weight.dtype == torch.float32)):
        return fallback()
    rows = x.permute(0, 2, 3, 1)
    if not rows.is_contiguous():
        return fallback()
    n, h, w, c = rows.shape
    rows = rows.reshape(n * h * w, c)
    if not native.bn_rows_supported(rows) or rows.shape[0] < MIN_ROWS_FUSED_LINEAR:
        return fallback()
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    y = _BatchNormRows.apply(rows, bn.weight, bn.bias, float(bn.eps), float(bn.momentum), rm, rv, relu)
    return y.view(n, h, w, c).permute(0, 3, 1, 2)
