"""MotionNet: host mirror of models/motionnet.py -- the drop-in boundary of the hot path.

Same constructor (`MotionNet(cfg)`), same `forward(input_dict) -> results` keys and dtypes, same sub-module
attribute names and therefore the same 195 state_dict keys, so a released checkpoint
loads through the reference's key-filtered partial_load (toolbox/utils.py:16-24).

What differs is where the work runs: voxel bookkeeping, per-pillar reductions, pillar scatter, bilinear
gathers, BEV warp and the per-point transform are single launches of the gfx950 kernels behind include/pcacc.h
(all batch elements and frames at once -- the reference loops over batch and frame in Python,
models/pillar_encoder.py:144,193, models/motionnet.py:97-135), on channels-last canvases.

cfg['misc']['compute_dtype'] ('fp32' default | 'bf16' | 'fp32x3'): element type of the BEV canvas and of the conv stacks
(autocast for bf16; 'fp32x3' = fp32 tensors with split-bf16 MFMA products, the fast mode that matches the reference to 1e-3).  Sinkhorn / Kabsch / normalisation / grid arithmetic always stay fp32.
"""
import contextlib
import os
import weakref

import torch
import torch.nn as nn

from . import native, ops
from .alignnet import AlignNet
from .cluster import Cluster
from .egomotion import EgoMotionHead
from .lazy import HostCopy, LazyDict, lazy_scalars
from .ops import PillarIndex
from .pillar_encoder import PillarFeatureNet, temporal_ungrid
from .stpn import STPN
from .unet import SegHead2D, UNet

MIN_POINTS = 15                          # models/motionnet.py:11


def share_with_stream(stream, *objs):
    """Tensor.record_stream(stream) for every CUDA tensor in `objs` (dicts, lists, tuples and PreparedInputs are walked)."""
    for o in objs:
        if torch.is_tensor(o):
            if o.is_cuda:
                o.record_stream(stream)
        elif isinstance(o, dict):
            share_with_stream(stream, *[dict.__getitem__(o, k) for k in o])
        elif isinstance(o, (list, tuple)):
            share_with_stream(stream, *o)
        elif isinstance(o, PreparedInputs):
            share_with_stream(stream, *o.tensors())


class PreparedInputs(object):
    """What MotionNet.forward derives from the batch alone (no weights involved): the pillar index with its CSR, the per-point
    frame / sample indices, per-pillar means and labels, the occupancy and label maps and the pillar encoder's 9 point features.
    A data pipeline can build it while the previous step is still on the GPU (MotionNet.prepare_inputs; bench.py does, on the
    stream that voxelises the next batch) and hand it over as input_dict['_prepared']."""

    FIELDS = ('pidx', 'batch_idx', 'frame_idx', 'pillar_mean', 'fb_labels_sub', 'occ_map', 'fb_seg_gt', 'features')

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, kw[k])

    def tensors(self):
        out = [getattr(self, k) for k in self.FIELDS[1:]]
        out += [v for v in vars(self.pidx).values() if torch.is_tensor(v)]
        out += [t for v in vars(self.pidx).values() if isinstance(v, (tuple, list)) for t in v if torch.is_tensor(t)]
        return [t for t in out if torch.is_tensor(t)]

    def matches(self, input_dict):
        """True when this object was built from THIS batch: the batch's `input_points` is the very tensor object prepare_inputs saw (a weak
        reference; an address + version pair can recur -- the caching allocator recycles addresses and a fresh tensor starts at version 0) and has
        not been written since.  A batch whose points were copied or moved after the preparation is prepared again inside the forward."""
        src = getattr(self, 'source', None)
        pts = input_dict['input_points']
        return (self.pidx.m == input_dict['coordinates'].shape[0] and self.pidx.n == pts.shape[0] and self.features.device == pts.device
                and (src is None or (src[0]() is pts and src[1] == pts._version)))


def grid_shape(cfg):
    """[nx, ny, nz, nt] as Voxelization computes it (libs/voxel_generator.py:123-125, fp32 round)."""
    vg = cfg['voxel_generator']
    r = torch.tensor(vg['range'], dtype=torch.float32)
    vs = torch.tensor(vg['voxel_size'], dtype=torch.float32)
    g = torch.round((r[3:] - r[:3]) / vs).to(torch.int64).tolist()
    return g + [int(vg['n_sweeps'])]


class MotionNet(nn.Module):
    def __init__(self, cfg):
        super(MotionNet, self).__init__()
        unet_cfg = cfg['unet']
        self.pillar_encoder = PillarFeatureNet(cfg['pillar_encoder'])
        self.unet = UNet(**unet_cfg)
        self.semseg_head = SegHead2D(unet_cfg['in_channels'], 2)
        self.ego_feats_head = SegHead2D(unet_cfg['in_channels'], cfg['pose_estimation']['feats_dim'])
        self.ego_motion_head = EgoMotionHead(cfg)
        self.resolution = cfg['voxel_generator']['voxel_size']
        self.pc_range = cfg['voxel_generator']['range']
        self.motionhead = STPN(cfg['stpn']['feat_dim'])
        self.cluster = Cluster(cfg)
        self.mode = cfg['misc']['mode']
        self.reconstructor = AlignNet(cfg)
        self.grid = grid_shape(cfg)
        # 'fp32x3': fp32 tensors, products of the dense stacks / wide row layers on the bf16 matrix cores from hi / lo halves (ops.set_split)
        self.compute_mode = cfg['misc'].get('compute_dtype', 'fp32')
        # 'mixed2' [r6]: 'mixed' with the STPN's per-point layers (positional code, final projection, the two SegHead1D) on bf16 rows -- the one stage the
        # precision map (profiles/r06_precision_map.txt) shows to hold north_star's 1e-3 with one-term bf16 products
        self.compute_dtype = {'fp32': torch.float32, 'fp32x3': torch.float32, 'mixed': torch.float32, 'mixed2': torch.float32, 'bf16': torch.bfloat16}[self.compute_mode]
        # pillars renumbered in canvas-cell order inside forward() (ops.PillarIndex); False keeps the voxeliser's numbering
        self.cell_ordered_pillars = bool(cfg['misc'].get('cell_ordered_pillars', True))
        # [r6] the pillar encoder on rows stored pillar by pillar (PillarIndex.pillar_major).  Bit-identical forward (tests/test_mixed.py); measured in the step on
        # uniform synthetic points (profiles/r06_pillar_major_ab.txt): 29.68 / 29.51 / 29.85 ms against 29.59 / 29.70 / 29.69 ms -- the poolings do not get faster
        # (223 vs 223 us, 165 vs 165 us: a frame's 20 MB of rows already sit in the Infinity Cache when they are gathered), the blocks gain 11 - 14 us each and
        # the feature build pays 51 us for its random point reads.  Off by default; PCACC_PILLAR_MAJOR=1 or misc.pillar_major_rows turn it on.
        self.pillar_major_rows = bool(cfg['misc'].get('pillar_major_rows', os.environ.get('PCACC_PILLAR_MAJOR', '0') != '0'))
        self._optimizer_watched = False          # watch_optimizer()
        self.after_ego = None                    # optional callable(results), see forward()
        self.before_sync = None                  # optional callable(): host work to do while the forward waits for its one host sync, see forward()
        self.side_stream = None                  # optional torch.cuda.Stream for stages 5-6 while after_ego's backward runs

    # ------------------------------------------------------------------------------------------------
    def early_parameters(self):
        """The parameters whose whole gradient comes from the loss terms of FuseLoss.early_terms (see `after_ego` in forward())."""
        mods = (self.pillar_encoder, self.unet, self.semseg_head, self.ego_feats_head, self.ego_motion_head)
        return [p for m in mods for p in m.parameters()]

    def train(self, mode=True):
        """nn.Module.train / eval; a mode switch also drops the prepared copies of the convolution weights (ops.weights_may_have_changed:
        a fused optimizer step between the last training forward and this switch leaves no trace in the parameters' version counters)."""
        ops.weights_may_have_changed()
        return super().train(mode)

    def watch_optimizer(self, optimizer=None):
        """Invalidate the prepared convolution weights at the writer: EVERY optimizer's step() in this process calls ops.weights_may_have_changed()
        (ops.watch_all_optimizers: torch's process-wide post-step hook, registered once) -- the optimizer DataParallelStep was built with, one
        re-created on resume or at a learning-rate phase change, a second one for another parameter group alike.  forward() then stops
        invalidating on its own, so the micro-steps of an accumulation window (iter_size > 1) reuse one set of copies.  `optimizer` is accepted for
        the call sites of earlier rounds and returned unchanged.  A writer that is neither an optimizer step nor visible to the parameters'
        version counters (raw pointers) still needs ops.weights_may_have_changed()."""
        ops.watch_all_optimizers()
        self._optimizer_watched = True
        return optimizer

    def channels_last_(self):
        """Store conv weights in the layout the channels-last activations want (no state_dict change)."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            elif isinstance(m, nn.Conv3d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last_3d)
        return self

    def _dense(self):
        if self.compute_dtype == torch.bfloat16:
            return torch.autocast(device_type='cuda', dtype=torch.bfloat16)
        return contextlib.nullcontext()

    # reference helpers kept for callers that use them directly ------------------------------------------
    def warp_feats(self, bev_feats, pose_estimation):
        """models/motionnet.py:82-114.  bev_feats [B,T,C,Ny,Nx], pose [B,T,4,4] -> [B,T,C,Ny,Nx]."""
        bev_cl = bev_feats.permute(0, 1, 3, 4, 2).contiguous()
        inv = torch.linalg.inv(pose_estimation.float())
        out = ops.bev_warp(bev_cl, inv, self.resolution[0], self.resolution[1], self.pc_range[0], self.pc_range[1])
        return out.permute(0, 1, 4, 2, 3)

    def transform_points(self, points, time_indice, transformation):
        """models/motionnet.py:117-135."""
        T = transformation.size(1)
        frame_idx = (time_indice[:, 0] * T + time_indice[:, 1]).to(torch.int32)
        return ops.rigid_transform(points, frame_idx, transformation)

    # ------------------------------------------------------------------------------------------------
    def prepare_inputs(self, input_dict):
        """The weight-independent head of forward(): see PreparedInputs.  (models/motionnet.py:150-166 and the feature build of
        models/pillar_encoder.py:98-110.)"""
        input_points = input_dict['input_points'].float()
        time_indice = input_dict['time_indice']
        fb_labels = input_dict['fb_labels']
        coordinates = input_dict['coordinates']
        B = input_dict['num_voxels'].size(0)
        nx, ny, nz, nt = self.grid
        device = coordinates.device
        pidx = PillarIndex(coordinates, input_dict['point_to_voxel_map'], B, self.grid, cell_order=self.cell_ordered_pillars)
        assert pidx.n == input_points.size(0)
        batch_idx = time_indice[:, 0].to(torch.int32).contiguous()
        frame_idx = (time_indice[:, 0] * nt + time_indice[:, 1]).to(torch.int32).contiguous()
        pillar_mean, fb_labels_sub = ops.segment_mean3_maxlabel(input_points, fb_labels, pidx)   # motionnet.py:159-160
        occ = ops.pillar_scatter(torch.ones((pidx.m, 1), device=device), pidx)
        fb_map = ops.pillar_scatter(fb_labels_sub.float().unsqueeze(1), pidx)
        # [r6] the encoder's rows in pillar order on the GPU (PillarIndex.pillar_major): nothing per point leaves the encoder, so the order is its own business
        features = self.pillar_encoder.point_features(input_points, pidx, pidx.coordinates, pillar_mean, time_indice,
                                                      pillar_major=self.pillar_major_rows and device.type == 'cuda')
        prep = PreparedInputs(pidx=pidx, batch_idx=batch_idx, frame_idx=frame_idx, pillar_mean=pillar_mean, fb_labels_sub=fb_labels_sub,
                              occ_map=occ.view(B, nt, 1, ny, nx), fb_seg_gt=fb_map.view(B, nt, 1, ny, nx).to(fb_labels.dtype), features=features)
        prep.source = (weakref.ref(input_dict['input_points']), input_dict['input_points']._version)      # a `_prepared` of another batch is refused
        return prep

    def forward(self, input_dict):
        input_points = input_dict['input_points'].float()                    # [N,3]
        time_indice = input_dict['time_indice']                                # [N,2] f64 (b,t)
        fb_labels = input_dict['fb_labels']                                    # [N,1]
        ego_motion_gt = input_dict['ego_motion_gt'].float()                    # [B,T,4,4]
        coordinates = input_dict['coordinates']                                # [M,5] f64 (b,z,y,x,t)
        num_voxels = input_dict['num_voxels']
        batch_size = num_voxels.size(0)
        nx, ny, nz, nt = self.grid                                             # == input_dict['shape'][0], no host sync
        self.Nx, self.Ny, self.nt = nx, ny, nt
        B, T, Ny, Nx = batch_size, nt, ny, nx
        device = coordinates.device
        ops.set_point_dtype(self.compute_dtype if device.type == 'cuda' else torch.float32)
        ops.set_split(self.compute_mode in ('fp32x3', 'mixed', 'mixed2') and device.type == 'cuda')
        # 'mixed': fp32x3 forward values, bf16 gradient graph inside the two convolution segments (ops.set_mixed)
        ops.set_mixed(self.compute_mode in ('mixed', 'mixed2') and device.type == 'cuda')
        ops.twins_clear()
        # Prepared (packed / split) copies of the convolution weights outlive writers that bypass the parameters' version counters (fused
        # optimizers).  With the optimizer watched (watch_optimizer: its step invalidates the copies) nothing is needed here and the micro-steps of a
        # gradient-accumulation window share one preparation; otherwise every forward that can be followed by an optimizer step -- gradients
        # enabled and a trainable parameter, in train() OR eval() mode (fine-tuning with frozen BatchNorm statistics) -- starts from fresh copies.
        if not self._optimizer_watched and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            ops.weights_may_have_changed()
        results = LazyDict()

        # 0. index structures shared by every irregular op of this forward, and everything else that follows from the batch alone
        prep = input_dict.get('_prepared')
        if prep is None or not prep.matches(input_dict):
            prep = self.prepare_inputs(input_dict)
        pidx, batch_idx, frame_idx = prep.pidx, prep.batch_idx, prep.frame_idx
        coordinates = pidx.coordinates                                         # rows in the numbering pidx uses from here on
        pillar_mean = prep.pillar_mean
        results['fb_seg_gt'], results['occ_map'] = prep.fb_seg_gt, prep.occ_map

        # 1. pillar encoder -> BEV canvas (channels-last, one streaming pass)
        with ops.stage('pillar_encoder'):
            enc_pidx = pidx.pillar_major() if self.pillar_major_rows and device.type == 'cuda' else pidx
            input_features, is_canvas = self.pillar_encoder(input_points, None, coordinates, pillar_mean, time_indice, pidx=enc_pidx, keep_dtype=True,
                                                            features=prep.features, canvas=True)
        # 'mixed' mode: the encoder's last pooling has written the canvas itself (ops.segment_max_canvas); otherwise rows -> canvas here
        canvas = input_features if is_canvas else ops.carry_amax(input_features, ops.pillar_scatter(input_features, pidx, self.compute_dtype))     # rows or zeros
        bev = ops.carry_amax(canvas, ops.canvas_as_nchw(canvas, pidx))         # [B*T, C, Ny, Nx]
        bev = ops.enter_mixed(bev)                                             # mixed mode: segment 1 = U-Net + the two heads on bf16 shadows

        # 2. backbone + 3. fg/bg head
        with self._dense():
            bev_feats = self.unet(bev)
            with ops.stage('fb_head'):
                fb_seg = self.semseg_head(bev_feats)
        fb_seg = fb_seg.float()
        results['fb_seg_est'] = fb_seg.view(B, T, 2, Ny, Nx)
        fb_est = (fb_seg[:, 1] > fb_seg[:, 0]).long()                          # argmax, ties -> 0 (motionnet.py:190)
        fb_est_pillar = ops.gather_rows(fb_est.reshape(-1, 1), pidx.cell)      # inverse scatter  [M,1]
        fb_est_per_point = ops.gather_rows(fb_est_pillar, pidx.p2v)            # [N,1]
        results['fb_est_per_points'] = fb_est_per_point

        # ---- the ONE host sync of the forward: every size the rest of the pass needs --------------------------------
        # (the reference syncs at every boolean-mask index and .item(): ~100 times per step; each sync drains the
        #  launch queue, so host and GPU stop overlapping)
        if self.mode in ['train', 'val']:
            fb_mask = torch.logical_or(fb_labels[:, 0] == 1, fb_est_per_point[:, 0] == 1)
            rec_mask = fb_labels[:, 0] == 1
        else:
            fb_mask = fb_est_per_point[:, 0] == 1
            rec_mask = None
        sorted_pillars, frame_offsets_dev = pidx.frame_pillars()
        bg_flag_sorted = ops.gather_rows(fb_est_pillar, sorted_pillars)[:, 0] == 0           # cell order, per frame
        bg_cum = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(bg_flag_sorted, 0)])
        pad_flags = None
        if rec_mask is not None:
            # TubeNet's instance bookkeeping (alignnet.py:121-140) depends on the input labels only: its flags ride along with the
            # sizes instead of costing a second queue drain in front of the ~700 small launches of the TubeNet
            counts = [m.size(0) for m in input_dict['inst_motion_gt']]
            base = native.upload_small([sum(counts[:b]) for b in range(len(counts))], torch.int64, device)
            merged = input_dict['inst_labels'][:, 0].long() + base[time_indice[:, 0].long()]
            pad_flags = AlignNet.padding_flags(merged, time_indice[:, 1].long(), sum(counts), T, weights=rec_mask)[0]
        sizes = HostCopy(torch.cat([frame_offsets_dev.long(), bg_cum[frame_offsets_dev.long()], fb_mask.sum()[None],
                                    (rec_mask.sum() if rec_mask is not None else fb_mask.sum())[None]]
                                   + ([pad_flags.reshape(-1).long()] if pad_flags is not None else [])))
        # The ego feature head (two full-resolution convolutions, needed by the ego head only) is queued between the request for the
        # sizes and the wait for them: the GPU still has work when the host wakes up and starts issuing the ego head's small launches.
        with self._dense(), ops.stage('ego_head'):
            # With the device key-point sampler the ego head reads <= 1024 rows per frame of this head's output: its last convolution is then
            # evaluated at those cells only (ops.SparseConvRows); the dense map otherwise (reference sampler: whole frames are indexed)
            if self.ego_motion_head.kpt_sampler == 'device' and bev_feats.is_cuda and getattr(self.ego_motion_head, 'flat_keypoints', True):
                geo_rows = self.ego_feats_head.sparse_rows(bev_feats)
            else:
                geo_rows = ops.nchw_as_rows(ops.exit_mixed(self.ego_feats_head(bev_feats)))      # the ego head reads fp32 (mixed mode: the twin)
        if self.before_sync is not None:
            # the host is about to wait for the GPU to reach the sizes (the whole lower half of the forward is still queued): anything the caller wants
            # issued that does not depend on this forward -- the next batch's voxelisation on its own stream -- costs no host time here, and its
            # kernels run into the host-bound stretch that follows the wait (the ego head's ~100 small launches) instead of into the backward
            self.before_sync()
        sizes = sizes.numpy().tolist()
        nf = B * T + 1
        frame_offsets, bg_at = sizes[:nf], sizes[nf:2 * nf]
        if self.cell_ordered_pillars and frame_offsets[-1] != pidx.m:
            raise ValueError('coordinates: %d pillars but %d occupied cells (duplicate or out-of-range rows); the cell-ordered pillar '
                             'numbering needs one cell per pillar -- set misc.cell_ordered_pillars = False for such input'
                             % (pidx.m, frame_offsets[-1]))
        bg_counts = [bg_at[i + 1] - bg_at[i] for i in range(B * T)]
        n_fb, n_rec = int(sizes[2 * nf]), int(sizes[2 * nf + 1])
        if pad_flags is not None:
            pad_flags = torch.tensor(sizes[2 * nf + 2:], dtype=torch.bool).view(2, -1)
        # boolean-mask indexing as one index list per mask (wave ballot / prefix-sum compaction: native.compact_mask; the counts came with the host sync)
        bg_sorted_idx = native.compact_mask(bg_flag_sorted, bg_at[-1])
        fb_idx = native.compact_mask(fb_mask, n_fb)
        results['_fb_idx'], results['_cell'] = fb_idx, pidx.cell
        if self.mode in ['train', 'val']:
            # FuseLoss.get_mos_loss supervises GT-or-estimated foreground in EVERY mode (libs/loss.py:145-147): only in train / val is
            # that this forward's mask, so only then is the index list handed over (no re-sync); in test mode the loss rebuilds it
            results['_mos_idx'] = fb_idx

        # 4. ego motion (fp32).  The per-cell L2 normalisation of motionnet.py:199 (no epsilon, trap 7) is applied to the
        #    gathered key-point rows inside the head instead of to the whole map.
        self.ego_motion_head.forward_pillars(geo_rows, pillar_mean, pidx, ego_motion_gt, results,
                                             frame_offsets, bg_sorted_idx, bg_counts)
        # Everything below works on detached features and poses (motionnet.py:205-209): the graph of the pillar encoder, the
        # U-Net, the two heads and the ego head is complete here.  A training step may hook in (`after_ego`) to evaluate the loss terms
        # that live on it and back-propagate them now (FuseLoss.early_terms, distributed.DataParallelStep): the largest kernels of
        # the step are then queued in front of the ~800 small launches of the motion heads and the TubeNet.
        fork = None
        if self.after_ego is not None:
            if self.side_stream is not None and input_points.is_cuda:
                fork = torch.cuda.Event()
                fork.record()                                  # the side stream joins HERE: in front of the backward the hook queues
            self.after_ego(results)

        # Stages 5 and 6 read detached tensors only: with a side stream set (training steps that back-propagate the early terms in the
        # hook above) they run there, concurrently with that backward on the main stream -- their hundreds of small kernels fill the
        # gaps the large ones leave instead of queueing behind them.  The caller joins the streams (DataParallelStep).
        def upper():
            nonlocal bev_feats, rec_mask, n_rec
            # 5. motion segmentation on ego-motion-compensated features
            pose_est = results['ego_motion_est'].float().detach()
            bev_feats = ops.carry_amax(bev_feats, bev_feats.detach())            # (the scale bound travels with the detached alias)
            C = bev_feats.size(1)
            bev_cl = bev_feats.permute(0, 2, 3, 1).contiguous().view(B, T, Ny, Nx, C)
            # mixed mode: segment 2 = STPN temporal stack + U-Net starts here (the warp writes the segment's bf16 shadow itself)
            warped = ops.bev_warp_enter_mixed(bev_cl, native.inv4x4(pose_est), self.resolution[0], self.resolution[1], self.pc_range[0], self.pc_range[1],
                                              amax_from=ops.twin_or_self(bev_feats))
            warped_feats = ops.carry_amax(warped, warped.permute(0, 4, 1, 2, 3))   # [B,C,T,H,W], channels_last_3d memory
            transformed_points = ops.rigid_transform(input_points, frame_idx, pose_est)
            results['transformed_points'] = transformed_points

            full_mos = torch.zeros(transformed_points.size(0), 2, device=device)
            full_offset = torch.zeros(transformed_points.size(0), 2, device=device)
            full_mos[:, 0] = 1
            mos_feats = None
            if n_fb > MIN_POINTS:
                with self._dense():
                    stpn_map = self.motionhead.backbone(warped_feats)
                fb_idx32 = fb_idx.to(torch.int32)                       # one conversion for every row gather of this index list
                mos, offset, mos_feats = self._stpn_heads(stpn_map, self._take_rows(transformed_points, fb_idx32), self._take_rows(batch_idx, fb_idx32))
                full_mos = full_mos.index_copy(0, fb_idx, mos)
                full_offset = full_offset.index_copy(0, fb_idx, offset)
            results['mos_est'] = full_mos
            results['offset_est'] = full_offset
            results['rec_est'] = transformed_points.clone()

            # 6. TubeNet
            if self.mode in ['train', 'val']:
                inst_labels = input_dict['inst_labels'][:, 0].long()
            else:
                results['_n_batches'] = B                     # spares the cluster step a .max() read-back
                self.cluster(transformed_points, full_mos.argmax(1), full_offset, time_indice, results, use_offset=True)
                inst_labels = results['inst_labels_est']
                rec_mask = inst_labels != 0
                n_rec = int(rec_mask.sum())                   # the one extra host sync of test mode
            if n_rec > MIN_POINTS:
                rec_idx = native.compact_mask(rec_mask, n_rec)
                rec_idx32 = rec_idx.to(torch.int32)
                results['_rec_idx'] = rec_idx
                if self.mode in ['train', 'val']:
                    results['_gtfg_idx'] = rec_idx            # = nonzero(fb_labels == 1), what FuseLoss.get_offset_loss supervises (libs/loss.py:199)
                # mos_feats exists whenever rec_mask passes in train/val (fb_mask is a superset of rec_mask)
                backbone_feats = ops.bilinear_gather(bev_feats, self._take_rows(input_points, rec_idx32), self._take_rows(frame_idx, rec_idx32),
                                                     abs(self.pc_range[0]), abs(self.pc_range[1]))       # temporal_ungrid
                tp_rec = self._take_rows(transformed_points, rec_idx32)
                motion_feats = ops.bilinear_gather(mos_feats, tp_rec, self._take_rows(batch_idx, rec_idx32),
                                                   abs(self.pc_range[0]), abs(self.pc_range[1]))          # ungrid
                reconstructor_input = {
                    'inst_labels': self._take_rows(inst_labels, rec_idx32),
                    'time_indice': self._take_rows(time_indice, rec_idx32),
                    'transformed_points': tp_rec,
                    'backbone_feats': backbone_feats,
                    'motion_feats': motion_feats,
                    'inst_motion_gt': input_dict['inst_motion_gt'],
                    'mos_labels': self._take_rows(input_dict['sd_labels'], rec_idx32)[:, 0].long(),
                    'ego_motion_est': results['ego_motion_est'].detach(),     # alignnet.py:240 detaches what is derived from it
                    'ego_motion_gt': results['ego_motion_gt'],
                    '_pad_flags': pad_flags if self.mode in ['train', 'val'] else None,
                }
                with ops.stage('tubenet'):
                    self.reconstructor(reconstructor_input, results)
                results['rec_est'] = results['rec_est'].index_copy(0, rec_idx, results['sub_rec_est'])
            self._resolve_scalars(results)
            return results

        if fork is None:
            return upper()
        # what stages 5-6 read was allocated on the main stream: tell the caching allocator that the side stream uses it too, or a
        # block could be handed out again on the main stream (the early backward allocates there) while a side-stream kernel reads it
        share_with_stream(self.side_stream, input_dict, results, bev_feats, input_points, frame_idx, batch_idx, time_indice, fb_idx, rec_mask)
        self.side_stream.wait_event(fork)
        with torch.cuda.stream(self.side_stream):
            return upper()

    @staticmethod
    def _take_rows(table, idx):
        """table[idx] for the per-point tables of the batch (no gradient; any element type whose rows are a multiple of 4 bytes: [N,3] f32 points, [N,2] f64
        (sample, frame), [N] i32 / i64 labels and indices): one row-gather launch of the library instead of the generic advanced-indexing / index_select
        kernels (73 - 160 us per call at 3.2 M rows).  idx: int32 (or int64, converted here)."""
        if table.is_cuda and table.dim() in (1, 2) and not table.requires_grad and table.is_contiguous():
            rows = table if table.dim() == 2 else table.view(-1, 1)
            if (rows.shape[1] * rows.element_size()) % 4 == 0 and rows.shape[0] > 0:
                out = native.gather_rows(rows, idx if idx.dtype == torch.int32 else idx.to(torch.int32))
                return out if table.dim() == 2 else out.view(-1)
        return table[idx.long()]

    @staticmethod
    def _resolve_scalars(results):
        """Python floats for the scalar results the reference produces with .item() (egomotion.py:456, alignnet.py:280-281):
        one asynchronous device->host transfer for all of them, waited for when a value is first read (lazy.py)."""
        keys = [k for k in ('ego_rot_error', 'ego_trans_error', 'inst_l2_error', 'dynamic_inst_l2_error')
                if k in results and torch.is_tensor(dict.__getitem__(results, k))]
        if keys:
            for k, v in zip(keys, lazy_scalars([dict.__getitem__(results, k) for k in keys])):
                results[k] = v

    def _stpn_heads(self, stpn_map, points, batch_idx):
        """Per-point part of STPN.forward (models/stpn.py:91-104) on the already computed map."""
        mh = self.motionhead
        with ops.stage('point_heads'):
            ungridded = ops.bilinear_gather(stpn_map, points, batch_idx, abs(self.pc_range[0]), abs(self.pc_range[1]))
            if self.compute_mode == 'mixed2' and points.is_cuda and points.shape[0] >= ops.MIN_ROWS_FUSED_LINEAR:
                # the gather above read the fp32 map; from here to the logits: bf16 rows (ops.bf16_rows)
                with ops.bf16_rows():
                    pos = mh.point_mlp(mh.positional_encoding, points / abs(self.pc_range[0]))
                    enc = mh.point_mlp(mh.final_proj, ops.cat_rows(pos, ungridded))
                    classes = mh.point_head(mh.mos_seg, enc)
                    offset = mh.safe_guard_offset(mh.point_head(mh.offset_head, enc))
                return classes, offset, stpn_map
            pos = mh.point_mlp(mh.positional_encoding, points / abs(self.pc_range[0]))          # rows in ops.point_dtype()
            enc = mh.point_mlp(mh.final_proj, ops.cat_rows(pos, ungridded))
            classes = mh.point_head(mh.mos_seg, enc)
            offset = mh.safe_guard_offset(mh.point_head(mh.offset_head, enc))
        return classes, offset, stpn_map
