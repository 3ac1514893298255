"""Ego-motion head: host mirror of models/egomotion.py plus the registration helpers it uses from
toolbox/register_utils.py (kabsch_transformation_estimation, get_relative_pose_torch, rotation_error,
translation_error) and toolbox/utils.py (square_distance, _EPS).

Numerics stay fp32 throughout (Sinkhorn logsumexp, inverse, SVD must not be bf16).
Key points are drawn with torch.randperm on the HOST generator exactly where the reference draws them
(models/egomotion.py:157,163: source first, then target, per pair, per batch element), so that the same
torch.manual_seed gives the same key points.
"""
import math

import os

import torch
import torch.nn as nn

from . import native, ops

_EPS = 1e-20                       # toolbox/utils.py:13


def square_distance(src, dst, normalised=False):
    """toolbox/utils.py:125-144.  src [B,N,C], dst [B,M,C] -> [B,N,M], clamped at 1e-12."""
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    if normalised:
        dist += 2
    else:
        dist += torch.sum(src ** 2, dim=-1)[:, :, None]
        dist += torch.sum(dst ** 2, dim=-1)[:, None, :]
    return torch.clamp(dist, min=1e-12, max=None)


def rotation_error(R1, R2):
    """toolbox/register_utils.py:19-43: acos((tr(R1^T R2) - 1) / 2) in degrees, [b,1]."""
    R_ = torch.matmul(R1.transpose(1, 2), R2)
    e = ((R_.diagonal(dim1=1, dim2=2).sum(-1) - 1) / 2).unsqueeze(1)
    ae = torch.acos(torch.clamp(e, -1, 1))
    return 180. * ae / math.pi


def translation_error(t1, t2):
    """toolbox/register_utils.py:46-56."""
    return torch.norm(t1 - t2, dim=(1, 2))


def get_relative_pose_torch(tsfm_src, tsfm_tgt, dataset):
    """toolbox/register_utils.py:184-197: T_tgt^-1 @ T_src (waymo / nuscene branch)."""
    if dataset not in ('waymo', 'nuscene'):
        raise NotImplementedError('only the waymo / nuscene branch of get_relative_pose_torch is on the hot path')
    return torch.linalg.solve_ex(tsfm_tgt, tsfm_src)[0]          # solve() without the singularity check (a host sync per call)


def kabsch_transformation_estimation(x1, x2, weights=None, normalize_w=True, eps=1e-7, best_k=0, w_threshold=0, fused=False):
    """toolbox/register_utils.py:247-317 (best_k = 0, w_threshold = 0 path).  Returns (R [b,3,3], t [b,3,1], res, flag).
    fused (GPU, weights given, x1 without gradient): means and covariance from ops.kabsch_cov, rotation / translation behind the SVD from
    ops.kabsch_rt -- one kernel each way each; res is not computed."""
    if fused and weights is not None and normalize_w and eps == 1e-7 and x1.is_cuda and not x1.requires_grad:
        cov_mat, x1_mean, x2_mean = ops.kabsch_cov(x1, x2, weights)
        u, s, v = ops.svd3(cov_mat)
        rotation, translation = ops.kabsch_rt(u, v, x1_mean, x2_mean)       # V diag(1, 1, det) U^T and x2_mean - R x1_mean, one thread per pair
        return rotation, translation, None, False
    if weights is None:
        weights = torch.ones(x1.shape[0], x1.shape[1]).type_as(x1).to(x1.device)
    if normalize_w:
        weights = weights / (torch.sum(weights, dim=1, keepdim=True) + eps)
    weights = weights.unsqueeze(2)
    assert best_k == 0 and w_threshold == 0
    wsum = torch.sum(weights, dim=1).unsqueeze(1) + eps
    x1_mean = torch.matmul(weights.transpose(1, 2), x1) / wsum
    x2_mean = torch.matmul(weights.transpose(1, 2), x2) / wsum
    x1_centered = x1 - x1_mean
    x2_centered = x2 - x2_mean
    # diag_embed(w) @ x2c == w * x2c exactly (the dropped terms are +0): no 1024x1024 dense weight matrix
    cov_mat = torch.matmul(x1_centered.transpose(1, 2), weights * x2_centered)
    try:
        # 3x3 Jacobi kernel on the GPU (no status word read back: torch.svd drains the launch queue twice per call)
        u, s, v = ops.svd3(cov_mat) if cov_mat.shape[-2:] == (3, 3) and cov_mat.dim() == 3 else torch.svd(cov_mat)
    except Exception:                                                  # register_utils.py:295-304
        r = torch.eye(3, device=x1.device).repeat(x1_mean.shape[0], 1, 1)
        t = torch.zeros((x1_mean.shape[0], 3, 1), device=x1.device)
        res = torch.norm((torch.matmul(r, x1.transpose(1, 2)) + t).transpose(1, 2) - x2, dim=2)
        return r, t, res, True
    det = torch.det(torch.matmul(v.transpose(1, 2), u.transpose(1, 2)))
    dmat = torch.diag_embed(torch.cat((torch.ones((det.shape[0], 2), device=x1.device), det.unsqueeze(1)), 1))
    rotation = torch.matmul(v, torch.matmul(dmat, u.transpose(1, 2)))
    translation = x2_mean.transpose(1, 2) - torch.matmul(rotation, x1_mean.transpose(1, 2))
    res = torch.norm((torch.matmul(rotation, x1.transpose(1, 2)) + translation).transpose(1, 2) - x2, dim=2)
    return rotation, translation, res, False


class PermList(list):
    """results['perm_matrix']: the per-pair [1,k,k] matrices of the reference (models/egomotion.py:352), plus the [P,k,k] tensor
    they are slices of when one batched solve produced them all (`stacked`), so that the outlier loss is two reductions -- and, from the
    fused matching kernels, the row / column sums of that tensor (`sums`), which are all the loss reads of it."""
    stacked = None
    sums = None
    chains = None       # (estimated, ground-truth) chained poses [B*T,4,4] when one batched solve produced the whole forward's


class _RowRef(object):
    """Rows `index` of a 2-D `source`, not gathered yet: the key-point features of all pairs are fetched with ONE index op,
    so the backward is one index_put into one zero map instead of 32 maps of [n_cells, C] summed pairwise (5 ms per step)."""

    def __init__(self, source, index):
        self.source, self.index = source, index


def _stack_rows(items):
    """[P] list of [k,C] tensors or _RowRef -> [P,k,C]."""
    if isinstance(items[0], _RowRef):
        return items[0].source[torch.stack([it.index for it in items])]
    return torch.stack(items)


class EgoMotionHead(nn.Module):
    """models/egomotion.py:30-469.  Parameters: alpha, beta (scalars, init -5)."""

    def __init__(self, config):
        nn.Module.__init__(self)
        pe = config['pose_estimation']
        self.slack = pe['add_slack']
        self.sinkhorn_iter = pe['sinkhorn_iter']
        self.beta = torch.nn.Parameter(torch.tensor(-5.0))
        self.alpha = torch.nn.Parameter(torch.tensor(-5.0))
        self.softplus = torch.nn.Softplus()
        self.ego_n_points = pe['n_kpts']
        self.frequence = config['data']['freq']
        self.n_sweeps = config['voxel_generator']['n_sweeps']
        self.ego_max_speed = config['data']['max_speed']
        self.dataset = config['data']['dataset']
        self.icp_threshold = pe['icp_threshold']
        self.icp_max_iter = pe['icp_max_iter']
        self.refine_with_icp = config['model']['ego_icp']
        if self.refine_with_icp:
            raise NotImplementedError('model.ego_icp (Open3D ICP refinement) is off the hot path (configs/default.yaml:115)')
        # 'reference': torch.randperm on the HOST generator, the reference's RNG stream (bit-reproducible against it);
        # 'device'   : the same uniform subset drawn with torch.randperm on the GPU generator (no 50 k-element host shuffle,
        #              1.8 ms each on the host, 32 of them per 4-sequence step).
        self.kpt_sampler = pe.get('kpt_sampler', 'reference')
        # the matching stage around the Sinkhorn iterations as four kernels (csrc/ego.hip) instead of the batched torch formulation; default: with
        # the device key-point sampler (the production configuration); the reference-sampler configuration keeps the formulation the
        # parity fixtures were validated with (both are fp32; gradient norms downstream are chaotic at the per-cent level)
        self.fused_matching = bool(pe.get('fused_matching', self.kpt_sampler == 'device'))
        self.seq_pose = pe['seq_pose']
        if self.seq_pose not in ('skip', 'chain', 'full'):
            raise NotImplementedError("pose_estimation.seq_pose='%s' (models/egomotion.py:53-61 knows skip / chain / full)" % self.seq_pose)

    def _pair_plan(self, T):
        """(source frame, target frame, duration) of every registration of one sequence, in the order the reference runs them:
        skip  (models/egomotion.py:309-357): every frame t >= 1 against the anchor frame 0;
        chain (:255-307): frame t + 1 against frame t;
        full  (:195-252): every pair (anchor, anchor + gap), gap-major."""
        if self.seq_pose == 'skip':
            return [(t, 0, t / self.frequence) for t in range(1, T)]
        if self.seq_pose == 'chain':
            return [(t + 1, t, 1.0 / self.frequence) for t in range(T - 1)]
        return [(a + gap, a, gap / self.frequence) for gap in range(1, T) for a in range(T - 1) if a + gap < T]

    def sinkhorn(self, log_alpha, n_iters=5, slack=True):
        """models/egomotion.py:100-137: slack row/column padded with zeros, never normalised themselves."""
        if slack and log_alpha.dim() == 3:
            return ops.sinkhorn(log_alpha, n_iters)                         # fused forward / replayed-backward kernels (csrc/ego.hip)
        la = torch.nn.functional.pad(log_alpha, (0, 1, 0, 1))
        for _ in range(n_iters):
            la = torch.cat((la[:, :-1, :] - torch.logsumexp(la[:, :-1, :], dim=2, keepdim=True), la[:, -1, None, :]), dim=1)
            la = torch.cat((la[:, :, :-1] - torch.logsumexp(la[:, :, :-1], dim=1, keepdim=True), la[:, :, -1, None]), dim=2)
        return la[:, :-1, :-1]

    def _choice(self, n, device=None):
        """models/egomotion.py:156-166: random subset when n > n_kpts, else arange with the tail clamped to n-1."""
        if n > self.ego_n_points:
            if self.kpt_sampler == 'device' and device is not None and device.type == 'cuda':
                return torch.randperm(n, device=device)[:self.ego_n_points]
            return torch.randperm(n)[:self.ego_n_points]
        c = torch.arange(self.ego_n_points)
        c[n:] = n - 1
        return c

    def pairwise_ego_motion_estimation(self, feats_s, feats_t, coor_s, coor_t, duration):
        """models/egomotion.py:140-192.  Returns (pose [4,4], perm_matrix [1,k,k])."""
        choice_source = self._choice(feats_s.size(0)).to(feats_s.device)
        choice_target = self._choice(feats_t.size(0)).to(feats_t.device)
        feats_s_ego, coor_s_ego = feats_s[choice_source][None], coor_s[choice_source][None]
        feats_t_ego, coor_t_ego = feats_t[choice_target][None], coor_t[choice_target][None]
        threshold_distance = duration * self.ego_max_speed
        support_ego = (square_distance(coor_s_ego, coor_t_ego, normalised=False) < threshold_distance ** 2).float()
        feat_dist = square_distance(feats_s_ego, feats_t_ego, normalised=True)
        affinity = -(feat_dist - self.softplus(self.alpha)) / (torch.exp(self.beta) + 0.02)
        log_perm = self.sinkhorn(affinity, n_iters=self.sinkhorn_iter, slack=self.slack)
        perm_matrix = torch.exp(log_perm) * support_ego
        rowsum = torch.sum(perm_matrix, dim=2, keepdim=True)
        weighted_t = perm_matrix @ coor_t_ego / (rowsum + _EPS)
        R_est, t_est, _, _ = kabsch_transformation_estimation(coor_s_ego, weighted_t, weights=rowsum[:, :, 0])
        pose_est = torch.eye(4, device=feats_s.device, dtype=t_est.dtype)
        pose_est[:3, :3] = R_est[0]
        pose_est[:3, 3] = t_est[0][:, 0]
        return pose_est, perm_matrix

    def sequence_pose_est_skip(self, points_list, feats_list, bg_mask_list, c_ego_motion_gt, T, perm_matrix_list,
                               relative_pose_est_list, relative_pose_gt_list, chained_pose_est_list, chained_pose_gt_list):
        """models/egomotion.py:309-357 for ONE batch element: every frame t >= 1 is registered against the anchor
        frame 0.  Kept with the reference's signature; it runs the batched estimator on this element's pairs."""
        return self._sequence_pose_est('skip', points_list, feats_list, bg_mask_list, c_ego_motion_gt, T, perm_matrix_list,
                                       relative_pose_est_list, relative_pose_gt_list, chained_pose_est_list, chained_pose_gt_list)

    def sequence_pose_est_chain(self, *args):
        """models/egomotion.py:255-307: frame t + 1 against frame t, poses chained."""
        return self._sequence_pose_est('chain', *args)

    def sequence_pose_est_full(self, *args):
        """models/egomotion.py:195-252: all pairs; the pairs against frame 0 give the sequence poses."""
        return self._sequence_pose_est('full', *args)

    def _sequence_pose_est(self, mode, points_list, feats_list, bg_mask_list, c_ego_motion_gt, T, *lists):
        keep, self.seq_pose = self.seq_pose, mode
        try:
            seq = self._sequence_from_masks(points_list, feats_list, bg_mask_list, c_ego_motion_gt)
            return self._estimate_pairs([seq], T, *lists)
        finally:
            self.seq_pose = keep

    @staticmethod
    def _sequence_from_masks(points_list, feats_list, bg_mask_list, gt):
        """Dense-map / mask inputs (reference signature) -> the index-list form _estimate_pairs consumes."""
        bg, getters = [], []
        for f, m in zip(feats_list, bg_mask_list):
            idx = torch.nonzero(m)[:, 0]
            bg.append((idx, int(idx.numel())))
            getters.append(lambda i, f=f: f[i])
        return points_list, getters, bg, gt

    def _estimate_pairs(self, sequences, T, perm_matrix_list, relative_pose_est_list, relative_pose_gt_list,
                        chained_pose_est_list, chained_pose_gt_list, normalise=False, flat=None):
        """All (T-1) registrations of all batch elements in ONE batched Sinkhorn / Kabsch evaluation
        ([P,1024,1024] instead of P sequential [1,1024,1024] problems: the reference issues ~60 tiny launches and
        several host syncs per pair).  Key points are still drawn pair by pair in the reference's
        order (source, then target), so with the 'reference' sampler a given torch.manual_seed gives the same draw.

        sequences: list of (points_list, feats_list, bg_list, gt) per batch element, where for frame t
          points_list[t] = xyz of all occupied pillars, feats_list[t] = callable(idx)-> features of pillars `idx` of frame t,
          bg_list[t] = (bg_idx LongTensor into the frame's pillar list, n_bg int)."""
        dev = sequences[0][0][0].device
        plan = self._pair_plan(T)
        fs, cs, ft, ct, durations = [], [], [], [], []
        drawn, d = None, 0
        if self.kpt_sampler == 'device' and dev.type == 'cuda':
            # every key-point draw of the step in one launch (source, then target, per pair -- the reference's order)
            counts = [n for _, _, bg_list, _ in sequences for src, tgt, _ in plan for n in (bg_list[src][1], bg_list[tgt][1])]
            seed = int(torch.empty((), dtype=torch.int64).random_())       # host generator: follows torch.manual_seed
            drawn = native.sample_subsets(native.upload_small(counts, torch.int32, dev), self.ego_n_points, seed)
        if drawn is not None and flat is not None:
            # pillar-level inputs + device sampler: the key points of all pairs in five gathers.  Entry e = (pair, source | target)
            # draws positions in the background list of its frame; bg_sorted_idx holds positions in the cell-ordered pillar list.
            frames = [f for b in range(len(sequences)) for src, tgt, _ in plan for f in (b * T + src, b * T + tgt)]
            offs = native.upload_small([flat['bg_at'][f] for f in frames], torch.int64, dev)
            pillar = flat['sp'][flat['bg_sorted_idx'][offs[:, None] + drawn.long()]]              # [2P,k] pillar ids
            coor = flat['pillar_mean'][pillar]                                                    # [2P,k,3]
            geo = flat['geo_rows']
            if hasattr(geo, 'at'):                                                                # ops.SparseConvRows: the head's last convolution at these cells only
                feats = geo.at(flat['cells'][pillar]).float()                                     # [2P,k,C]
            else:
                feats = geo[flat['cells'][pillar]].float()                                        # [2P,k,C], one index op
            feats_s, feats_t, coor_s, coor_t = feats[0::2], feats[1::2], coor[0::2], coor[1::2]
            durations = [dur for _ in sequences for _, _, dur in plan]
            sequences_loop = []
        else:
            sequences_loop = sequences
        for points_list, feats_list, bg_list, _ in sequences_loop:
            for ref, anchor, dur in plan:
                a_idx, a_n = bg_list[anchor]
                r_idx, r_n = bg_list[ref]
                if drawn is not None:
                    choice_s, choice_t = drawn[d], drawn[d + 1]
                    d += 2
                else:
                    choice_s = self._choice(r_n, dev).to(dev)             # models/egomotion.py:156-166, same order
                    choice_t = self._choice(a_n, dev).to(dev)
                si, ti = r_idx[choice_s], a_idx[choice_t]                   # key-point pillars (indices into the frame lists)
                fs.append(feats_list[ref](si)); cs.append(points_list[ref][si])
                ft.append(feats_list[anchor](ti)); ct.append(points_list[anchor][ti])
                durations.append(dur)
        if sequences_loop:
            feats_s, feats_t = _stack_rows(fs).float(), _stack_rows(ft).float()  # [P,k,C]
            coor_s, coor_t = torch.stack(cs), torch.stack(ct)                    # [P,k,3]
        if normalise:
            # models/motionnet.py:199 divides the WHOLE [B*T,64,Ny,Nx] map by its per-cell L2 norm (no epsilon); only the
            # 2*P*1024 key-point rows are ever read, and the norm is per cell, so normalising the gathered rows is the same
            # arithmetic on 0.03 % of the data (and of the backward).
            feats_s = feats_s / torch.norm(feats_s, p=2, dim=2, keepdim=True)
            feats_t = feats_t / torch.norm(feats_t, p=2, dim=2, keepdim=True)
        thr2 = native.upload_small((torch.tensor(durations, dtype=torch.float32) * self.ego_max_speed) ** 2, torch.float32, dev)
        if not torch.is_grad_enabled():
            # eval / val / test: the fused fp32 HIP pipeline (affinity, Sinkhorn, soft targets, Kabsch + 3x3 SVD)
            params = torch.stack([self.softplus(self.alpha), torch.exp(self.beta) + 0.02]).float().detach()
            perm, pose_est = native.sinkhorn_kabsch(feats_s.contiguous(), feats_t.contiguous(), coor_s.contiguous().float(),
                                                    coor_t.contiguous().float(), thr2.contiguous(), params.contiguous(),
                                                    self.sinkhorn_iter)
            return self._collect_pairs(sequences, T, perm, pose_est, perm_matrix_list, relative_pose_est_list,
                                       relative_pose_gt_list, chained_pose_est_list, chained_pose_gt_list)
        fused = self.fused_matching if os.environ.get('PCACC_EGO_FUSED') is None else os.environ['PCACC_EGO_FUSED'] != '0'
        if fused and feats_s.is_cuda and self.slack and feats_s.dim() == 3:
            # the matching stage around the Sinkhorn iterations as four kernels (csrc/ego.hip): the batched torch formulation below costs
            # ~17 element-wise passes over the [P, k, k] matrices each way
            affinity = ops.ego_affinity(feats_s, feats_t, self.softplus(self.alpha), torch.exp(self.beta) + 0.02)         # :177-180
            perm, rowsum, weighted_t, colsum = ops.ego_perm(self.sinkhorn(affinity, n_iters=self.sinkhorn_iter, slack=True), coor_s, coor_t, thr2)
            if isinstance(perm_matrix_list, PermList):
                perm_matrix_list.sums = (rowsum[:, :, 0], colsum) if len(perm_matrix_list) == 0 else None     # all the outlier loss reads
        else:
            support = (square_distance(coor_s, coor_t, normalised=False) < thr2[:, None, None]).float()     # :173-174
            feat_dist = square_distance(feats_s, feats_t, normalised=True)                                   # :177
            affinity = -(feat_dist - self.softplus(self.alpha)) / (torch.exp(self.beta) + 0.02)              # :180
            perm = torch.exp(self.sinkhorn(affinity, n_iters=self.sinkhorn_iter, slack=self.slack)) * support
            rowsum = torch.sum(perm, dim=2, keepdim=True)
            weighted_t = perm @ coor_t / (rowsum + _EPS)
        R_est, t_est, _, _ = kabsch_transformation_estimation(coor_s, weighted_t, weights=rowsum[:, :, 0], fused=fused)
        P = R_est.shape[0]
        pose_est = torch.eye(4, device=dev, dtype=t_est.dtype).repeat(P, 1, 1)
        pose_est[:, :3, :3] = R_est
        pose_est[:, :3, 3] = t_est[:, :, 0]
        return self._collect_pairs(sequences, T, perm, pose_est, perm_matrix_list, relative_pose_est_list, relative_pose_gt_list,
                                   chained_pose_est_list, chained_pose_gt_list)

    def _collect_pairs(self, sequences, T, perm, pose_est, perm_matrix_list, relative_pose_est_list, relative_pose_gt_list,
                       chained_pose_est_list, chained_pose_gt_list):
        """Per-pair bookkeeping of sequence_pose_est_skip (models/egomotion.py:334-355): GT poses, losses, pose lists."""
        if self.seq_pose != 'skip':
            return self._collect_pairs_general(sequences, T, perm, pose_est, perm_matrix_list, relative_pose_est_list,
                                               relative_pose_gt_list, chained_pose_est_list, chained_pose_gt_list)
        dev = pose_est.device
        P = pose_est.shape[0]
        identity = torch.eye(4, device=dev)
        # T_anchor^-1 @ T_ref, T_(t-1)^-1 @ T_t for the GT and for the estimated chain, all samples in three batched solves
        # (get_relative_pose_torch, register_utils.py:184-197; the reference solves per sample)
        B = len(sequences)
        gt_all = torch.stack([seq[3] for seq in sequences])                                     # [B,T,4,4]
        pose_gt_all = get_relative_pose_torch(gt_all[:, 1:], gt_all[:, 0:1].expand(B, T - 1, 4, 4), self.dataset)
        rel_gt_all = get_relative_pose_torch(gt_all[:, 1:], gt_all[:, :-1], self.dataset)
        chain = torch.cat((identity.expand(B, 1, 4, 4), pose_est.view(B, T - 1, 4, 4)), dim=1)
        rel_est_all = get_relative_pose_torch(chain[:, 1:], chain[:, :-1], self.dataset)
        p = 0
        ref_pts, lens = [], []
        if isinstance(perm_matrix_list, PermList):
            whole = len(perm_matrix_list) == 0
            perm_matrix_list.stacked = perm if whole else None     # all pairs of this forward as one tensor
            # the two stacks _finish would build from the 2*B*T list entries below, as the tensors they are slices of
            perm_matrix_list.chains = (chain.reshape(B * T, 4, 4), torch.cat((identity.expand(B, 1, 4, 4).to(pose_gt_all.dtype), pose_gt_all), dim=1)
                                       .reshape(B * T, 4, 4)) if whole else None
        for b, (points_list, feats_list, bg_list, gt) in enumerate(sequences):
            for lst in (relative_pose_est_list, relative_pose_gt_list, chained_pose_est_list, chained_pose_gt_list):
                lst.append(identity)
            for frame_idx in range(T - 1):
                ref_pts.append(points_list[frame_idx + 1])
                lens.append(points_list[frame_idx + 1].shape[0])
                perm_matrix_list.append(perm[p + frame_idx:p + frame_idx + 1])
                chained_pose_est_list.append(pose_est[p + frame_idx])
                chained_pose_gt_list.append(pose_gt_all[b, frame_idx])
                relative_pose_gt_list.append(rel_gt_all[b, frame_idx])
                relative_pose_est_list.append(rel_est_all[b, frame_idx])
            p += T - 1
        # pc_est - pc_gt of every reference pillar under its pair's two poses (models/egomotion.py:342-346), all pairs in one
        # indexed-transform launch: x -> (R_est - R_gt) x + (t_est - t_gt).  The reference's per-pair matmul has a
        # [3, n] x [n, 3] product in its backward (0.23 ms each on MI355X, 16 per step); here the gradient of the pose
        # table is a segment sum.
        pts = torch.cat(ref_pts, dim=0)
        pair = torch.repeat_interleave(torch.arange(P, device=dev), native.upload_small(lens, torch.int64, dev), output_size=pts.shape[0])
        diff = ops.transform_by_index(pts, pair, pose_est - pose_gt_all.reshape(P, 4, 4).to(pose_est.dtype))
        norms = torch.stack((torch.norm(diff, p=1, dim=1), torch.norm(diff, p=2, dim=1)), dim=1)
        means = ops.scatter(norms, pair, dim=0, dim_size=P, reduce="mean", plan=ops.ScatterPlan(pair, P))
        return means[:, 0].sum(), means[:, 1].sum(), P

    def _collect_pairs_general(self, sequences, T, perm, pose_est, perm_matrix_list, relative_pose_est_list, relative_pose_gt_list,
                               chained_pose_est_list, chained_pose_gt_list):
        """The bookkeeping of sequence_pose_est_chain (models/egomotion.py:255-307) and sequence_pose_est_full (:195-252) on the
        batched solve: ground-truth poses of every pair, the chained / relative pose lists, the point-wise L1 / L2 losses over all
        pillars of each pair's source frame."""
        dev = pose_est.device
        plan = self._pair_plan(T)
        n_pairs, B = len(plan), len(sequences)
        P = pose_est.shape[0]
        identity = torch.eye(4, device=dev)
        src = torch.tensor([p[0] for p in plan], device=dev)
        tgt = torch.tensor([p[1] for p in plan], device=dev)
        gt_all = torch.stack([seq[3] for seq in sequences])                                     # [B,T,4,4]
        pose_gt_all = get_relative_pose_torch(gt_all[:, src], gt_all[:, tgt], self.dataset)     # [B,n_pairs,4,4]
        to_anchor_gt = get_relative_pose_torch(gt_all[:, 1:], gt_all[:, 0:1].expand(B, T - 1, 4, 4), self.dataset)
        rel_gt = get_relative_pose_torch(gt_all[:, 1:], gt_all[:, :-1], self.dataset)
        ref_pts, lens = [], []
        for b, (points_list, feats_list, bg_list, gt) in enumerate(sequences):
            for lst in (relative_pose_est_list, relative_pose_gt_list, chained_pose_est_list, chained_pose_gt_list):
                lst.append(identity)
            chained = identity
            for j, (s, a, _) in enumerate(plan):
                p = b * n_pairs + j
                ref_pts.append(points_list[s])
                lens.append(points_list[s].shape[0])
                if self.seq_pose == 'chain':
                    perm_matrix_list.append(perm[p:p + 1])
                    relative_pose_est_list.append(pose_est[p])
                    chained = chained @ pose_est[p]
                    chained_pose_est_list.append(chained)
                    relative_pose_gt_list.append(pose_gt_all[b, j])
                    chained_pose_gt_list.append(to_anchor_gt[b, s - 1])
                elif a == 0:                                                # 'full': the pairs against frame 0 carry the sequence pose
                    chained_pose_est_list.append(pose_est[p])
                    chained_pose_gt_list.append(pose_gt_all[b, j])
                    relative_pose_gt_list.append(rel_gt[b, s - 1])
                    relative_pose_est_list.append(get_relative_pose_torch(chained_pose_est_list[-1], chained_pose_est_list[-2], self.dataset))
                    perm_matrix_list.append(perm[p:p + 1])
        pts = torch.cat(ref_pts, dim=0)
        pair = torch.repeat_interleave(torch.arange(P, device=dev), native.upload_small(lens, torch.int64, dev), output_size=pts.shape[0])
        diff = ops.transform_by_index(pts, pair, pose_est - pose_gt_all.reshape(P, 4, 4).to(pose_est.dtype))
        norms = torch.stack((torch.norm(diff, p=1, dim=1), torch.norm(diff, p=2, dim=1)), dim=1)
        means = ops.scatter(norms, pair, dim=0, dim_size=P, reduce="mean", plan=ops.ScatterPlan(pair, P))
        return means[:, 0].sum(), means[:, 1].sum(), P

    def _finish(self, B, T, total_l1, total_l2, count, perm_matrix_list, chained_pose_est_list, chained_pose_gt_list, results):
        """models/egomotion.py:448-469."""
        chains = getattr(perm_matrix_list, 'chains', None)
        if chains is not None and chains[0].shape[0] == len(chained_pose_est_list):
            chained_pose_est, chained_pose_gt = chains
        else:
            chained_pose_est = torch.stack(chained_pose_est_list)
            chained_pose_gt = torch.stack(chained_pose_gt_list)
        rot_est, rot_gt = chained_pose_est[:, :3, :3], chained_pose_gt[:, :3, :3]
        trans_est, trans_gt = chained_pose_est[:, :3, 3].unsqueeze(-1), chained_pose_gt[:, :3, 3].unsqueeze(-1)
        # 0-d tensors here; MotionNet.forward turns them into Python floats (the reference's .item(), egomotion.py:456)
        # together with the other scalar results in ONE host sync at the end of the forward
        rot_error = rotation_error(rot_est, rot_gt).mean() * self.n_sweeps / (self.n_sweeps - 1)
        trans_error = translation_error(trans_est, trans_gt).mean() * self.n_sweeps / (self.n_sweeps - 1)
        results['ego_l1_loss'] = total_l1 / count
        results['ego_l2_loss'] = total_l2 / count
        results['ego_rot_error'] = rot_error
        results['ego_trans_error'] = trans_error
        results['perm_matrix'] = perm_matrix_list
        results['ego_motion_est'] = chained_pose_est.view(B, T, 4, 4)
        results['ego_motion_gt'] = chained_pose_gt.view(B, T, 4, 4)

    def forward_pillars(self, geo_rows, pillar_mean, pidx, ego_motion_gt, results, frame_offsets, bg_sorted_idx, bg_counts):
        """Same computation as forward(), fed from pillar-level tensors instead of dense canvases, with no host sync:
          geo_rows [n_cells, C]  rows of the (un-normalised) ego feature map, normalised after the key-point gather;  pillar_mean [M,3]
          frame_offsets  host list [B*T+1]: slice of the cell-ordered pillar list (pidx.frame_pillars) per frame
          bg_sorted_idx  LongTensor: positions in that cell-ordered list whose pillar is predicted background
          bg_counts      host list [B*T]: how many of them fall into each frame
        Frame lists enumerate occupied pillars in ascending cell order, as models/egomotion.py:419-424 does through the
        boolean occupancy mask."""
        B, T = pidx.batch_size, pidx.nt
        sp = pidx.frame_pillars()[0].long()
        cells = pidx.cell.long()
        perm_l, rel_est, rel_gt, ch_est, ch_gt = PermList(), [], [], [], []
        sequences, bg_start, bg_at = [], 0, []
        # all key points in five gathers (_estimate_pairs); flat_keypoints = False keeps the per-pair loop (tests compare the two)
        fast = self.kpt_sampler == 'device' and geo_rows.is_cuda and getattr(self, 'flat_keypoints', True)
        if not fast and hasattr(geo_rows, 'dense'):
            geo_rows = geo_rows.dense()                                      # whole frames are indexed below: the dense map as rows
        mean_sorted = pillar_mean[sp]                                        # xyz of every pillar in frame / cell order: one gather
        for b in range(B):
            points_list, getters, bg_list = [], [], []
            for t in range(T):
                f = b * T + t
                lo, hi = frame_offsets[f], frame_offsets[f + 1]
                points_list.append(mean_sorted[lo:hi])
                n_bg = bg_counts[f]
                bg_at.append(bg_start)
                if fast:
                    bg_list.append((None, n_bg))
                    getters.append(None)
                else:
                    ids = sp[lo:hi]                                          # pillar ids of this frame, cell order
                    getters.append(lambda i, ids=ids: _RowRef(geo_rows, cells[ids[i]]))
                    bg_list.append((bg_sorted_idx[bg_start:bg_start + n_bg] - lo, n_bg))
                bg_start += n_bg
            sequences.append((points_list, getters, bg_list, ego_motion_gt[b]))
        flat = dict(sp=sp, cells=cells, geo_rows=geo_rows, pillar_mean=pillar_mean, bg_sorted_idx=bg_sorted_idx, bg_at=bg_at) if fast else None
        total_l1, total_l2, count = self._estimate_pairs(sequences, T, perm_l, rel_est, rel_gt, ch_est, ch_gt, normalise=True, flat=flat)
        self._finish(B, T, total_l1, total_l2, count, perm_l, ch_est, ch_gt, results)

    def forward(self, bev_feats, fb_est, occ_map, pts_mean_map, ego_motion_gt, input_points, fb_est_per_point, time_indice, results):
        """Reference signature (models/egomotion.py:387-469): dense [B,T,C,Ny,Nx] maps in, results dict filled."""
        B, T, C, Ny, Nx = bev_feats.size()
        perm_l, rel_est, rel_gt, ch_est, ch_gt = PermList(), [], [], [], []
        sequences = []
        for b in range(B):
            points_list, feats_list, bg_list = [], [], []
            for t in range(T):
                occ = occ_map[b, t, 0].reshape(-1) > 0
                points_list.append(pts_mean_map[b, t].permute(1, 2, 0).reshape(Ny * Nx, 3)[occ])
                feats_list.append(bev_feats[b, t].permute(1, 2, 0).reshape(Ny * Nx, C)[occ])
                bg_list.append((fb_est[b, t, 0].reshape(-1) == 0)[occ])
            sequences.append(self._sequence_from_masks(points_list, feats_list, bg_list, ego_motion_gt[b]))
        total_l1, total_l2, count = self._estimate_pairs(sequences, T, perm_l, rel_est, rel_gt, ch_est, ch_gt)
        self._finish(B, T, total_l1, total_l2, count, perm_l, ch_est, ch_gt, results)
        for k in ('ego_rot_error', 'ego_trans_error'):                       # reference signature returns floats
            results[k] = results[k].item()
