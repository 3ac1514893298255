"""Training loss and the IoU bookkeeping that defines `mos_iou`: host mirror of libs/loss.py (FuseLoss,
compute_iou), libs/lovasz_softmax.py (Lovasz-Softmax, Berman et al. 2018, MIT) and libs/outlier_loss.py.

Needed so that BASELINE.json's "fwd+bwd" metric runs end to end.  The terms that read the path's large tensors -- the two
segmentation losses (cross entropy + Lovasz + IoU counters) and the offset loss -- are fused HIP passes (SURVEY.md 8f rank 3,
csrc/loss.hip, ops.seg_loss / ops.offset_loss); the rest is a handful of scalar operations.  The cluster-evaluation hook (test
mode only) is not carried over.
"""
import numpy as np
import torch
import torch.nn as nn

from . import native, ops
from .lazy import HostCopy, LazyDict, LazyValue, raw
from .tpointnet import ego_motion_compensation, reconstruct_sequence

_EPS = 1e-20


def compute_iou(predictions, gt, n_class, ignore_index):
    """libs/loss.py:17-50: per-class intersection / union / positives, each divided by 1e3."""
    inter, union, pred_pos, gt_pos = [], [], [], []
    for idx in range(n_class):
        if idx == ignore_index:
            continue
        sel_gt, sel_pred = gt == idx, predictions == idx
        n_pred, n_gt = sel_pred.sum().item() / 1e3, sel_gt.sum().item() / 1e3
        i = (predictions[sel_gt] == idx).sum().item() / 1e3
        pred_pos.append(n_pred)
        gt_pos.append(n_gt)
        inter.append(i)
        union.append(n_pred + n_gt - i)
    return {'intersection': np.array(inter), 'union': np.array(union),
            'pred_positives': np.array(pred_pos), 'gt_positives': np.array(gt_pos)}


def _metric_dict(t):
    t = t.cpu().numpy() if torch.is_tensor(t) else t
    return {'intersection': t[0], 'union': t[1], 'pred_positives': t[2], 'gt_positives': t[3]}


def mean_iou(stats_list):
    """toolbox/metrics.py:43-60 over a list of per-batch stats: sum(I) / (sum(U) + 1e-20), mean over classes."""
    i = sum(s['intersection'] for s in stats_list)
    u = sum(s['union'] for s in stats_list)
    return float((i / (u + _EPS)).mean())


def lovasz_grad(gt_sorted):
    """Gradient of the Lovasz extension of the Jaccard loss w.r.t. the sorted errors (Berman et al. 2018, alg. 1; libs/lovasz_softmax.py:56-68
    is the same formula).  With G = #foreground and, after rank r (inclusive), f_r foreground and b_r background members, the Jaccard value of
    the first r + 1 errors is J_r = 1 - (G - f_r) / (G + b_r); the result is J_0, J_1 - J_0, J_2 - J_1, ...  (fallback path only: the
    product's loss runs in csrc/loss.hip, seg_lovasz_kernel, which forms the same differences per rank)."""
    fg = gt_sorted.to(torch.float32)
    ranks = torch.arange(1, fg.numel() + 1, device=fg.device, dtype=torch.float32)
    f = torch.cumsum(fg, 0)
    total = f[-1] if fg.numel() else fg.sum()
    j = 1.0 - (total - f) / (total + (ranks - f))
    return torch.diff(j, prepend=j.new_zeros(1))


def lovasz_softmax_flat(probas, labels):
    """libs/lovasz_softmax.py:71-94: mean over PRESENT classes of <sorted errors, Lovasz gradient>.  The reference skips
    absent classes with a host-side `if fg.sum() == 0`; here every class is evaluated and weighted by its presence on the
    device (same value, no sync)."""
    if probas.numel() == 0:
        return probas * 0.
    total, present = 0, 0
    for c in range(probas.size(1)):
        fg = (labels == c).float()
        here = (fg.sum() > 0).to(probas.dtype)
        errors = (fg - probas[:, c]).abs()
        errors_sorted, perm = torch.sort(errors, 0, descending=True)
        grad = torch.nan_to_num(lovasz_grad(fg[perm.data]), nan=0.0)           # gts == 0 gives 0/0 in the first entries
        total = total + here * torch.dot(errors_sorted, grad)
        present = present + here
    return total / torch.clamp(present, min=1.0)


class Lovasz_softmax(nn.Module):
    def forward(self, probas, labels):
        return lovasz_softmax_flat(probas, labels)


class OutlierLoss(object):
    """libs/outlier_loss.py: mean mass missing from the rows and columns of each permutation matrix."""

    def __call__(self, perm_matrix):
        sums = getattr(perm_matrix, 'sums', None)
        if sums is not None and getattr(perm_matrix, 'stacked', None) is not None:     # row sums [P,k] (dim 2) and column sums [P,k] (dim 1) from the kernel
            return torch.mean(1.0 - sums[1]) + torch.mean(1.0 - sums[0])
        stacked = getattr(perm_matrix, 'stacked', None)
        if stacked is not None:                                            # equal-sized pairs: the means of the concatenations
            return torch.mean(1.0 - torch.sum(stacked, dim=1)) + torch.mean(1.0 - torch.sum(stacked, dim=2))
        ref = torch.cat([1.0 - torch.sum(p, dim=1) for p in perm_matrix], 1)
        src = torch.cat([1.0 - torch.sum(p, dim=2) for p in perm_matrix], 0)
        return torch.mean(ref) + torch.mean(src)


class FuseLoss(nn.Module):
    def __init__(self, config):
        super(FuseLoss, self).__init__()
        self.outlier_loss = OutlierLoss()
        self.lovasz_loss = Lovasz_softmax()
        self.n_classes = 2
        self.ignore_index = -1
        self.softmax = nn.Softmax(dim=1)
        self._w_cache = {}
        for k, v in config.items():
            if k.startswith('w_') or k == 'obj_gamma':
                setattr(self, k, v)

    def get_ce_weights(self, gt_label, max_weights=50):
        """libs/loss.py:90-108, 'sqrt_inv_freq' mode (counts stay on the device: no .item())."""
        counts = torch.stack([(gt_label == c).sum() for c in range(self.n_classes)]).to(torch.float32) + _EPS
        return torch.clamp(torch.sqrt(counts.sum() / counts), 0, max_weights)

    def _weights(self, device, *names):
        """Loss weights as a device vector (built once): the kernels return their terms as one tensor and the weighted sum is
        one dot product."""
        key = (str(device),) + names
        if key not in self._w_cache:
            self._w_cache[key] = native.upload_small([float(getattr(self, n)) if n else 0.0 for n in names], torch.float32, device)
        return self._w_cache[key]

    def get_seg_loss(self, gt, est, rows=None):
        """libs/loss.py:110-137 on rows `rows` of est / gt (all rows when None): one fused pass (csrc/loss.hip, L1) for the
        weighted cross entropy (class weights of :90-108), the Lovasz-Softmax term and compute_iou's counters.  'terms' is the
        differentiable [2] tensor (cross entropy, Lovasz); 'metric' stays on the device until forward()'s single transfer."""
        terms, metric = ops.seg_loss(est, gt, rows)
        return {'terms': terms, 'bce_loss': terms.detach()[0], 'lovasz_loss': terms.detach()[1], 'metric': metric}

    def _empty_seg(self, device):
        zero = {k: np.zeros(2) for k in ('intersection', 'union', 'pred_positives', 'gt_positives')}
        z = torch.zeros(2, device=device, requires_grad=True)
        return {'metric': zero, 'terms': z, 'bce_loss': z.detach()[0], 'lovasz_loss': z.detach()[1]}

    def get_mos_loss(self, predictions, input_dict):
        """libs/loss.py:140-165: supervised on points that are foreground in GT or in the estimate."""
        mos_gt, mos_est = input_dict['sd_labels'][:, 0], predictions['mos_est']
        if '_mos_idx' in predictions:                                     # index list MotionNet already built (train / val: no re-sync)
            fb_idx = predictions['_mos_idx']
        else:
            fb_idx = torch.nonzero(torch.logical_or(input_dict['fb_labels'][:, 0] == 1, predictions['fb_est_per_points'][:, 0] == 1))[:, 0]
        if fb_idx.numel():
            return self.get_seg_loss(mos_gt, mos_est, fb_idx)
        return self._empty_seg(mos_est.device)

    def get_fb_loss(self, predictions):
        """libs/loss.py:167-191: only occupied pillars are supervised.  The [B,T,2,H,W] head output is read in place (no
        permute copy); rows = the cell index of every occupied pillar."""
        est, gt = predictions['fb_seg_est'], predictions['fb_seg_gt']
        if '_cell' in predictions:                                        # occupied cells = the pillars' cell indices
            return self.get_seg_loss(gt, est, predictions['_cell'])
        return self.get_seg_loss(gt, est, torch.nonzero(predictions['occ_map'].reshape(-1) == 1)[:, 0])

    def get_offset_loss(self, input_dict, predictions):
        """libs/loss.py:194-250: one fused pass (csrc/loss.hip, L2).  GT reconstruction and instance centres of all samples at
        once (the reference loops over samples with boolean masks, :216-232: four host syncs per sample): points carry their
        sample in time_indice[:, 0], the instance tables are concatenated with per-sample label offsets known on the host."""
        input_points = input_dict['input_points']
        bbox_tsfm = input_dict['inst_motion_gt']
        device = input_points.device
        if '_gtfg_idx' in predictions:
            rows = predictions['_gtfg_idx']                                # index list of the GT-foreground points (train / val)
        else:
            rows = torch.nonzero(input_dict['fb_labels'][:, 0] == 1)[:, 0]
        if rows.numel() == 0:
            z = torch.zeros(3, device=device, requires_grad=True)
            return z, None
        sizes = [m.shape[0] for m in bbox_tsfm]
        base = native.upload_small([sum(sizes[:b]) for b in range(len(sizes))], torch.int64, device)
        out, gt_offset = ops.offset_loss(predictions['offset_est'], input_points, input_dict['time_indice'], input_dict['inst_labels'][:, 0], base,
                                         input_dict['ego_motion_gt'], torch.cat([m.to(device) for m in bbox_tsfm], dim=0),
                                         predictions['transformed_points'], rows)
        predictions['offset_gt'] = gt_offset
        return out, gt_offset

    def get_tpointnet_loss(self, predictions):
        """libs/loss.py:253-263."""
        total, n_th = 0, 1
        n_it = len(predictions['tpointnet_loss_terms'])
        for value in predictions['tpointnet_loss_terms'].values():
            pose_loss = self.w_obj_trans_loss * value['trans_loss'] + self.w_obj_rot_loss * value['rot_loss']
            total = total + (self.w_obj_l1_loss * value['l1_loss'] + self.w_obj_pose_loss * pose_loss) * self.obj_gamma ** (n_it - n_th)
            n_th += 1
        return total

    def early_terms(self, predictions):
        """The terms of forward() that only need the ego head, the fg/bg head and what lies below them (ego pose loss, permutation
        loss, fg/bg segmentation): available as soon as MotionNet's ego head has run, i.e. BEFORE the motion branch and the TubeNet.
        A training step can back-propagate them right there (DataParallelStep, `pipelined`): the backward of the pillar encoder and
        the U-Net -- most of the step's GPU work -- is then queued in front of the TubeNet's hundreds of small launches, which the
        host issues while the GPU is busy instead of starving it.  Returns a LazyDict with these stats and 'loss_early'."""
        stats = LazyDict()
        ego = self.w_pose_l1_loss * predictions['ego_l1_loss']
        total = ego
        stats['ego_l1_loss'] = ego
        stats['ego_l2_loss'] = predictions['ego_l2_loss']
        perm_loss = self.outlier_loss(predictions['perm_matrix']) * self.w_perm_loss
        total = total + perm_loss
        stats['perm_loss'] = perm_loss
        fb = self.get_fb_loss(predictions)
        fb_loss = torch.dot(fb['terms'], self._weights(total.device, 'w_fb_bce_loss', 'w_fb_lovasz_loss'))
        total = total + fb_loss
        stats['fb_loss'], stats['fb_metric'] = fb_loss, fb['metric']
        stats['loss_early'] = total
        return stats

    def forward(self, predictions, input_dict, early=None):
        """libs/loss.py:280-327.  `early`: the result of early_terms() when those terms were already evaluated (and possibly
        back-propagated) -- they enter the total as constants then."""
        if early is None:
            early = self.early_terms(predictions)
            total = early['loss_early']
        else:
            total = early['loss_early'].detach()
        stats = LazyDict()
        for k in ('ego_l1_loss', 'ego_l2_loss', 'perm_loss', 'fb_loss', 'fb_metric'):
            stats[k] = raw(early, k)
        stats['ego_rot_error'] = raw(predictions, 'ego_rot_error')            # still in flight (lazy.py): carried along unread
        stats['ego_trans_error'] = raw(predictions, 'ego_trans_error')
        dev = total.device
        mos = self.get_mos_loss(predictions, input_dict)
        mos_loss = torch.dot(mos['terms'], self._weights(dev, 'w_mos_bce_loss', 'w_mos_lovasz_loss'))
        total = total + mos_loss
        stats['mos_loss'], stats['mos_metric'] = mos_loss, mos['metric']
        off, _ = self.get_offset_loss(input_dict, predictions)
        offset_loss = torch.dot(off, self._weights(dev, 'w_offset_norm_loss', 'w_offset_dir_loss', None))
        total = total + offset_loss
        off = off.detach()
        stats.update(offset_loss=offset_loss, offset_l1_loss=off[0], offset_dir_loss=off[1], offset_l2_error=off[2])
        if 'tpointnet_loss_terms' in predictions:
            obj_loss = self.get_tpointnet_loss(predictions) * self.w_obj_loss
            total = total + obj_loss
            stats['obj_loss'] = obj_loss
            stats['inst_l2_error'] = raw(predictions, 'inst_l2_error')
            stats['dynamic_inst_l2_error'] = raw(predictions, 'dynamic_inst_l2_error')
        stats['loss'] = total
        # one asynchronous device->host transfer for everything the reference reads with .item() (loss.py:30-35, 226); the
        # values materialise when they are first read from `stats` (lazy.py), i.e. after the backward pass has been queued
        pend = [k for k in ('fb_metric', 'mos_metric', 'offset_l2_error') if torch.is_tensor(dict.__getitem__(stats, k))]
        if pend:
            flat = [dict.__getitem__(stats, k).detach().double().reshape(-1) for k in pend]
            copy = HostCopy(torch.cat(flat))
            off = 0
            for k, f in zip(pend, flat):
                conv = (lambda v: float(v[0])) if k == 'offset_l2_error' else (lambda v: _metric_dict(v.reshape(4, -1)))
                stats[k] = LazyValue(copy, off, off + f.numel(), conv)
                off += f.numel()
        return stats


def scene_flow_epe(predictions, input_dict, n_frames):
    """libs/tester.py:58-77: per-point end-point error ||(rec_est - x) - (rec_gt - x)|| for points with t > 0
    (batch size 1, as in SegTrainer.test)."""
    x = input_dict['input_points'].float()      # the tester multiplies by ego_motion_gt.float(): fp32 throughout
    t = input_dict['time_indice'][:, 1].long()
    ego = input_dict['ego_motion_gt'].float()[0]
    comp = ego_motion_compensation(x, t, ego)
    rec_gt = reconstruct_sequence(comp, t, input_dict['inst_labels'][:, 0], input_dict['inst_motion_gt'][0].to(x.device).float(), n_frames)
    err = torch.norm((predictions['rec_est'] - x) - (rec_gt - x), p=2, dim=1)
    return err[t > 0]
