"""Instance-wise reconstruction wrapper: host mirror of models/alignnet.py (AlignNet, update_gt_inst_motion).
Plain PyTorch-ROCm; the Open3D ICP refinements (`model.tpointnet_icp`) are off the hot path."""
import torch

from . import native
from .ops import scatter
from .tpointnet import TPointNet, BaseModel, reconstruct_sequence

_EPS = 1e-20


def update_gt_inst_motion(inst_motion_gt, ego_motion_gt, ego_motion_est):
    """models/alignnet.py:9-38: GT instance motion expressed relative to the ESTIMATED ego motion, per sample
    `motion @ ego_gt @ inv(ego_est)` -- evaluated for all samples at once (one inverse launch, two batched products over the
    concatenated instance table) and handed back as the reference's list of per-sample views."""
    device = ego_motion_gt.device
    sizes = [m.size(0) for m in inst_motion_gt]
    T = ego_motion_gt.size(1)
    if sum(sizes) == 0:
        return [m.to(device).float().view(0, T, 4, 4) for m in inst_motion_gt]
    motion = (torch.cat([m.to(device) for m in inst_motion_gt], dim=0) if len(sizes) > 1 else inst_motion_gt[0].to(device)).float()     # one conversion, not one per sample
    est_inv = native.inv4x4(ego_motion_est.detach().float())                   # inv() minus its host sync and its dozen launches
    if len(sizes) > 1:
        sample = native.upload_small([b for b, k in enumerate(sizes) for _ in range(k)], torch.int64, device)
        gt, inv = ego_motion_gt.float().index_select(0, sample), est_inv.index_select(0, sample)          # [K, T, 4, 4]
    else:
        gt, inv = ego_motion_gt.float().expand(sizes[0], -1, -1, -1), est_inv.expand(sizes[0], -1, -1, -1)
    out = torch.matmul(torch.matmul(motion, gt), inv)                          # the reference's association: (motion @ gt) @ inv
    return list(out.split(sizes, dim=0))


class AlignNet(BaseModel):
    def __init__(self, config):
        BaseModel.__init__(self, config)
        self.alignment = TPointNet(config)
        self.n_iterations = config['tpointnet']['n_iterations']
        self.pc_range = config['voxel_generator']['range']
        self.icp_threshold = config['tpointnet']['icp_threshold']
        self.mode = config['misc']['mode']
        self.refine_with_icp = config['model']['tpointnet_icp']
        if self.refine_with_icp:
            raise NotImplementedError('model.tpointnet_icp (Open3D ICP) is off the hot path (configs/default.yaml:117)')

    @staticmethod
    def padding_flags(inst_indice, time_indice, K, T, weights=None):
        """(orphan, alive) per instance as one [2,K] device tensor: alive = has points, orphan = has points but none in the anchor
        frame (models/alignnet.py:121-140).  `weights` restricts the count to a subset (1 = counted) so that MotionNet can
        evaluate the flags on all points before its one host sync of the forward."""
        slot = (inst_indice * T + time_indice).long()
        ones = torch.ones(slot.size(0), device=slot.device) if weights is None else weights.float()
        per_slot = scatter(ones, slot, dim=0, dim_size=K * T, reduce='sum')
        per_inst = per_slot.view(K, T).sum(dim=1)
        return torch.stack(((per_slot[::T] == 0) & (per_inst > 0), per_inst > 0)), per_slot, slot

    def padding(self, inst_indice, time_indice, inst_motion, flags=None):
        """models/alignnet.py:115-163.  Returns (extra point indices or None, motions of the non-empty instances,
        compacted instance label per point).  Instances that have points but none in the anchor frame get the points of
        their first populated frame appended as frame-0 stand-ins.  `flags` = padding_flags already on the host."""
        device = inst_indice.device
        K, T = inst_motion.size(0), inst_motion.size(1)
        if flags is None:
            dev_flags, per_slot, slot = self.padding_flags(inst_indice, time_indice, K, T)
            flags = dev_flags.cpu()                                          # the ONE host sync of this step
        else:
            per_slot = slot = None
        if bool(flags[0].any()) and slot is None:                            # rare: an instance without anchor-frame points
            _, per_slot, slot = self.padding_flags(inst_indice, time_indice, K, T)
        extra = []
        for k in torch.where(flags[0])[0].tolist():
            first = torch.where(per_slot[k * T:(k + 1) * T] > 0)[0][0]
            extra.append(torch.where(slot == k * T + first)[0])
        extra = torch.cat(extra) if extra else None
        alive_idx = torch.where(flags[1])[0]
        relabel = -1 * torch.ones(K).long()
        relabel[alive_idx] = torch.arange(alive_idx.numel())
        table = native.upload_small(relabel, torch.int64, device)
        if inst_indice.is_cuda and inst_indice.numel():                             # a row gather of the small table (the generic index kernel: 150 us at 320 k points)
            compact = native.gather_rows(table.view(-1, 1), inst_indice.to(torch.int32)).view(-1)
        else:
            compact = table[inst_indice]                                            # every point's instance is alive by construction
        return extra, inst_motion[native.upload_small(alive_idx, torch.int64, device)], compact

    def _merge_batch_instances(self, labels, batch_col, motions):
        """Instance ids of sample b are shifted by the number of instances in samples < b (alignnet.py:199-206)."""
        # The reference skips samples without points when it advances the base (`if sel.sum()`); with points sorted by sample and
        # every sample of a collated batch non-empty this is the running sum of the instance counts, known on the host.
        sizes = [m.size(0) for m in motions]
        base = native.upload_small([sum(sizes[:b]) for b in range(len(sizes))], labels.dtype, labels.device)
        if labels.is_cuda and labels.numel():
            return labels + native.gather_rows(base.view(-1, 1), batch_col.to(torch.int32)).view(-1), torch.cat(motions)
        return labels + base[batch_col.long()], torch.cat(motions)

    def forward(self, input_dict, results):
        """models/alignnet.py:166-285: iterative per-instance pose regression on the foreground points."""
        moving = input_dict['mos_labels']
        labels = input_dict['inst_labels'].clone()
        tcol = input_dict['time_indice']
        start_points = input_dict['transformed_points'].clone()
        device = labels.device
        n_pts = labels.size(0)
        gt_list = input_dict['inst_motion_gt']
        if self.mode == 'test':
            gt_list = [torch.eye(4)[None, None].repeat(labels.max() + 1, self.n_frames, 1, 1).to(device)]
        # GT instance motion relative to the ESTIMATED ego motion, all samples merged into one instance table
        labels, remaining = self._merge_batch_instances(
            labels, tcol[:, 0], update_gt_inst_motion(gt_list, input_dict['ego_motion_gt'], input_dict['ego_motion_est']))
        extra, remaining, labels = self.padding(labels, tcol[:, 1], remaining, input_dict.get('_pad_flags'))
        gt_motion = remaining
        K, T = remaining.size(0), remaining.size(1)

        take = torch.arange(n_pts, device=device).long()
        frames = tcol[:, 1]
        if extra is not None:                                              # stand-in points live in frame 0
            frames = torch.cat((frames, torch.zeros_like(extra).long()))
            take = torch.cat((take, extra))
        # without stand-in points `take` is the identity: no gather (and no index_put in its backward) of the feature rows
        rows = (lambda x: x) if extra is None else (lambda x: x.index_select(0, take))
        p_labels, cloud = rows(labels), rows(start_points)
        net_in = {'frame_feats': rows(input_dict['backbone_feats']), 'time_indice': frames, 'inst_labels': p_labels,
                  'mos_labels': rows(moving), 'mos_feats': rows(input_dict['motion_feats'])}

        results['tpointnet_loss_terms'] = dict()
        total = None
        net_in['inst_motion_gt'] = remaining
        shared = self.alignment.shared_terms(net_in)                       # what the iterations have in common, once
        for it in range(self.n_iterations):
            net_in['points'] = cloud.detach()
            net_in['inst_motion_gt'] = remaining.detach()
            net_in['total'] = total
            out = self.alignment(net_in, shared)
            results['tpointnet_loss_terms'][f'{it}_th'] = out
            step = out['inst_est_motion']                                  # [K,T,4,4]
            cloud = reconstruct_sequence(cloud, frames, p_labels, step, T)
            # what is left of the GT motion after this step (remaining <- remaining @ step^-1) and the composed estimate
            # (total <- step @ total), alignnet.py:257-263: both come out of the slot kernel
            remaining, total = out['remaining'], out['total']

        src = input_dict['transformed_points']
        moved_est = reconstruct_sequence(src, tcol[:, 1], labels, total, T)
        moved_gt = reconstruct_sequence(src, tcol[:, 1], labels, gt_motion, T)
        err = torch.norm(moved_est - moved_gt, p=2, dim=1)
        later = tcol[:, 1] > 0
        later_moving = (moving == 1) & later
        # 0-d tensors; MotionNet.forward converts them to floats in its single end-of-forward sync (alignnet.py:280-281)
        results['inst_l2_error'] = (err * later).sum() / (later.sum() + _EPS)
        results['dynamic_inst_l2_error'] = (err * later_moving).sum() / (later_moving.sum() + _EPS)
        results['inst_labels_adjusted'] = labels
        results['inst_pose_est'] = total
        results['sub_rec_est'] = moved_est
        return results
