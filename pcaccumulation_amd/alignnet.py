"""Instance-wise reconstruction wrapper: host mirror of models/alignnet.py (AlignNet, update_gt_inst_motion).
Plain PyTorch-ROCm; the Open3D ICP refinements (`model.tpointnet_icp`) are off the hot path."""
import torch

from .ops import scatter
from .tpointnet import TPointNet, BaseModel, reconstruct_sequence

_EPS = 1e-20


def update_gt_inst_motion(inst_motion_gt, ego_motion_gt, ego_motion_est):
    """models/alignnet.py:9-38: GT instance motion expressed relative to the ESTIMATED ego motion."""
    device = ego_motion_gt.device
    out = []
    for b, motion in enumerate(inst_motion_gt):
        motion = motion.to(device).float()
        K = motion.size(0)
        gt = ego_motion_gt[b][None].repeat(K, 1, 1, 1).view(-1, 4, 4)
        est = ego_motion_est[b][None].repeat(K, 1, 1, 1).view(-1, 4, 4)
        out.append((motion.view(-1, 4, 4) @ gt @ torch.linalg.inv(est)).view(K, -1, 4, 4))
    return out


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

    def padding(self, inst_indice, time_indice, inst_motion):
        """models/alignnet.py:115-163: drop empty instances, relabel, and for instances without anchor-frame
        points duplicate the points of their first populated frame as frame-0 padding."""
        device = inst_indice.device
        K, T, _, _ = inst_motion.size()
        frame_indice = (inst_indice * T + time_indice).long()
        count = torch.ones(frame_indice.size(0), device=device)
        frame_count = scatter(count, frame_indice, dim=0, dim_size=K * T, reduce='sum')
        inst_count = scatter(count, inst_indice, dim=0, dim_size=K, reduce='sum')
        anchor_count = frame_count[::T]
        padding_list = []
        sel_inst = (anchor_count == 0) & (inst_count > 0)
        if sel_inst.sum():
            for inst_idx in torch.where(sel_inst)[0].tolist():
                c_count = frame_count[inst_idx * T:(inst_idx + 1) * T]
                pad_frame_idx = inst_idx * T + torch.where(c_count > 0)[0][0]
                padding_list.append(torch.where(frame_indice == pad_frame_idx)[0])
        padding_indice = torch.cat(padding_list) if len(padding_list) else None
        sel_inst = inst_count > 0
        inst_motion = inst_motion[sel_inst]
        mapping = -1 * torch.ones(K).long()
        mapping[sel_inst.cpu()] = torch.arange(int(sel_inst.sum()))
        updated = mapping.to(device)[inst_indice]
        assert updated.min() != -1
        return padding_indice, inst_motion, updated

    def forward(self, input_dict, results):
        """models/alignnet.py:166-285."""
        mos_labels = input_dict['mos_labels']
        inst_labels = input_dict['inst_labels'].clone()
        time_indice = input_dict['time_indice']
        transformed_points = input_dict['transformed_points'].clone()
        inst_motion_gt = input_dict['inst_motion_gt']
        backbone_feats = input_dict['backbone_feats']
        mos_feats = input_dict['motion_feats']
        ego_motion_est, ego_motion_gt = input_dict['ego_motion_est'], input_dict['ego_motion_gt']
        device = inst_labels.device
        n_points = inst_labels.size(0)

        if self.mode == 'test':
            n_instance = inst_labels.max() + 1
            inst_motion_gt = [torch.eye(4)[None, None].repeat(n_instance, self.n_frames, 1, 1).to(device)]
        updated_inst_motion = update_gt_inst_motion(inst_motion_gt, ego_motion_gt, ego_motion_est)

        running_idx = 0
        for b in range(len(updated_inst_motion)):
            sel = time_indice[:, 0] == b
            if sel.sum():
                inst_labels[sel] += running_idx
                running_idx += updated_inst_motion[b].size(0)
        updated_inst_motion = torch.cat(updated_inst_motion)

        padding_indice, updated_inst_motion, inst_labels = self.padding(inst_labels, time_indice[:, 1], updated_inst_motion)
        inst_motion_gt = updated_inst_motion.clone()
        K, T, _, _ = updated_inst_motion.size()
        if padding_indice is not None:
            padded_time_indice = torch.cat((time_indice[:, 1], torch.zeros_like(padding_indice).long()))
            padding_indice = torch.cat((torch.arange(n_points, device=device).long(), padding_indice))
        else:
            padding_indice = torch.arange(n_points, device=device).long()
            padded_time_indice = time_indice[:, 1]

        padded_inst_labels = inst_labels[padding_indice]
        padded_points = transformed_points[padding_indice]
        tpointnet_input = {'frame_feats': backbone_feats[padding_indice], 'time_indice': padded_time_indice,
                           'inst_labels': padded_inst_labels, 'mos_labels': mos_labels[padding_indice],
                           'mos_feats': mos_feats[padding_indice]}
        results['tpointnet_loss_terms'] = dict()
        final_pose_est = None
        for idx in range(self.n_iterations):
            tpointnet_input['points'] = padded_points.detach()
            tpointnet_input['inst_motion_gt'] = updated_inst_motion.detach()
            predictions = self.alignment(tpointnet_input)
            results['tpointnet_loss_terms'][f'{idx}_th'] = predictions
            c_pose = predictions['inst_est_motion']
            padded_points = reconstruct_sequence(padded_points, padded_time_indice, padded_inst_labels, c_pose, T)
            # re-express the remaining GT motion after this iteration's estimate (alignnet.py:259-263)
            updated_inst_motion = updated_inst_motion.view(-1, 4, 4)
            c_pose = c_pose.view(-1, 4, 4)
            updated_inst_motion[:, :3, :3] = torch.matmul(updated_inst_motion[:, :3, :3], c_pose[:, :3, :3].transpose(1, 2))
            updated_inst_motion[:, :3, 3] = updated_inst_motion[:, :3, 3] - torch.matmul(
                updated_inst_motion[:, :3, :3], c_pose[:, :3, 3].unsqueeze(-1)).squeeze(-1)
            updated_inst_motion = updated_inst_motion.view(K, T, 4, 4)
            final_pose_est = c_pose if final_pose_est is None else torch.matmul(c_pose, final_pose_est)

        final_pose_est = final_pose_est.view(K, T, 4, 4)
        rec_est = reconstruct_sequence(input_dict['transformed_points'], time_indice[:, 1], inst_labels, final_pose_est, T)
        rec_gt = reconstruct_sequence(input_dict['transformed_points'], time_indice[:, 1], inst_labels, inst_motion_gt, T)
        l2_error = torch.norm(rec_est - rec_gt, p=2, dim=1)
        weights = time_indice[:, 1] > 0
        weights_mos = (input_dict['mos_labels'] == 1) & (time_indice[:, 1] > 0)
        # 0-d tensors; MotionNet.forward converts them to floats in its single end-of-forward sync (alignnet.py:280-281)
        results['inst_l2_error'] = (l2_error * weights).sum() / (weights.sum() + _EPS)
        results['dynamic_inst_l2_error'] = (l2_error * weights_mos).sum() / (weights_mos.sum() + _EPS)
        results['inst_labels_adjusted'] = inst_labels
        results['inst_pose_est'] = final_pose_est
        results['sub_rec_est'] = rec_est
        return results
