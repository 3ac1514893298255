"""Per-instance rigid-motion regressor ("TubeNet"): host mirror of models/tpointnet.py.

Stays plain PyTorch-ROCm (0.6 % of the FLOPs); the per-instance poolings use
ops.scatter (torch scatter_reduce) on K*T-row outputs.  state_dict keys: `alignment.{geo_embed,motion_embed,
pos_embed,regressor}.*` under `reconstructor` (appendix B).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.spatial.transform import Rotation as R

from .chamfer_distance import ChamferDistance
from .ops import scatter, ScatterPlan, linear_rows, transform_by_index

_EPS = 1e-20


def quat2mat(quat):
    """toolbox/se3_utils.py:44-64: [x,y,z,w] quaternion (scipy order) -> [B,3,3]."""
    x, y, z, w = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    B = quat.size(0)
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).reshape(B, 3, 3)


def apply_tsfm(src, tsfm):
    """toolbox/register_utils.py:199-206."""
    return (tsfm[:3, :3] @ src.T + tsfm[:3, 3][:, None]).T


def reconstruct_sequence(points, time_indice, inst_labels, tsfm, n_frames):
    """toolbox/register_utils.py:73-93: apply tsfm[inst, t] to every point."""
    assert n_frames == tsfm.size(1)
    idx = (inst_labels.long() * n_frames + time_indice).long()
    return transform_by_index(points, idx, tsfm.reshape(-1, 4, 4))


def ego_motion_compensation(points, time_indice, tsfm):
    """toolbox/register_utils.py:59-70."""
    return transform_by_index(points, time_indice.long(), tsfm)


def batch_quat2mat(pose_est_rep):
    """models/tpointnet.py:20-40: [N,7] (quat xyzw, trans) -> [N,4,4]; the quaternion is normalised first."""
    quat = F.normalize(pose_est_rep[:, :4], p=2, dim=1)
    out = torch.eye(4, device=pose_est_rep.device)[None].repeat(pose_est_rep.size(0), 1, 1)
    out[:, :3, :3] = quat2mat(quat)
    out[:, :3, 3] = pose_est_rep[:, 4:]
    return out


def batch_mat2quat(pose_gt, centroids):
    """models/tpointnet.py:43-73: ground-truth poses re-expressed for centred clouds, plus their 7-vector
    (scipy as_quat on the host, as in the reference)."""
    n_frames = pose_gt.size(1)
    device = pose_gt.device
    tsfm = pose_gt.clone().view(-1, 4, 4)
    cen = centroids.repeat_interleave(n_frames, 0).unsqueeze(2)
    B = tsfm.size(0)
    tsfm[:, :3, 3] += torch.matmul(tsfm[:, :3, :3] - torch.eye(3, device=device)[None].repeat(B, 1, 1), cen).squeeze(2)
    host = tsfm.detach().cpu().numpy()
    rep = np.concatenate([R.from_matrix(host[:, :3, :3]).as_quat(), host[:, :3, 3]], axis=1)
    return tsfm, torch.from_numpy(rep).to(device)        # float64, as to_tensor(np.array(...)) yields in the reference


def evaluate_pose(pose_est_rep, pose_gt_rep, weights):
    """models/tpointnet.py:76-94."""
    quat_est = F.normalize(pose_est_rep[:, :4], p=2, dim=1)
    dq = pose_gt_rep[:, :4] - quat_est
    dt = pose_gt_rep[:, 4:] - pose_est_rep[:, 4:]
    rot_loss = (torch.norm(dq, p=2, dim=1) * weights).sum() / (weights.sum() + _EPS)
    trans_loss = (torch.norm(dt, p=2, dim=1) * weights).sum() / (weights.sum() + _EPS)
    return rot_loss, trans_loss


class BaseModel(nn.Module):
    """models/tpointnet.py:97-163: alignment-error helpers around the Chamfer module."""

    def __init__(self, config):
        super().__init__()
        self.n_frames = config['voxel_generator']['n_sweeps']
        self.chamfer_dist = ChamferDistance()

    def align_frames(self, points, time_indice, poses):
        points = points.clone()
        for idx in range(self.n_frames):
            sel = time_indice == idx
            if sel.sum():
                points[sel] = apply_tsfm(points[sel], poses[idx])
        return points

    def get_chamfer_distance(self, est_points, gt_points, weights):
        dist1, dist2 = self.chamfer_dist(gt_points[None], est_points[None])
        return ((dist1 * weights).sum() + (dist2 * weights).sum()) / 2

    def get_l2_distance(self, est_points, gt_points, weights):
        return (torch.norm(est_points - gt_points, dim=1) * weights).sum()

    def get_alignment_errors(self, points, time_indice, est_poses, gt_poses):
        est_points = self.align_frames(points, time_indice, est_poses)
        gt_points = self.align_frames(points, time_indice, gt_poses)
        weights = torch.zeros(est_points.size(0), device=est_points.device)
        weights[time_indice == 1] = 1.0
        weights = weights / (weights.sum() + _EPS)
        return (self.get_chamfer_distance(est_points, gt_points, weights),
                self.get_l2_distance(est_points, gt_points, weights))


def _mlp(dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1], bias=True))
        if i < len(dims) - 2:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


def _embed(seq, x):
    """(Linear, ReLU)*, Linear with the fused row-linear kernels."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        x = linear_rows(x, mods[i], post_relu=relu)
        i += 2 if relu else 1
    return x


class TPointNet(BaseModel):
    """models/tpointnet.py:167-305."""

    def __init__(self, config):
        BaseModel.__init__(self, config)
        self.geo_embed = _mlp([32, 32, 64, 128])
        self.motion_embed = _mlp([64, 64, 128, 128])
        self.pos_embed = _mlp([4, 32, 64, 128])
        self.regressor = nn.Sequential(nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(),
                                       nn.Linear(256, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Linear(128, 7))
        self.min_points_per_frame = config['tpointnet']['min_points']

    def forward(self, input_dict):
        mos_feat, frame_feats = input_dict['mos_feats'], input_dict['frame_feats']
        points = input_dict['points']
        time_indice, inst_indice = input_dict['time_indice'], input_dict['inst_labels']
        inst_motion_gt = input_dict['inst_motion_gt']
        mos_labels = input_dict['mos_labels']
        K, T, _, _ = inst_motion_gt.size()
        device = mos_feat.device
        frame_indice = (inst_indice * T + time_indice).long()
        per_frame = ScatterPlan(frame_indice, K * T)        # one CSR per index vector, shared by the poolings below
        per_inst = ScatterPlan(inst_indice, K)

        # 1. per (instance, frame) weights: enough points, moving, later frames count more (tpointnet.py:223-237)
        count = torch.ones(frame_indice.size(0), device=device)
        frame_count = scatter(count, frame_indice, dim=0, dim_size=K * T, reduce='sum', plan=per_frame)
        frame_weights = (frame_count > self.min_points_per_frame).float()
        inst_mos_label = scatter(mos_labels, frame_indice, dim=0, dim_size=K * T, reduce='max', plan=per_frame)
        mos_weights = torch.ones_like(inst_mos_label)
        mos_weights[inst_mos_label == 0] = 0.2
        temporal_weights = (torch.arange(self.n_frames) + 1).to(device).repeat(K) / self.n_frames
        frame_weights = frame_weights * mos_weights * temporal_weights

        # 2. pooled embeddings (tpointnet.py:240-262)
        mos_embedding = scatter(_embed(self.motion_embed, mos_feat), inst_indice, dim=0, dim_size=K, reduce='max', plan=per_inst)
        geo_embedding = scatter(_embed(self.geo_embed, frame_feats), inst_indice, dim=0, dim_size=K, reduce='max', plan=per_inst)
        frame_centroid = scatter(points, frame_indice, dim=0, dim_size=K * T, reduce='mean', plan=per_frame)
        inst_centroid = frame_centroid[::T]
        centered_points = points - inst_centroid[inst_indice]
        frame_input = torch.cat((centered_points, time_indice.unsqueeze(-1) / T), dim=1).float()
        frame_embedding = scatter(_embed(self.pos_embed, frame_input), frame_indice, dim=0, dim_size=K * T, reduce='max', plan=per_frame)

        # 3. regress one pose per (instance, frame) (tpointnet.py:264-273)
        anchor_embedding = frame_embedding[::T].repeat_interleave(T, 0)
        regressor_input = torch.cat((geo_embedding.repeat_interleave(T, 0), mos_embedding.repeat_interleave(T, 0),
                                     frame_embedding, anchor_embedding), dim=1)
        pose_est_rep = self.regressor(regressor_input)
        pose_est_tsfm = batch_quat2mat(pose_est_rep)

        # 4. losses (tpointnet.py:275-289; the names l1/l2 are swapped in the reference and kept so)
        pose_gt_tsfm, pose_gt_rep = batch_mat2quat(inst_motion_gt, inst_centroid)
        rec_est = reconstruct_sequence(centered_points, time_indice, inst_indice, pose_est_tsfm.view(K, T, 4, 4), T)
        rec_gt = reconstruct_sequence(centered_points, time_indice, inst_indice, pose_gt_tsfm.view(K, T, 4, 4), T)
        diff = rec_est - rec_gt
        l1_loss = torch.norm(diff, p=2, dim=1)
        l2_loss = torch.norm(diff, p=1, dim=1)
        frame_l1 = scatter(l1_loss, frame_indice, dim=0, dim_size=K * T, reduce='mean', plan=per_frame)
        frame_l2 = scatter(l2_loss, frame_indice, dim=0, dim_size=K * T, reduce='mean', plan=per_frame)
        l1_loss = (frame_l1 * frame_weights).sum() / (frame_weights.sum() + _EPS)
        l2_loss = (frame_l2 * frame_weights).sum() / (frame_weights.sum() + _EPS)
        rot_loss, trans_loss = evaluate_pose(pose_est_rep, pose_gt_rep, frame_weights)

        # 5. undo the centring; frame 0 is the identity (tpointnet.py:291-296)
        cen = inst_centroid.repeat_interleave(T, 0).unsqueeze(-1)
        pose_est_tsfm[:, :3, 3] += torch.matmul(torch.eye(3, device=device)[None].repeat(K * T, 1, 1) - pose_est_tsfm[:, :3, :3], cen).squeeze(2)
        pose_est_tsfm = pose_est_tsfm.view(K, T, 4, 4)
        pose_est_tsfm[:, 0] = torch.eye(4, device=device)[None].repeat(K, 1, 1)
        return {'l1_loss': l1_loss, 'l2_loss': l2_loss, 'rot_loss': rot_loss, 'trans_loss': trans_loss,
                'inst_est_motion': pose_est_tsfm}
