"""Per-instance rigid-motion regressor ("TubeNet"): host mirror of models/tpointnet.py.

Stays plain PyTorch-ROCm (0.6 % of the FLOPs); the per-instance poolings use
ops.scatter (torch scatter_reduce) on K*T-row outputs.  state_dict keys: `alignment.{geo_embed,motion_embed,
pos_embed,regressor}.*` under `reconstructor` (appendix B).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .chamfer_distance import ChamferDistance
from . import ops
from .ops import scatter, ScatterPlan, linear_rows, transform_by_index, point_dtype, tube_rows, tube_code, tube_pose

_EPS = 1e-20


_CONST = {}


def _const(name, device, build):
    """Small constant tables (built once per device) that turn the per-entry formulas below into one gather or one product."""
    key = (name, str(device))
    if key not in _CONST:
        _CONST[key] = build().to(device)
    return _CONST[key]


def _quat_products_to_pose():
    """[20,16] table: row = product q_a*q_b (a*4+b, order x,y,z,w), then the translation (3) and a one; column = entry of the
    row-major 4x4 pose. Rotation block as toolbox/se3_utils.py:44-64, translation in the last column, last row (0,0,0,1)."""
    x, y, z, w = 0, 1, 2, 3
    rot = {(0, 0): [(w, w, 1), (x, x, 1), (y, y, -1), (z, z, -1)], (0, 1): [(x, y, 2), (w, z, -2)], (0, 2): [(w, y, 2), (x, z, 2)],
           (1, 0): [(w, z, 2), (x, y, 2)], (1, 1): [(w, w, 1), (x, x, -1), (y, y, 1), (z, z, -1)], (1, 2): [(y, z, 2), (w, x, -2)],
           (2, 0): [(x, z, 2), (w, y, -2)], (2, 1): [(w, x, 2), (y, z, 2)], (2, 2): [(w, w, 1), (x, x, -1), (y, y, -1), (z, z, 1)]}
    table = torch.zeros(20, 16)
    for (r, c), terms in rot.items():
        for a, b, coef in terms:
            table[a * 4 + b, r * 4 + c] += coef
    for r in range(3):
        table[16 + r, r * 4 + 3] = 1
    table[19, 15] = 1
    return table


def quat2mat(quat):
    """toolbox/se3_utils.py:44-64: [x,y,z,w] quaternion (scipy order) -> [B,3,3]. The nine entries are sums of products
    q_a*q_b: one outer product and one [B,16]x[16,9] product instead of ~40 element-wise launches."""
    table = _const('q2m', quat.device, _quat_products_to_pose)[:16].reshape(16, 4, 4)[:, :3, :3].reshape(16, 9)
    outer = (quat[:, :, None] * quat[:, None, :]).reshape(-1, 16)
    return (outer @ table.to(quat.dtype)).reshape(-1, 3, 3)


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
    """models/tpointnet.py:20-40: [N,7] (quat xyzw, trans) -> [N,4,4]; the quaternion is normalised first. Rotation, translation
    and the constant row come out of one product with the table above."""
    quat = F.normalize(pose_est_rep[:, :4], p=2, dim=1)
    n = pose_est_rep.size(0)
    outer = (quat[:, :, None] * quat[:, None, :]).reshape(n, 16)
    terms = torch.cat((outer, pose_est_rep[:, 4:], torch.ones_like(pose_est_rep[:, :1])), dim=1)
    return (terms @ _const('q2m', pose_est_rep.device, _quat_products_to_pose).to(terms.dtype)).reshape(n, 4, 4)


def _mat2quat_tables():
    """Index tables for mat2quat: candidate c (0..2: component c formed first, 3: w first), component e -> the two entries of
    the row-major 3x3 matrix that are added (sign +1) or subtracted (sign -1); the component formed first is marked instead."""
    first, second, sign, lead = torch.zeros(16, dtype=torch.long), torch.zeros(16, dtype=torch.long), torch.zeros(16), torch.zeros(16, dtype=torch.long)
    is_lead = torch.zeros(16, dtype=torch.bool)
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        is_lead[i * 4 + i] = True; lead[i * 4 + i] = i
        for e, (a, b, sg) in ((j, (j * 3 + i, i * 3 + j, 1.0)), (k, (k * 3 + i, i * 3 + k, 1.0)), (3, (k * 3 + j, j * 3 + k, -1.0))):
            first[i * 4 + e], second[i * 4 + e], sign[i * 4 + e] = a, b, sg
    for e, (a, b) in enumerate(((2 * 3 + 1, 1 * 3 + 2), (0 * 3 + 2, 2 * 3 + 0), (1 * 3 + 0, 0 * 3 + 1))):
        first[12 + e], second[12 + e], sign[12 + e] = a, b, -1.0
    is_lead[15] = True; lead[15] = 3
    return first, second, sign.double(), lead, is_lead


def mat2quat(rot):
    """[N,3,3] rotation matrices -> [N,4] quaternions (x, y, z, w) in float64, on the tensor's device: the branch scheme of
    scipy.spatial.transform.Rotation.from_matrix(...).as_quat() (largest of the three diagonal entries and the trace decides
    which component is formed first; ties go to the first), which the reference calls on the host (models/tpointnet.py:62-66).
    Identical to scipy 1.15 to the last bit on proper rotations and to 2e-8 on float32-rounded ones, which scipy
    re-orthogonalises first (tests/test_host_logic.py); no device -> host round trip."""
    dev = rot.device
    first = _const('m2q_first', dev, lambda: _mat2quat_tables()[0])
    second = _const('m2q_second', dev, lambda: _mat2quat_tables()[1])
    sign = _const('m2q_sign', dev, lambda: _mat2quat_tables()[2])
    lead = _const('m2q_lead', dev, lambda: _mat2quat_tables()[3])
    is_lead = _const('m2q_is_lead', dev, lambda: _mat2quat_tables()[4])
    m = rot.double().reshape(-1, 9)
    diag = m[:, 0::4]
    tr = (m[:, 0] + m[:, 4] + m[:, 8])[:, None]
    leading = torch.cat((1 - tr + 2 * diag, 1 + tr), dim=1)                         # the component formed first, per candidate
    cands = torch.where(is_lead, leading[:, lead], m[:, first] + sign * m[:, second]).reshape(-1, 4, 4)
    choice = torch.cat((diag, tr), dim=1).argmax(dim=1)
    quat = cands.gather(1, choice[:, None, None].expand(-1, 1, 4))[:, 0]
    return quat / torch.norm(quat, dim=1, keepdim=True)


def batch_mat2quat(pose_gt, centroids):
    """models/tpointnet.py:43-73: ground-truth poses re-expressed for centred clouds, plus their 7-vector (float64, as
    to_tensor(np.array(...)) yields in the reference)."""
    n_frames = pose_gt.size(1)
    device = pose_gt.device
    tsfm = pose_gt.clone().view(-1, 4, 4)
    cen = centroids.repeat_interleave(n_frames, 0).unsqueeze(2)
    B = tsfm.size(0)
    tsfm[:, :3, 3] += torch.matmul(tsfm[:, :3, :3] - torch.eye(3, device=device)[None].repeat(B, 1, 1), cen).squeeze(2)
    det = tsfm.detach()
    rep = torch.cat((mat2quat(det[:, :3, :3]), det[:, :3, 3].double()), dim=1)
    return tsfm, rep


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
    pd = point_dtype() if x.is_cuda else x.dtype                  # bf16 rows in the bf16 compute mode (GPU only)
    if pd != x.dtype and mods[0].in_features >= 32:
        x = x.to(pd)                                               # wide first layer: take the matrix-core path from the start
    # 'mixed' mode: the chain runs as bf16 shadows of fp32 rows (fp32x3 forward, bf16 backward): a wide fp32 input enters the shadow graph,
    # a narrow one (the 4-feature positional rows, no gradient) is taken as it is by the first layer
    head = False
    n_lin = sum(isinstance(m, nn.Linear) for m in mods)
    if ops.mixed_mode() and x.is_cuda and x.dtype == torch.float32 and x.shape[0] >= ops.MIN_ROWS_FUSED_LINEAR and n_lin > 1:
        if mods[0].in_features >= 32:
            x = ops.enter_mixed(x.contiguous())
        else:
            head = True
    k = 0
    while i < len(mods):
        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        k += 1
        if k == n_lin:
            # the LAST layer leaves the bf16 graph: the pooled codes feed Linear -> BatchNorm1d (the regressor), whose backward hands back a gradient
            # that sums to zero over the instances -- this layer's bias gradient is that sum, exact in fp32 (1e-7) and 1e-3 from bf16-rounded rows
            x = ops.exit_mixed(x)
        x = linear_rows(x, mods[i], post_relu=relu, out_dtype=pd, mixed=head and i == 0)
        i += 2 if relu else 1
    return x


class TPointNet(BaseModel):
    """Per-(instance, frame) pose regressor, models/tpointnet.py:167-305, organised here as four steps:
    frame weights -> pooled embeddings -> pose regression -> losses / un-centring."""

    def __init__(self, config):
        BaseModel.__init__(self, config)
        self.geo_embed = _mlp([32, 32, 64, 128])
        self.motion_embed = _mlp([64, 64, 128, 128])
        self.pos_embed = _mlp([4, 32, 64, 128])
        self.regressor = nn.Sequential(nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(),
                                       nn.Linear(256, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Linear(128, 7))
        self.min_points_per_frame = config['tpointnet']['min_points']

    # -- step 1 ---------------------------------------------------------------------------------------------
    def _frame_weights(self, slot, per_slot, moving, n_inst, n_slots, device):
        """Weight of every (instance, frame) slot: populated enough, moving (static slots get 0.2 -- assigned into an
        integer tensor in the reference, models/tpointnet.py:231-233, i.e. 0; kept), later frames count more."""
        ones = torch.ones(slot.size(0), device=device)
        population = scatter(ones, slot, dim=0, dim_size=n_slots, reduce='sum', plan=per_slot)
        enough = (population > self.min_points_per_frame).float()
        slot_label = scatter(moving, slot, dim=0, dim_size=n_slots, reduce='max', plan=per_slot)
        label_w = torch.ones_like(slot_label)
        label_w[slot_label == 0] = 0.2
        ramp = (torch.arange(self.n_frames, device=device) + 1).repeat(n_inst) / self.n_frames
        return enough * label_w * ramp

    # -- step 2 ---------------------------------------------------------------------------------------------
    def shared_terms(self, input_dict):
        """Everything of forward() that the refinement iterations of AlignNet have in common: the reference recomputes it in every
        iteration (models/alignnet.py:236-247 hands the same labels and features to models/tpointnet.py:200-262 each time), but
        instance / frame indices, the slot weights and the pooled motion and geometry codes depend only on labels, features and
        weights -- none of which moves between iterations.  Computed once; the gradients of all iterations meet in one graph."""
        feats_motion, feats_geo = input_dict['mos_feats'], input_dict['frame_feats']
        t_idx, inst = input_dict['time_indice'], input_dict['inst_labels']
        n_inst, T = input_dict['inst_motion_gt'].size(0), input_dict['inst_motion_gt'].size(1)
        n_slots = n_inst * T
        slot = (inst * T + t_idx).long()                                   # flat (instance, frame) index of every point
        per_slot = ScatterPlan(slot, n_slots)                              # one CSR per index vector, shared below
        per_inst = ScatterPlan(inst, n_inst)
        weights = self._frame_weights(slot, per_slot, input_dict['mos_labels'], n_inst, n_slots, feats_motion.device)
        e_motion = scatter(_embed(self.motion_embed, feats_motion), inst, dim=0, dim_size=n_inst, reduce='max', plan=per_inst)
        e_geo = scatter(_embed(self.geo_embed, feats_geo), inst, dim=0, dim_size=n_inst, reduce='max', plan=per_inst)
        return {'slot': slot, 'per_slot': per_slot, 'per_inst': per_inst, 'weights': weights,
                'e_motion': e_motion.float(), 'e_geo': e_geo.float()}     # pooled codes: [K,128], fp32 from here

    def _frame_embedding(self, xyz, slot, per_slot, n_slots, T):
        slot_centre = scatter(xyz, slot, dim=0, dim_size=n_slots, reduce='mean', plan=per_slot)      # row k*T: the anchor frame's centroid
        rows = tube_rows(xyz, per_slot, slot_centre, T)                                              # (xyz - anchor centre, t / T)
        e_frame = scatter(_embed(self.pos_embed, rows), slot, dim=0, dim_size=n_slots, reduce='max', plan=per_slot)
        return e_frame.float(), slot_centre, rows                           # [K*T,128]

    def forward(self, input_dict, shared=None):
        """models/tpointnet.py:200-305.  Beyond the reference's keys the result carries the two running pose tables of AlignNet's loop
        ('remaining' = inst_motion_gt @ step^-1, 'total' = step @ input_dict['total'], models/alignnet.py:257-263): the slot kernel
        that builds the poses updates them on the way."""
        xyz = input_dict['points']
        gt_motion = input_dict['inst_motion_gt']
        n_inst, T = gt_motion.size(0), gt_motion.size(1)
        n_slots = n_inst * T
        if shared is None:
            shared = self.shared_terms(input_dict)
        slot, per_slot, weights = shared['slot'], shared['per_slot'], shared['weights']
        e_frame, slot_centre, rows = self._frame_embedding(xyz, slot, per_slot, n_slots, T)

        # step 3: one 7-vector (quaternion xyzw + translation) per slot from [geometry | motion | frame | anchor frame]
        pose_vec = self.regressor(tube_code(shared['e_geo'], shared['e_motion'], e_frame, T))

        # step 4: poses, losses on the centred clouds (the reference's l1 / l2 names are swapped, tpointnet.py:281-282; kept),
        # un-centring (t += (I - R) c) with frame 0 pinned to the identity (tpointnet.py:291-296)
        l1_loss, l2_loss, rot_loss, trans_loss, step, remaining, total = tube_pose(
            pose_vec, rows, per_slot, gt_motion, input_dict.get('total'), slot_centre, weights, T)
        shape = (n_inst, T, 4, 4)
        return {'l1_loss': l1_loss, 'l2_loss': l2_loss, 'rot_loss': rot_loss, 'trans_loss': trans_loss,
                'inst_est_motion': step.view(shape), 'remaining': remaining.view(shape), 'total': total.view(shape)}
