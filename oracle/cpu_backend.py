"""CPU stand-in for pcaccumulation_amd.native, built on the oracle -- TEST INFRASTRUCTURE.

Same signatures as pcaccumulation_amd/native.py, CPU torch tensors in and out.  Two users, both allowed to
touch oracle/: (1) the CPU-only test suite monkeypatches these functions over the ctypes bindings to exercise
the host logic of the product (autograd wrappers, module wiring, layouts, result keys) in a container without
a GPU; (2) bench.py's `cpu_baseline` leg times the whole path on the host cores with them.  The product never
imports this module and has no switch that selects it: `install()` is an explicit patch applied by the caller.
"""
import numpy as np
import torch

import oracle
from oracle import twin


def _np(t):
    return t.detach().cpu().numpy()


def voxelize(points, voxel_size, pc_range, grid, nt, max_voxels):
    out = oracle.voxelize(_np(points), voxel_size, pc_range, nt, max_voxels=max_voxels)
    m = int(out['num_voxels'][0])
    coords = torch.zeros((max_voxels, 4), dtype=torch.int32)
    coords[:m] = torch.from_numpy(out['coordinates'])
    return coords, torch.from_numpy(out['point_to_voxel_map'][:, 0].copy()), torch.tensor([m], dtype=torch.int32)


def cell_index(coords, nx, ny, nt, n_batch):
    cell, c2p = twin.cell_index(_np(coords), nx, ny, nt, n_batch)
    return torch.from_numpy(cell), torch.from_numpy(c2p)


def frame_pillars(cell2pillar, cells_per_frame, m):
    c2p = _np(cell2pillar)
    occ = c2p >= 0
    offs = np.concatenate([[0], np.cumsum(occ.reshape(-1, cells_per_frame).sum(1))]).astype(np.int32)
    sp = np.zeros(m, np.int32)                                 # like the HIP binding: [m], entries beyond the occupied-cell count are 0
    sp[:int(occ.sum())] = c2p[occ]
    return torch.from_numpy(sp), torch.from_numpy(offs)


def csr_build(p2v, m):
    offs, order = twin.csr_build(_np(p2v), m)
    return torch.from_numpy(offs), torch.from_numpy(order)


def _p2v_from_csr(offs, order):
    o, r = _np(offs), _np(order)
    seg = np.repeat(np.arange(o.shape[0] - 1), np.diff(o))
    p2v = np.empty(r.shape[0], np.int64)
    p2v[r] = seg
    return p2v


def segment_mean3_maxlabel(points, labels, offs, order, m):
    mean, lab = twin.segment_mean3_maxlabel(_np(points), _np(labels) if labels is not None else None, _np(offs), _np(order), m)
    return torch.from_numpy(mean), (torch.from_numpy(lab) if lab is not None else None)


def segment_max(src, offs, order, m):
    out, arg = twin.segment_max(_np(src.float()), _np(offs), _np(order), m)
    return torch.from_numpy(out).to(src.dtype), torch.from_numpy(arg)


def segment_max_backward(grad_out, arg, p2v, n, out_dtype=None):
    out = torch.from_numpy(twin.segment_max_backward(_np(grad_out.float()), _np(arg), _np(p2v), n))
    return out.to(out_dtype) if out_dtype is not None else out.to(grad_out.dtype)


def segment_sum(src, offs, order, m):
    return torch.from_numpy(twin.segment_sum(_np(src.float()), _np(offs), _np(order), m))


def pillar_scatter(feats, cell2pillar, out_dtype=torch.float32):
    return torch.from_numpy(twin.pillar_scatter(_np(feats.float()), _np(cell2pillar))).to(out_dtype)


def gather_rows(src, idx):
    return torch.from_numpy(twin.gather_rows(_np(src), _np(idx)))


def bilinear_gather(fmap, points, map_idx, x_scale, y_scale):
    return torch.from_numpy(twin.bilinear_gather(_np(fmap.float()), _np(points), _np(map_idx), x_scale, y_scale))


def bilinear_gather_backward(grad_out, shape, points, map_idx, x_scale, y_scale):
    n, h, w, c = shape
    with torch.enable_grad():                       # called from inside an autograd backward
        f = torch.zeros((n, c, h, w), requires_grad=True)
        tot = 0
        for b in range(n):
            sel = map_idx == b
            if sel.any():
                grid = torch.stack([points[sel, 0] / x_scale, points[sel, 1] / y_scale], 1).view(1, -1, 1, 2)
                s = torch.nn.functional.grid_sample(f[b:b + 1], grid, mode='bilinear', padding_mode='border', align_corners=False)
                tot = tot + (s[0, :, :, 0].T * grad_out[sel].detach()).sum()
        if torch.is_tensor(tot):
            tot.backward()
            return f.grad.permute(0, 2, 3, 1).contiguous()
    return torch.zeros(shape)


def bev_warp(bev, inv_pose, x_reso, y_reso, x_min, y_min):
    return torch.from_numpy(twin.bev_warp(_np(bev.float()), _np(inv_pose), x_reso, y_reso, x_min, y_min)).to(bev.dtype)


def rigid_transform(points, frame_idx, tsfm):
    return torch.from_numpy(twin.rigid_transform(_np(points), _np(frame_idx), _np(tsfm)))


def chamfer_forward(xyz1, xyz2):
    return tuple(torch.from_numpy(a) for a in oracle.chamfer_forward(_np(xyz1), _np(xyz2)))


def chamfer_backward(xyz1, xyz2, gd1, gd2, i1, i2):
    return tuple(torch.from_numpy(a) for a in oracle.chamfer_backward(_np(xyz1), _np(xyz2), _np(gd1), _np(gd2), _np(i1), _np(i2)))


def rows_linear_supported(k, n):
    return k in (2, 3, 4, 9, 32, 64, 128) and 1 <= n <= 128


def rows_linear(x, w, bias=None, residual=None, pre_relu=False, post_relu=False, in_mask=None, out_mask=None, out_dtype=None):
    out_dtype = out_dtype or x.dtype
    h = torch.relu(x.float()) if pre_relu else x.float()
    if in_mask is not None:
        h = h * (in_mask > 0)
    y = torch.nn.functional.linear(h, w, bias)
    if residual is not None:
        y = y + residual.float()
    if post_relu:
        y = torch.relu(y)
    if out_mask is not None:
        y = y * (out_mask > 0)
    return y.to(out_dtype)


def rows_wgrad(dy, x, dy_mask=None, x_relu=False, split=False):
    g = dy.float() * (dy_mask > 0) if dy_mask is not None else dy.float()
    h = torch.relu(x.float()) if x_relu else x.float()
    aug = torch.cat([h, torch.ones(h.shape[0], 1)], dim=1)
    out = g.t() @ aug
    return (out[:, :-1].contiguous(), out[:, -1].contiguous()) if split else out


def pfn_features(points, p2v, pillar_mean, coords, time_indice, vx, vy, x_offset, y_offset, scale, n_frames):
    return torch.from_numpy(twin.pfn_features(_np(points), _np(p2v), _np(pillar_mean), _np(coords), _np(time_indice), vx, vy, x_offset,
                                              y_offset, scale, n_frames))


def scatter_sum_small(src, idx, m):
    return torch.zeros((m, src.shape[1]), dtype=torch.float32).index_add_(0, idx.long(), src.float())


def sinkhorn_kabsch(feats_s, feats_t, coor_s, coor_t, thr2, params, n_iters):
    P, k, _ = feats_s.shape
    perms, poses = [], []
    sp, den = float(params[0]), float(params[1])
    for p in range(P):
        f32 = np.float32
        fs, ft, cs, ct = (_np(t[p]).astype(f32) for t in (feats_s, feats_t, coor_s, coor_t))
        support = (oracle.square_distance(cs, ct) < f32(thr2[p])).astype(f32)
        aff = -(oracle.square_distance(fs, ft, normalised=True) - f32(sp)) / f32(den)
        perm = np.exp(oracle.sinkhorn(aff, n_iters)) * support
        rowsum = perm.sum(1, keepdims=True)
        r, t = oracle.kabsch(cs, (perm @ ct) / (rowsum + f32(1e-20)), rowsum[:, 0])
        pose = np.eye(4, dtype=f32)
        pose[:3, :3] = r
        pose[:3, 3] = t[:, 0]
        perms.append(perm)
        poses.append(pose)
    return torch.from_numpy(np.stack(perms)), torch.from_numpy(np.stack(poses))


def cluster(points, offset, sel, batch, n_batches, voxel_size, eps, min_samples, min_p_cluster):
    use_offset = offset is not None
    assert abs(voxel_size - (0.05 if use_offset else 0.15)) < 1e-6
    off = _np(offset) if use_offset else np.zeros((points.shape[0], 2), np.float32)
    ti = np.stack([_np(batch).astype(np.int64), np.zeros(points.shape[0], np.int64)], 1)
    lab = np.zeros(points.shape[0], np.int64)
    if points.shape[0]:
        lab = oracle.cluster_forward(_np(points), _np(sel).astype(np.int64), off, ti, eps, min_samples, min_p_cluster, use_offset)
    return torch.from_numpy(lab)


# ---- TubeNet slot algebra: torch restatement of models/tpointnet.py:249-305 and models/alignnet.py:257-263 -------------------------
def _tube_slot_terms(pose_vec, remaining, slot_centre, T):
    """(pose_c [S,4,4], gt_c [S,4,4], gt 7-vector [S,7] f64, anchor centres per slot [S,3]) as the reference builds them."""
    from pcaccumulation_amd.tpointnet import batch_quat2mat, batch_mat2quat
    anchor = slot_centre.view(-1, T, 3)[:, 0]
    pose_c = batch_quat2mat(pose_vec)                                        # models/tpointnet.py:265
    gt_c, gt_vec = batch_mat2quat(remaining.view(-1, T, 4, 4), anchor)       # models/tpointnet.py:268
    return pose_c, gt_c, gt_vec, anchor.repeat_interleave(T, 0)


def tube_rows(xyz, slot, slot_centre, n_frames):
    s = slot.long()
    anchor = slot_centre[s - s % n_frames]
    return torch.cat((xyz - anchor, ((s % n_frames).unsqueeze(-1) / n_frames).float()), dim=1)   # models/tpointnet.py:246-250


def tube_code(geo, motion, frame, n_frames):
    T = n_frames                                                             # models/tpointnet.py:259-262
    return torch.cat((geo.repeat_interleave(T, 0), motion.repeat_interleave(T, 0), frame, frame[::T].repeat_interleave(T, 0)), dim=1)


def tube_code_backward(grad_code, n_inst, n_frames, c):
    g = grad_code.view(n_inst, n_frames, 4, c)
    g_frame = g[:, :, 2].clone()
    g_frame[:, 0] += g[:, :, 3].sum(1)
    return g[:, :, 0].sum(1), g[:, :, 1].sum(1), g_frame.reshape(-1, c)


def tube_pose_forward(pose_vec, remaining, total, slot_centre, weights, n_frames):
    from pcaccumulation_amd.tpointnet import evaluate_pose
    T, S = n_frames, pose_vec.shape[0]
    pose_c, gt_c, gt_vec, centre = _tube_slot_terms(pose_vec.detach(), remaining, slot_centre, T)
    rot, trans = evaluate_pose(pose_vec.detach(), gt_vec, weights)            # models/tpointnet.py:288
    step = pose_c.clone()                                                     # models/tpointnet.py:291-296
    step[:, :3, 3] += torch.matmul(torch.eye(3)[None] - step[:, :3, :3], centre.unsqueeze(-1)).squeeze(2)
    step.view(-1, T, 4, 4)[:, 0] = torch.eye(4)
    rem = remaining.clone().view(-1, 4, 4)                                    # models/alignnet.py:259-263
    rem[:, :3, :3] = torch.matmul(rem[:, :3, :3], step[:, :3, :3].transpose(1, 2))
    rem[:, :3, 3] = rem[:, :3, 3] - torch.matmul(rem[:, :3, :3], step[:, :3, 3].unsqueeze(-1)).squeeze(-1)
    total_out = step.clone() if total is None else torch.matmul(step, total.view(-1, 4, 4))
    flat = lambda m: torch.cat((m[:, :3, :3].reshape(S, 9), m[:, :3, 3]), dim=1).contiguous()
    return flat(pose_c), flat(gt_c), step, rem, total_out, torch.stack((rot, trans)).double(), (weights.sum() + 1e-20).reshape(1)


def _tube_apply(rows, slot, pose12):
    r = pose12[:, :9].view(-1, 3, 3)[slot.long()]
    return torch.matmul(r, rows[:, :3].unsqueeze(-1)).squeeze(-1) + pose12[slot.long(), 9:]


def tube_gap_forward(rows, slot, pose_c, gt_c):
    gap = _tube_apply(rows, slot, pose_c) - _tube_apply(rows, slot, gt_c)     # models/tpointnet.py:276-281
    z = torch.zeros(rows.shape[0])
    return torch.stack((torch.norm(gap, p=2, dim=1), torch.norm(gap, p=1, dim=1), z, z), dim=1)


def tube_finish(slot_sums, count, weights, wsum):
    mean = slot_sums[:, :2] / count.clamp(min=1.0)[:, None]
    return (mean * weights[:, None]).sum(0) / wsum                            # models/tpointnet.py:282-286


def tube_gap_backward(rows, slot, pose_c, gt_c, weights, count, wsum, grad_l1, grad_l2):
    pc = pose_c.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        gap = _tube_apply(rows, slot, pc) - _tube_apply(rows, slot, gt_c)
        coef = (weights / (count.clamp(min=1.0) * wsum))[slot.long()]
        g1 = grad_l1 if grad_l1 is not None else torch.zeros(())
        g2 = grad_l2 if grad_l2 is not None else torch.zeros(())
        per_point = (g1 * torch.norm(gap, p=2, dim=1) + g2 * torch.norm(gap, p=1, dim=1)) * coef
        # the kernel's per-point rows: gradient of this point's term w.r.t. the 12 pose entries of its slot
        ge = torch.autograd.grad(per_point.sum(), gap)[0]
    out = torch.zeros(rows.shape[0], 16)
    out[:, :9] = (ge[:, :, None] * rows[:, None, :3]).reshape(-1, 9)
    out[:, 9:12] = ge
    return out


def tube_pose_backward(pose_vec, remaining, slot_centre, weights, wsum, grad_pose, grad_rot, grad_trans, n_frames):
    from pcaccumulation_amd.tpointnet import evaluate_pose
    pv = pose_vec.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        pose_c, _, gt_vec, _ = _tube_slot_terms(pv, remaining, slot_centre, n_frames)
        flat = torch.cat((pose_c[:, :3, :3].reshape(-1, 9), pose_c[:, :3, 3]), dim=1)
        obj = (flat * grad_pose[:, :12]).sum().double()
        rot, trans = evaluate_pose(pv, gt_vec, weights)
        if grad_rot is not None:
            obj = obj + grad_rot * rot
        if grad_trans is not None:
            obj = obj + grad_trans * trans
    return torch.autograd.grad(obj, pv)[0]


def inv4x4(m):
    """torch.linalg.inv of the pose tables (models/motionnet.py:100, models/alignnet.py:33)."""
    return torch.linalg.inv(m.float())


NAMES = ['voxelize', 'cell_index', 'frame_pillars', 'csr_build', 'segment_mean3_maxlabel', 'segment_max',
         'segment_max_backward', 'segment_sum', 'pillar_scatter', 'gather_rows', 'bilinear_gather',
         'bilinear_gather_backward', 'bev_warp', 'rigid_transform', 'chamfer_forward', 'chamfer_backward',
         'rows_linear', 'rows_wgrad', 'rows_linear_supported', 'pfn_features', 'scatter_sum_small', 'sinkhorn_kabsch', 'cluster', 'sample_subsets', 'upload_small', 'bilinear_gather_backward_sorted', 'prep_points', 'sinkhorn_forward', 'sinkhorn_backward',
         'seg_loss_forward', 'seg_loss_backward', 'offset_loss_forward', 'offset_loss_backward', 'frames_max', 'frames_max_backward', 'svd3', 'svd3_backward',
         'tube_rows', 'tube_code', 'tube_code_backward', 'tube_pose_forward', 'tube_gap_forward', 'tube_finish', 'tube_gap_backward', 'tube_pose_backward', 'inv4x4', 'compact_mask']


def compact_mask(mask, size):
    """Indices of the non-zero entries, ascending (include/pcacc.h: pcacc_compact_mask)."""
    return torch.nonzero_static(mask, size=int(size))[:, 0]


def install(monkeypatch=None):
    """Patch pcaccumulation_amd.native in this process (pytest monkeypatch when given, else plain setattr)."""
    from pcaccumulation_amd import native
    for name in NAMES:
        if monkeypatch is not None:
            monkeypatch.setattr(native, name, globals()[name])
        else:
            setattr(native, name, globals()[name])


def sample_subsets(counts, k, seed):
    g = torch.Generator().manual_seed(int(seed) % (2 ** 63))
    rows = []
    for n in counts.tolist():
        if n > k:
            rows.append(torch.randperm(n, generator=g)[:k])
        else:
            c = torch.arange(k)
            c[n:] = max(n - 1, 0)
            rows.append(c)
    return torch.stack(rows)


def upload_small(values, dtype, device):
    return torch.as_tensor(values, dtype=dtype).contiguous()


def bilinear_gather_backward_sorted(grad_out, shape, points, map_idx, x_scale, y_scale, out_dtype=torch.float32):
    return bilinear_gather_backward(grad_out.float(), shape, points, map_idx, x_scale, y_scale).to(out_dtype)


def prep_points(points, tsfm12, noise, noise_scale, scale, crop_xy, z_min, z_max, remove_ground, ground_z):
    p = points.double().numpy().copy()
    if tsfm12 is not None:
        t = tsfm12.double().numpy()
        p = (t[:9].reshape(3, 3) @ p.T + t[9:][:, None]).T
    if noise is not None:
        p = p + (noise.double().numpy() - 0.5) * noise_scale
    if tsfm12 is not None or noise is not None:
        p = p * scale
    keep = (np.abs(p[:, 0]) < crop_xy) & (np.abs(p[:, 1]) < crop_xy) & (p[:, 2] < z_max) & (p[:, 2] > z_min)
    if remove_ground:
        keep &= p[:, 2] > ground_z
    return torch.from_numpy(np.ascontiguousarray(p)), torch.from_numpy(keep.astype(np.uint8))


def _sinkhorn_ref(log_alpha, n_iters):
    la = torch.nn.functional.pad(log_alpha, (0, 1, 0, 1))
    for _ in range(n_iters):
        la = torch.cat((la[:, :-1, :] - torch.logsumexp(la[:, :-1, :], dim=2, keepdim=True), la[:, -1, None, :]), dim=1)
        la = torch.cat((la[:, :, :-1] - torch.logsumexp(la[:, :, :-1], dim=1, keepdim=True), la[:, :, -1, None]), dim=2)
    return la[:, :-1, :-1]


def sinkhorn_forward(log_alpha, n_iters):
    n = torch.tensor([n_iters])                        # rides along in the slot of the recorded vectors
    return _sinkhorn_ref(log_alpha.detach(), n_iters), n, n


def sinkhorn_backward(grad_log_perm, log_alpha, lse_rows, lse_cols):
    with torch.enable_grad():
        x = log_alpha.detach().clone().requires_grad_(True)
        _sinkhorn_ref(x, int(lse_rows[0])).backward(grad_log_perm)
    return x.grad


def _seg_rows(logits, plane, rows, n):
    """The selected [n,2] logit rows of either layout (see include/pcacc.h L1) and their flat positions."""
    flat = logits.detach().reshape(-1)
    i = rows if rows is not None else torch.arange(n)
    a = (i // plane) * 2 * plane + i % plane if plane > 0 else i * 2
    step = plane if plane > 0 else 1
    return torch.stack((flat[a], flat[a + step]), 1).float(), a, step


def seg_loss_forward(logits, plane, labels, rows, n):
    z, _, _ = _seg_rows(logits, plane, rows, n)
    y = labels[rows] if rows is not None else labels[:n]
    r = oracle.seg_loss(_np(z), _np(y))
    metric = np.stack([r['metric'][k] for k in ('intersection', 'union', 'pred_positives', 'gt_positives')])
    grads = torch.from_numpy(np.stack([r['grad_bce'], r['grad_lovasz']]))      # rides along in the slot of the Jaccard gradients
    return torch.tensor([r['bce_loss'], r['lovasz_loss']], dtype=torch.float32), torch.from_numpy(metric), grads, torch.zeros(8)


def seg_loss_backward(logits, plane, labels, rows, n, lovasz_grad, saved, grad_bce, grad_lovasz):
    _, a, step = _seg_rows(logits, plane, rows, n)
    gb = grad_bce if grad_bce is not None else 0.0
    gl = grad_lovasz if grad_lovasz is not None else 0.0
    g = (lovasz_grad[0] * gb + lovasz_grad[1] * gl).to(logits.dtype)
    out = torch.zeros(logits.numel(), dtype=logits.dtype)
    out[a] = g[:, 0]
    out[a + step] = g[:, 1]
    return out.reshape(logits.shape)


def offset_loss_forward(points, time_indice, inst_labels, label_base, ego_motion, inst_motion, n_frames, transformed_points, offset_est, rows):
    n = points.shape[0]
    fb = np.zeros(n, np.int64)
    fb[_np(rows) if rows is not None else slice(None)] = 1
    base = _np(label_base).tolist() + [inst_motion.shape[0]]
    motions = [_np(inst_motion)[base[b]:base[b + 1]] for b in range(len(base) - 1)]
    r = oracle.offset_loss(_np(points), _np(time_indice), _np(inst_labels), fb, _np(ego_motion), motions, _np(transformed_points), _np(offset_est))
    return (torch.tensor([r['offset_norm_loss'], r['offset_dir_loss'], r['offset_l2_error']], dtype=torch.float32),
            torch.from_numpy(r['offset_gt'].astype(np.float32)))


def offset_loss_backward(offset_gt, offset_est, rows, grad_norm, grad_dir):
    est = offset_est.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        e = est[rows] if rows is not None else est
        norm = torch.abs(offset_gt - e).mean(dim=0).sum()
        ngt = offset_gt / (torch.norm(offset_gt, dim=1, p=2).unsqueeze(-1) + 1e-20)
        nest = e / (torch.norm(e, dim=1, p=2).unsqueeze(-1) + 1e-20)
        dirl = (1 - (ngt * nest).sum(-1)).mean()
        total = norm * (grad_norm if grad_norm is not None else 0.0) + dirl * (grad_dir if grad_dir is not None else 0.0)
    return torch.autograd.grad(total, est)[0]


def frames_max(x):
    """models/stpn.py:83: max over the frame axis; the winning frame is the lowest index attaining it."""
    v = x.detach().float()
    out = v.max(dim=1)[0]
    arg = (v == out.unsqueeze(1)).to(torch.uint8).argmax(dim=1).to(torch.uint8)
    return out.to(x.dtype), arg


def frames_max_backward(grad_out, arg, frames):
    t = torch.arange(frames).view((1, frames) + (1,) * (grad_out.dim() - 1))
    return torch.where(arg.unsqueeze(1).long() == t, grad_out.unsqueeze(1), torch.zeros((), dtype=grad_out.dtype))


def svd3(a):
    """toolbox/register_utils.py:293: the library SVD the reference calls."""
    u, s, v = torch.svd(a.detach())
    return u.contiguous(), s.contiguous(), v.contiguous()


def svd3_backward(u, s, v, gu, gs, gv):
    """Gradient of a = u diag(s) v^T through the library's own SVD derivative."""
    a = ((u * s[:, None, :]) @ v.transpose(1, 2)).detach().requires_grad_(True)
    with torch.enable_grad():
        uu, ss, vv = torch.svd(a)
        # the decomposition of the rebuilt matrix may differ from (u, v) by a sign per column: align the incoming gradients
        sign = torch.sign((uu * u).sum(dim=1, keepdim=True))
        total = 0
        if gu is not None:
            total = total + (uu * sign * gu).sum()
        if gv is not None:
            total = total + (vv * sign * gv).sum()
        if gs is not None:
            total = total + (ss * gs).sum()
    return torch.autograd.grad(total, a)[0]
