"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain C for loop-shaped integer work, numpy fp32 for the rest) of the algorithms on
the PCAccumulation hot path and of the section 8f rows built on top of it
(C1 clustering, D1 data step, E1 evaluation, L1/L2 loss terms).  Every function cites the
reference file:line it follows (paths relative to /root/reference).

Who may import this package: tests/, __graft_entry__.smoke(), and the cpu_baseline leg of bench.py --
as the checker only.  Nothing under pcaccumulation_amd/ imports it; the product path fails loudly when
the HIP library is missing instead of falling back to this code.

Pinned by: tests/golden/*.npz, emitted by tests/golden/make_golden.py from the reference itself
imported in the build container (the reference ships no tests of its own), and
by oracle/_ref (the reference's own Chamfer C++ CPU path compiled from its sources) where present.
Third-party semantics restated here because the modules are absent from /root/reference:
torch_scatter.scatter (unpinned wheel, README.md:28) -- "parity unpinned" for that dependency beyond
its documented definition (sum/mean/max per index, empty segments = 0).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """libpcacc_oracle.so = csrc/pcacc_oracle.c (scalar restatements) + csrc/pcacc_twin.c (the OpenMP CPU twin of the C ABI)."""
    so = os.path.join(_HERE, 'libpcacc_oracle.so')
    srcs = [os.path.join(_HERE, 'csrc', f) for f in ('pcacc_oracle.c', 'pcacc_twin.c')]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fopenmp', '-fPIC', '-shared', '-o', so] + srcs + ['-lm'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ------------------------------------------------------------------------------------------------
# A1  voxelisation  (libs/voxel_generator.py:4-61, 64-114, 117-154)
# ------------------------------------------------------------------------------------------------
def grid_size(voxel_size, pc_range):
    """libs/voxel_generator.py:123-125: round((max - min) / voxel_size) in fp32."""
    vs = np.asarray(voxel_size, np.float32)
    r = np.asarray(pc_range, np.float32)
    return np.round((r[3:] - r[:3]) / vs).astype(np.int64)


def voxelize(points, voxel_size, pc_range, n_sweeps, max_voxels=None):
    """Voxelization.__call__ (libs/voxel_generator.py:131-154). points [N,4] f32 = (x,y,z,t)."""
    points = np.ascontiguousarray(points, np.float32)
    vs = np.ascontiguousarray(voxel_size, np.float32)
    r = np.ascontiguousarray(pc_range, np.float32)
    g = grid_size(vs, r)
    if max_voxels is None:
        max_voxels = int(g[0] * g[1] * g[2] * n_sweeps)
    g32 = np.ascontiguousarray(g, np.int32)
    table = np.full(int(g[2] * g[1] * g[0] * n_sweeps), -1, np.int32)
    coors = np.zeros((max_voxels, 4), np.int32)
    npv = np.zeros(max_voxels, np.int32)
    p2v = np.empty((points.shape[0], 1), np.int32)
    f = lib().orc_voxelize
    f.restype = ctypes.c_int
    m = f(_p(points), ctypes.c_int64(points.shape[0]), _p(vs), _p(r), _p(g32), ctypes.c_int32(n_sweeps),
          ctypes.c_int32(max_voxels), _p(table), _p(coors), _p(npv), _p(p2v))
    return {
        'coordinates': coors[:m].copy(),
        'num_voxels': np.array([m], dtype=np.int64),
        'shape': np.hstack((g, np.array([n_sweeps]))).astype(np.int64),
        'point_to_voxel_map': p2v,
        'num_points_per_voxel': npv[:m].copy(),
    }


# ------------------------------------------------------------------------------------------------
# A3 / A4 pooling  (models/motionnet.py:159-160, models/pillar_encoder.py:116,120)
# ------------------------------------------------------------------------------------------------
def segment_mean(src, seg, m):
    src = np.ascontiguousarray(src, np.float32)
    seg = np.ascontiguousarray(seg, np.int64)
    out = np.empty((m, src.shape[1]), np.float32)
    lib().orc_segment_mean_f32(_p(src), _p(seg), ctypes.c_int64(src.shape[0]), ctypes.c_int64(m),
                               ctypes.c_int(src.shape[1]), _p(out))
    return out


def segment_max_label(labels, seg, m):
    labels = np.ascontiguousarray(labels, np.int64).reshape(-1)
    seg = np.ascontiguousarray(seg, np.int64)
    out = np.empty(m, np.int64)
    lib().orc_segment_max_i64(_p(labels), _p(seg), ctypes.c_int64(labels.shape[0]), ctypes.c_int64(m), _p(out))
    return out[:, None]


def segment_max(src, seg, m):
    """Returns (max [m,C], arg [m,C] = lowest point index attaining it)."""
    src = np.ascontiguousarray(src, np.float32)
    seg = np.ascontiguousarray(seg, np.int64)
    out = np.empty((m, src.shape[1]), np.float32)
    arg = np.empty((m, src.shape[1]), np.int32)
    lib().orc_segment_max_f32(_p(src), _p(seg), ctypes.c_int64(src.shape[0]), ctypes.c_int64(m),
                              ctypes.c_int(src.shape[1]), _p(out), _p(arg))
    return out, arg


def _linear(x, w, b=None):
    y = x @ w.T.astype(np.float32)
    return y + b if b is not None else y


def _relu(x):
    return np.maximum(x, 0)


def _resblock(x, sd, prefix):
    """ResnetBlockFC.forward, models/pillar_encoder.py:46-55."""
    net = _linear(_relu(x), sd[prefix + 'fc_0.weight'], sd[prefix + 'fc_0.bias'])
    dx = _linear(_relu(net), sd[prefix + 'fc_1.weight'], sd[prefix + 'fc_1.bias'])
    return _linear(x, sd[prefix + 'shortcut.weight']) + dx


def pfn_features(points, p2v, coordinates, pillar_mean, time_indice, voxel_size, pc_range, n_frames):
    """The 9 per-point inputs, models/pillar_encoder.py:98-110 (fp32 after the .float() cast)."""
    points = np.asarray(points, np.float32)
    vx, vy = voxel_size[0], voxel_size[1]
    x_off, y_off = vx / 2 + pc_range[0], vy / 2 + pc_range[1]
    mc = np.asarray(coordinates, np.float64)[p2v]
    d_mean = points - np.asarray(pillar_mean, np.float32)[p2v]
    f_center = np.zeros_like(points[:, :2])
    # f64 coordinate * python float, subtracted from the f32 point and stored into an f32 tensor
    f_center[:, 0] = (points[:, 0].astype(np.float64) - (mc[:, 3] * vx + x_off)).astype(np.float32)
    f_center[:, 1] = (points[:, 1].astype(np.float64) - (mc[:, 2] * vy + y_off)).astype(np.float32)
    feats = np.concatenate([points.astype(np.float64), d_mean.astype(np.float64), f_center.astype(np.float64),
                            np.asarray(time_indice, np.float64)[:, 1:2]], axis=1).astype(np.float32)
    feats[:, :-1] /= np.float32(abs(pc_range[0]))
    feats[:, -1] /= np.float32(n_frames)
    return feats


def pfn_forward(sd, feats, p2v, m, depth=3, prefix='pillar_encoder.'):
    """PillarFeatureNet.forward after feature build, models/pillar_encoder.py:112-122."""
    net = _linear(feats, sd[prefix + 'fc_pos.weight'], sd[prefix + 'fc_pos.bias'])
    net = _resblock(net, sd, prefix + 'blocks.0.')
    for i in range(1, depth):
        pooled = segment_max(net, p2v, m)[0][p2v]
        net = _resblock(np.concatenate([net, pooled], axis=1), sd, prefix + 'blocks.%d.' % i)
    out = _linear(net, sd[prefix + 'fc_c.weight'], sd[prefix + 'fc_c.bias'])
    return segment_max(out, p2v, m)[0]


# ------------------------------------------------------------------------------------------------
# A5 / A6 pillar <-> BEV canvas  (models/pillar_encoder.py:125-174, 177-204)
# ------------------------------------------------------------------------------------------------
def _cell_index(coords, nx, ny):
    c = np.asarray(coords)
    return (c[:, 4] * nx * ny + c[:, 2] * nx + c[:, 3]).astype(np.int64)


def scatter_point_pillar(voxel_features, coords, batch_size, input_shape):
    """[M,C] + coords [M,5]=(b,z,y,x,t) -> [B,C,nt,ny,nx]; later rows win on duplicate cells."""
    vf = np.asarray(voxel_features)
    nx, ny, nt = int(input_shape[0]), int(input_shape[1]), int(input_shape[3])
    c = vf.shape[1]
    out = np.zeros((batch_size, c, nt * ny * nx), vf.dtype)
    b = np.asarray(coords)[:, 0].astype(np.int64)
    idx = _cell_index(coords, nx, ny)
    for bi in range(batch_size):
        sel = b == bi
        out[bi][:, idx[sel]] = vf[sel].T
    return out.reshape(batch_size, c, nt, ny, nx)


def inverse_scatter_point_pillar(canvas, coords, batch_size, input_shape):
    """[B,C,nt,ny,nx] -> [M,C], rows grouped by batch index ascending (pillar_encoder.py:193-203)."""
    cv = np.asarray(canvas)
    nx, ny = int(input_shape[0]), int(input_shape[1])
    c = cv.shape[1]
    b = np.asarray(coords)[:, 0].astype(np.int64)
    idx = _cell_index(coords, nx, ny)
    outs = []
    for bi in range(batch_size):
        sel = b == bi
        outs.append(cv[bi].reshape(c, -1)[:, idx[sel]].T)
    return np.concatenate(outs, axis=0)


# ------------------------------------------------------------------------------------------------
# A9 / A11 bilinear sampling  (F.grid_sample, mode='bilinear', align_corners=False)
# ------------------------------------------------------------------------------------------------
def _grid_sample(feat, gx, gy, padding):
    """feat [C,H,W] f32, normalised coords gx (width axis), gy (height axis) [K] -> [K,C].

    ATen grid_sampler_2d: x = ((g + 1) * W - 1) / 2; 'border' clamps x to [0, W-1] before the
    corner split; 'zeros' drops out-of-range corners.  All arithmetic in fp32.
    """
    feat = np.asarray(feat, np.float32)
    c, h, w = feat.shape
    f32 = np.float32
    x = ((np.asarray(gx, f32) + f32(1)) * f32(w) - f32(1)) / f32(2)
    y = ((np.asarray(gy, f32) + f32(1)) * f32(h) - f32(1)) / f32(2)
    if padding == 'border':
        x = np.minimum(np.maximum(x, f32(0)), f32(w - 1))
        y = np.minimum(np.maximum(y, f32(0)), f32(h - 1))
    x0, y0 = np.floor(x), np.floor(y)
    x1, y1 = x0 + f32(1), y0 + f32(1)
    wts = [((x1 - x) * (y1 - y), x0, y0), ((x - x0) * (y1 - y), x1, y0),
           ((x1 - x) * (y - y0), x0, y1), ((x - x0) * (y - y0), x1, y1)]
    out = np.zeros((x.shape[0], c), f32)
    flat = feat.reshape(c, -1)
    for wt, xi, yi in wts:
        ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
        xi_c = np.clip(xi, 0, w - 1).astype(np.int64)
        yi_c = np.clip(yi, 0, h - 1).astype(np.int64)
        v = flat[:, yi_c * w + xi_c].T
        out += (wt * ok.astype(f32))[:, None] * v
    return out


def ungrid(feats, points, pc_range, time_indice):
    """models/pillar_encoder.py:231-267: per-point bilinear sample (border) of [B,C,H,W] at
    (x/|x_min|, y/|y_min|); output rows grouped by batch index ascending.  Does not mutate points."""
    feats = np.asarray(feats, np.float32)
    pts = np.asarray(points, np.float32)
    u = pts[:, 0] / np.float32(abs(pc_range[0]))
    v = pts[:, 1] / np.float32(abs(pc_range[1]))
    b = np.asarray(time_indice)[:, 0]
    outs = []
    for bi in range(feats.shape[0]):
        sel = b == bi
        outs.append(_grid_sample(feats[bi], u[sel], v[sel], 'border'))
    return np.concatenate(outs, axis=0) if outs else np.zeros((0, feats.shape[1]), np.float32)


def temporal_ungrid(feats, points, pc_range, time_indice):
    """models/pillar_encoder.py:206-228. feats [B,T,C,H,W]."""
    feats = np.asarray(feats, np.float32)
    ti = np.asarray(time_indice)
    out = np.zeros((points.shape[0], feats.shape[2]), np.float32)
    for t in range(feats.shape[1]):
        sel = ti[:, 1] == t
        if sel.sum():
            out[sel] = ungrid(feats[:, t], np.asarray(points)[sel], pc_range, ti[sel])
    return out


def get_transformed_grid(pose, h, w, x_reso, y_reso, x_min, y_min):
    """models/motionnet.py:45-80: pixel centres -> metres -> pose[:2,:2] @ g + pose[:2,3] -> /|min|."""
    f32 = np.float32
    xx = np.tile(np.arange(w, dtype=f32)[None, :] + f32(0.5), (h, 1))
    yy = np.tile(np.arange(h, dtype=f32)[:, None] + f32(0.5), (1, w))
    gx = (xx * f32(x_reso) + f32(x_min)).reshape(-1)
    gy = (yy * f32(y_reso) + f32(y_min)).reshape(-1)
    p = np.asarray(pose, f32)
    tx = p[0, 0] * gx + p[0, 1] * gy + p[0, 3]
    ty = p[1, 0] * gx + p[1, 1] * gy + p[1, 3]
    return tx / f32(abs(x_min)), ty / f32(abs(y_min))


def warp_feats(bev_feats, pose_est, resolution, pc_range):
    """models/motionnet.py:82-114.  bev_feats [B,T,C,H,W]; frames 1..T-1 are resampled (zeros
    padding) at the inverse-pose grid; slot 0 holds frame T-1 UNWARPED (the loop variable leaks,
    motionnet.py:100,111) -- kept, the released weights were trained with it."""
    bev = np.asarray(bev_feats, np.float32)
    b_, t_, c, h, w = bev.shape
    out = np.empty_like(bev)
    for b in range(b_):
        out[b, 0] = bev[b, t_ - 1]
        for t in range(1, t_):
            inv = np.linalg.inv(np.asarray(pose_est[b, t], np.float32)).astype(np.float32)
            gx, gy = get_transformed_grid(inv, h, w, resolution[0], resolution[1], pc_range[0], pc_range[1])
            out[b, t] = _grid_sample(bev[b, t], gx, gy, 'zeros').T.reshape(c, h, w)
    return out


def transform_points(points, time_indice, transformation):
    """models/motionnet.py:117-135: p' = R_{b,t} p + t_{b,t} per point (fp32)."""
    pts = np.asarray(points, np.float32)
    ti = np.asarray(time_indice).astype(np.int64)
    tr = np.asarray(transformation, np.float32)[ti[:, 0], ti[:, 1]]
    return (np.einsum('nij,nj->ni', tr[:, :3, :3], pts) + tr[:, :3, 3]).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# A8 ego-motion: cost, Sinkhorn, weighted Kabsch
# ------------------------------------------------------------------------------------------------
def square_distance(src, dst, normalised=False):
    """toolbox/utils.py:125-144 (clamped at 1e-12)."""
    src, dst = np.asarray(src, np.float32), np.asarray(dst, np.float32)
    d = np.float32(-2) * (src @ dst.T)
    if normalised:
        d = d + np.float32(2)
    else:
        d = d + (src ** 2).sum(-1)[:, None] + (dst ** 2).sum(-1)[None, :]
    return np.maximum(d, np.float32(1e-12))


def _logsumexp(a, axis):
    m = a.max(axis=axis, keepdims=True)
    return m + np.log(np.exp(a - m).sum(axis=axis, keepdims=True))


def sinkhorn(log_alpha, n_iters):
    """models/egomotion.py:100-137: zero-padded slack row/column, last row/column not normalised."""
    la = np.asarray(log_alpha, np.float32)
    pad = np.zeros((la.shape[0] + 1, la.shape[1] + 1), np.float32)
    pad[:-1, :-1] = la
    for _ in range(n_iters):
        pad[:-1, :] = pad[:-1, :] - _logsumexp(pad[:-1, :], 1)
        pad[:, :-1] = pad[:, :-1] - _logsumexp(pad[:, :-1], 0)
    return pad[:-1, :-1]


def kabsch(x1, x2, weights, eps=1e-7):
    """toolbox/register_utils.py:247-317 with normalize_w=True, best_k=0, w_threshold=0."""
    f32 = np.float32
    x1, x2 = np.asarray(x1, f32), np.asarray(x2, f32)
    w = np.asarray(weights, f32)
    w = w / (w.sum() + f32(eps))
    x1_mean = (w[None] @ x1) / (w.sum() + f32(eps))
    x2_mean = (w[None] @ x2) / (w.sum() + f32(eps))
    x1c, x2c = x1 - x1_mean, x2 - x2_mean
    cov = x1c.T @ (w[:, None] * x2c)
    u, s, vt = np.linalg.svd(cov.astype(f32))
    v = vt.T
    det = np.linalg.det((v.T @ u.T).astype(np.float64))
    d = np.diag(np.array([1, 1, det], f32))
    r = (v @ d @ u.T).astype(f32)
    t = x2_mean.T - r @ x1_mean.T
    return r, t.astype(f32)


def pairwise_ego_motion(feats_s, feats_t, coor_s, coor_t, choice_s, choice_t, duration, max_speed,
                        alpha, beta, n_iters):
    """models/egomotion.py:169-192 after key-point choice.  Returns (pose [4,4], perm [n,n])."""
    f32 = np.float32
    fs, cs = np.asarray(feats_s, f32)[choice_s], np.asarray(coor_s, f32)[choice_s]
    ft, ct = np.asarray(feats_t, f32)[choice_t], np.asarray(coor_t, f32)[choice_t]
    thr = duration * max_speed
    support = (square_distance(cs, ct) < f32(thr ** 2)).astype(f32)
    feat_dist = square_distance(fs, ft, normalised=True)
    softplus = f32(np.log1p(np.exp(np.float64(alpha))))
    affinity = -(feat_dist - softplus) / (f32(np.exp(np.float64(beta))) + f32(0.02))
    perm = np.exp(sinkhorn(affinity, n_iters)) * support
    rowsum = perm.sum(1, keepdims=True)
    weighted_t = (perm @ ct) / (rowsum + f32(1e-20))
    r, t = kabsch(cs, weighted_t, rowsum[:, 0])
    pose = np.eye(4, dtype=f32)
    pose[:3, :3] = r
    pose[:3, 3] = t[:, 0]
    return pose, perm


# ------------------------------------------------------------------------------------------------
# A12 Chamfer  (chamfer_distance/chamfer_distance.cpp:59-111, 114-177)
# ------------------------------------------------------------------------------------------------
def chamfer_forward(xyz1, xyz2):
    xyz1 = np.ascontiguousarray(xyz1, np.float32)
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1, d2 = np.empty((b, n), np.float32), np.empty((b, m), np.float32)
    i1, i2 = np.empty((b, n), np.int32), np.empty((b, m), np.int32)
    f = lib().orc_nnsearch
    f(b, n, m, _p(xyz1), _p(xyz2), _p(d1), _p(i1))
    f(b, m, n, _p(xyz2), _p(xyz1), _p(d2), _p(i2))
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, gd1, gd2, idx1, idx2):
    xyz1 = np.ascontiguousarray(xyz1, np.float32)
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    gd1, gd2 = np.ascontiguousarray(gd1, np.float32), np.ascontiguousarray(gd2, np.float32)
    idx1, idx2 = np.ascontiguousarray(idx1, np.int32), np.ascontiguousarray(idx2, np.int32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1, g2 = np.empty_like(xyz1), np.empty_like(xyz2)
    lib().orc_chamfer_backward(b, n, m, _p(xyz1), _p(xyz2), _p(gd1), _p(gd2), _p(idx1), _p(idx2), _p(g1), _p(g2))
    return g1, g2


# ------------------------------------------------------------------------------------------------
# M1-M3 metric definitions
# ------------------------------------------------------------------------------------------------
def compute_iou(predictions, gt, n_class=2, ignore_index=-1):
    """libs/loss.py:17-50: per-class counts divided by 1e3."""
    predictions, gt = np.asarray(predictions), np.asarray(gt)
    inter, union, pp, gp = [], [], [], []
    for c in range(n_class):
        if c == ignore_index:
            continue
        sg, sp = gt == c, predictions == c
        i = (predictions[sg] == c).sum() / 1e3
        pp.append(sp.sum() / 1e3)
        gp.append(sg.sum() / 1e3)
        inter.append(i)
        union.append(sp.sum() / 1e3 + sg.sum() / 1e3 - i)
    return {'intersection': np.array(inter), 'union': np.array(union),
            'pred_positives': np.array(pp), 'gt_positives': np.array(gp)}


def mean_iou(stats_list):
    """toolbox/metrics.py:43-60 over accumulated batch stats: IoU = sum(I) / (sum(U) + 1e-20), class mean."""
    i = sum(s['intersection'] for s in stats_list)
    u = sum(s['union'] for s in stats_list)
    return float((i / (u + 1e-20)).mean())


def rotation_error(r1, r2):
    """toolbox/register_utils.py:19-43 (degrees)."""
    r = np.einsum('bji,bjk->bik', np.asarray(r1, np.float32), np.asarray(r2, np.float32))
    e = np.clip((np.trace(r, axis1=1, axis2=2) - 1) / 2, -1, 1)
    return np.degrees(np.arccos(e))


def translation_error(t1, t2):
    """toolbox/register_utils.py:46-56."""
    return np.linalg.norm(np.asarray(t1, np.float32) - np.asarray(t2, np.float32), axis=(1, 2))


def ego_motion_compensation(points, time_indice, tsfm):
    """toolbox/register_utils.py:59-70."""
    tr = np.asarray(tsfm)[np.asarray(time_indice).astype(np.int64)]
    return np.einsum('nij,nj->ni', tr[:, :3, :3], np.asarray(points)) + tr[:, :3, 3]


def reconstruct_sequence(points, time_indice, inst_labels, tsfm, n_frames):
    """toolbox/register_utils.py:73-93."""
    tr = np.asarray(tsfm).reshape(-1, 4, 4)
    idx = (np.asarray(inst_labels).astype(np.int64) * n_frames + np.asarray(time_indice).astype(np.int64))
    tr = tr[idx]
    return np.einsum('nij,nj->ni', tr[:, :3, :3], np.asarray(points)) + tr[:, :3, 3]


def scene_flow_epe(rec_est, input_points, time_indice, ego_motion_gt, inst_labels, inst_motion_gt, n_frames):
    """libs/tester.py:67-77: EPE per point = ||(rec_est - x) - (rec_gt - x)||_2 for points with t > 0."""
    x = np.asarray(input_points, np.float64)
    t = np.asarray(time_indice).astype(np.int64)
    gt = reconstruct_sequence(ego_motion_compensation(x, t, np.asarray(ego_motion_gt, np.float32)), t,
                              inst_labels, np.asarray(inst_motion_gt, np.float64), n_frames)
    err = np.linalg.norm((np.asarray(rec_est, np.float64) - x) - (gt - x), axis=1)
    return err[t > 0]


# ------------------------------------------------------------------------------------------------
# C1  test-mode instance clustering  (models/cluster.py:9-111)
#
# Third-party pieces, both absent from /root/reference:
#   torchsparse v1.4.0 (README.md:27) sparse_quantize(coords, voxel_size=1, return_index, return_inverse):
#     floor(coords / voxel_size) as int32, ravel hash (row-major over the min-shifted columns), then
#     np.unique(hash, return_index, return_inverse).  The reference keeps a copy of the same algorithm at
#     dataset_toolbox/prep_nuscene_waymo_sf/libs/spv_utils.py:7-75.  np.unique sorts the keys, so the kept
#     points come out in lexicographic (x, y, z) voxel order and each voxel keeps its FIRST point.
#   scikit-learn DBSCAN (unpinned): fit_predict with metric='euclidean' builds a KD-tree over the float64
#     copy of X; a point is core when >= min_samples points (itself included) have squared distance
#     <= eps*eps; clusters are numbered in the order of their smallest core index and a border point takes
#     the smallest-numbered cluster that has a core point within eps (sklearn/cluster/_dbscan_inner.pyx:
#     cluster k is expanded completely before cluster k+1 starts).
# dbscan() below restates that; tests pin it against sklearn.cluster.DBSCAN itself, which IS installed here.
# ------------------------------------------------------------------------------------------------
def sparse_quantize(coords):
    """torchsparse v1.4.0 torchsparse/utils/quantize.py (voxel_size=1): -> (first-occurrence indices, inverse map)."""
    c = np.floor(np.asarray(coords)).astype(np.int32)
    c = c - c.min(0)
    c = c.astype(np.uint64)
    cmax = c.max(0).astype(np.uint64) + np.uint64(1)
    h = np.zeros(c.shape[0], np.uint64)
    for k in range(c.shape[1] - 1):
        h += c[:, k]
        h *= cmax[k + 1]
    h += c[:, -1]
    _, idx, inv = np.unique(h, return_index=True, return_inverse=True)
    return idx, inv


def voxel_downsample(points, voxel_size):
    """models/cluster.py:9-13."""
    return sparse_quantize(np.round(np.asarray(points) / voxel_size))


def dbscan(x, eps, min_samples):
    """sklearn.cluster.DBSCAN(eps, min_samples, metric='euclidean').fit_predict(x), restated (see the block comment)."""
    x = np.asarray(x, np.float64)
    n = x.shape[0]
    r2 = float(eps) * float(eps)
    nbrs = []
    for s in range(0, n, 1024):
        d = np.zeros((min(1024, n - s), n))
        for k in range(x.shape[1]):                      # sum in column order, like the tree's rdist loop
            t = x[s:s + 1024, k:k + 1] - x[None, :, k]
            d = d + t * t
        nbrs += [np.nonzero(row <= r2)[0] for row in d]
    core = np.array([len(a) >= min_samples for a in nbrs], bool)
    labels = np.full(n, -1, np.int64)
    k = 0
    for i in range(n):
        if labels[i] != -1 or not core[i]:
            continue
        stack = [i]
        while stack:
            v = stack.pop()
            if labels[v] == -1:
                labels[v] = k
                if core[v]:
                    stack.extend(int(u) for u in nbrs[v] if labels[u] == -1)
        k += 1
    return labels


def canonicalise_random_indice(indice):
    """toolbox/utils.py:237-250."""
    uniq = {v: i for i, v in enumerate(sorted(set(indice)))}
    return [uniq[v] for v in indice]


def cluster_labels(points, eps, min_samples, min_p_cluster, estimator=None):
    """models/cluster.py:23-49 (inst_labels=None)."""
    lab = dbscan(points, eps, min_samples) if estimator is None else estimator.fit_predict(points)
    lab = np.asarray(lab).copy()
    for u in np.unique(lab).tolist():
        if (lab == u).sum() < min_p_cluster:
            lab[lab == u] = -1
    assert lab.min() <= 0
    if lab.min() == -1:
        return np.array(canonicalise_random_indice(lab.tolist()))
    return np.array(canonicalise_random_indice(lab.tolist())) + 1


def cluster_per_batch(mos, offset, points, eps, min_samples, min_p_cluster, use_offset=True, estimator=None):
    """models/cluster.py:52-84 with fb_labels=None.  mos [N], offset [N,2], points [N,3] (float32)."""
    mos, offset, points = np.asarray(mos), np.asarray(offset, np.float32), np.asarray(points, np.float32)
    sel = mos == 1
    full = np.zeros(mos.shape[0], np.int64)
    if sel.sum() > min_p_cluster:
        shifted = points.copy()
        shifted[:, :2] += offset
        src = (shifted if use_offset else points)[sel].copy()
        sub, inv = voxel_downsample(src, 0.05 if use_offset else 0.15)
        src[:, -1] = 0
        full[sel] = cluster_labels(src[sub], eps, min_samples, min_p_cluster, estimator)[inv]
    return full


def cluster_forward(points, mos, offset, time_indice, eps, min_samples, min_p_cluster, use_offset=True, estimator=None):
    """models/cluster.py:86-111: per-sample clustering of the points predicted moving -> inst_labels_est [N] (0 = none)."""
    b = np.asarray(time_indice)[:, 0]
    out = []
    for i in range(int(b.max() + 1)):
        s = b == i
        if s.sum():
            out.append(cluster_per_batch(np.asarray(mos)[s], np.asarray(offset)[s], np.asarray(points)[s], eps,
                                         min_samples, min_p_cluster, use_offset, estimator))
    return np.concatenate(out).astype(np.int64)


# ------------------------------------------------------------------------------------------------
# D1  host data step in front of the path  (libs/dataset.py:93-204: BaseDataset.prep_input)
# Random numbers come from numpy's global generator in the reference's order (seed it with np.random.seed).
# ------------------------------------------------------------------------------------------------
def sample_random_tsfm(rot_aug, shift_range):
    """libs/dataset.py:106-116: rotation about z by U(0, pi*rot_aug), shift U(-r, r) in x and y."""
    from scipy.spatial.transform import Rotation
    euler = [0, 0, np.random.uniform(0, np.pi * rot_aug)]
    rot = Rotation.from_euler('xyz', euler).as_matrix()
    shift = [np.random.uniform(-shift_range, shift_range), np.random.uniform(-shift_range, shift_range), 0]
    tsfm = np.eye(4)
    tsfm[:3, :3] = rot
    tsfm[:3, 3] = np.array(shift)
    return tsfm


def update_transformation_after_data_augmentation(aug_tsfm, ego_motion, inst_motion, n_frames):
    """libs/dataset.py:118-139: T' @ T @ T'^-1 for the ego poses and the instance motions."""
    a = aug_tsfm[None].repeat(n_frames, 0)
    ego = a @ ego_motion @ np.linalg.inv(a)
    im = inst_motion.reshape(-1, 4, 4)
    a = aug_tsfm[None].repeat(im.shape[0], 0)
    im = (a @ im @ np.linalg.inv(a)).reshape(-1, n_frames, 4, 4)
    return ego, im


def prep_points(raw_points, sd_labels, fb_labels, inst_labels, time_indice, ego_motion_gt, inst_motion_gt, p, augmentation=True):
    """libs/dataset.py:147-182 (everything of prep_input before the voxeliser).  p: dict with augment_noise, augment_shift_range,
    augment_scale_min, augment_scale_max, rot_aug, crop_xy, crop_z_min, crop_z_max, remove_ground, ground_height (+slack), n_frames."""
    pts = np.asarray(raw_points).copy()
    if augmentation:
        tsfm = sample_random_tsfm(p['rot_aug'], p['augment_shift_range'])
        pts = (tsfm[:3, :3] @ pts.T + tsfm[:3, 3][:, None]).T                                   # apply_tsfm, register_utils.py:199-206
        pts += (np.random.rand(pts.shape[0], 3) - 0.5) * p['augment_noise']
        pts = pts * np.random.uniform(p['augment_scale_min'], p['augment_scale_max'])
        ego_motion_gt, inst_motion_gt = update_transformation_after_data_augmentation(tsfm, ego_motion_gt, inst_motion_gt, p['n_frames'])
    keep = (np.abs(pts[:, 0]) < p['crop_xy']) & (np.abs(pts[:, 1]) < p['crop_xy']) & (pts[:, 2] < p['crop_z_max']) & (pts[:, 2] > p['crop_z_min'])
    pts, time_indice, sd_labels, fb_labels, inst_labels = (a[keep] for a in (pts, time_indice, sd_labels, fb_labels, inst_labels))
    if p['remove_ground']:
        up = pts[:, 2] > p['ground_height']
        pts, time_indice, sd_labels, fb_labels, inst_labels = (a[up] for a in (pts, time_indice, sd_labels, fb_labels, inst_labels))
    return {'input_points': pts, 'num_points': np.array([pts.shape[0]], dtype=np.int64), 'time_indice': time_indice[:, None],
            'sd_labels': sd_labels[:, None], 'inst_labels': inst_labels[:, None], 'ego_motion_gt': ego_motion_gt,
            'inst_motion_gt': inst_motion_gt, 'fb_labels': fb_labels[:, None]}


# ------------------------------------------------------------------------------------------------
# E1  scene-flow evaluation  (libs/tester.py:58-83; toolbox/sf_eval_utils.py:51-86)
# ------------------------------------------------------------------------------------------------
def flow_errors(rec_est, input_points, time_indice, ego_motion_gt, inst_labels, inst_motion_gt, n_frames):
    """libs/tester.py:58-83: (epe_per_point, relative_error) of the points with t > 0, float32 throughout (the tester's matmul
    with ego_motion_gt.float() only type-checks for float32 points)."""
    x = np.asarray(input_points, np.float32)
    t = np.asarray(time_indice).astype(np.int64)
    comp = ego_motion_compensation(x, t, np.asarray(ego_motion_gt, np.float32)).astype(np.float32)
    gt = reconstruct_sequence(comp, t, inst_labels, np.asarray(inst_motion_gt, np.float32), n_frames).astype(np.float32)
    est_flow, gt_flow = np.asarray(rec_est, np.float32) - x, gt - x
    epe = np.linalg.norm(est_flow - gt_flow, axis=1)
    rel = epe / (np.linalg.norm(gt_flow, axis=1) + 1e-20)
    return epe[t > 0], rel[t > 0]


def compute_sf_metrics(epe, rel):
    """toolbox/sf_eval_utils.py:71-86 (compute_sf_metrics_torch; its numpy twin :51-69 differs only in the median of an even
    count: np.median averages the two middle values, torch.median returns the lower one)."""
    return {'EPE3D': float(epe.mean()), 'EPE3D_med': float(np.sort(epe)[(epe.shape[0] - 1) // 2]),
            'Acc3DS': float(np.logical_or(epe < 0.05, rel < 0.05).mean()), 'Acc3DR': float(np.logical_or(epe < 0.1, rel < 0.1).mean()),
            'Outlier': float(np.logical_or(epe > 0.3, rel > 0.1).mean()), 'ROutlier': float(np.logical_and(epe > 0.3, rel > 0.3).mean())}


# ------------------------------------------------------------------------------------------------
# L1/L2  loss terms on the path's tensors  (libs/loss.py:90-137,194-250; libs/lovasz_softmax.py:56-94)
# ------------------------------------------------------------------------------------------------
def ce_weights(labels, n_classes=2, max_weights=50.0):
    """libs/loss.py:90-108 ('sqrt_inv_freq'): float32 counts + 1e-20, sqrt(total / count) clamped to [0, 50]."""
    counts = np.array([float((labels == c).sum()) + 1e-20 for c in range(n_classes)], np.float32)
    with np.errstate(over='ignore'):
        return np.clip(np.sqrt(counts.sum() / counts), 0, max_weights).astype(np.float32)


def lovasz_grad(fg_sorted):
    """libs/lovasz_softmax.py:56-68 (float32 cumulative sums, as the reference's .float().cumsum)."""
    fg_sorted = fg_sorted.astype(np.float32)
    gts = fg_sorted.sum(dtype=np.float32)
    inter = gts - np.cumsum(fg_sorted, dtype=np.float32)
    union = gts + np.cumsum(1 - fg_sorted, dtype=np.float32)
    jac = (1.0 - inter / union).astype(np.float32)
    jac[1:] = jac[1:] - jac[:-1]
    return jac


def seg_loss(logits, labels, n_classes=2, ignore_index=-1):
    """libs/loss.py:110-137: weighted cross entropy (ignore_index rows dropped, weights from the label frequencies), Lovasz-Softmax
    over the classes that are present (libs/lovasz_softmax.py:71-94, rows with the ignore label count as background of every
    class, as in the reference which passes no ignore value), the IoU counters of compute_iou, and the gradients of the two loss
    terms w.r.t. the logits (the Lovasz gradient vector is a constant, lovasz_softmax.py:92).  Ties between equal errors are
    ordered by row index (stable sort); the loss value does not depend on that choice, the (sub)gradient does."""
    z = np.asarray(logits, np.float32)
    y = np.asarray(labels).astype(np.int64)
    n = z.shape[0]
    m = z.max(1, keepdims=True)
    ez = np.exp(z - m)
    p = (ez / ez.sum(1, keepdims=True)).astype(np.float32)
    logp = (z - m) - np.log(ez.sum(1, keepdims=True))
    w = ce_weights(y, n_classes)
    keep = y != ignore_index
    safe = np.where(keep, y, 0)
    wi = w[safe] * keep
    wsum = wi.sum(dtype=np.float64)
    bce = -(wi * logp[np.arange(n), safe]).sum(dtype=np.float64) / wsum
    onehot = np.zeros_like(p)
    onehot[np.arange(n), safe] = 1
    grad_bce = (wi[:, None] * (p - onehot) / wsum).astype(np.float32)
    losses, dprob = [], np.zeros_like(p, dtype=np.float64)
    present = [c for c in range(n_classes) if (y == c).any()]
    for c in present:
        fg = (y == c).astype(np.float32)
        err = np.abs(fg - p[:, c])
        perm = np.argsort(-err, kind='stable')
        g = lovasz_grad(fg[perm])
        losses.append(np.dot(err[perm].astype(np.float64), g.astype(np.float64)))
        by_row = np.empty(n, np.float32)
        by_row[perm] = g
        dprob[:, c] = by_row * -np.sign(fg - p[:, c]) / len(present)
    lovasz = float(np.mean(losses)) if losses else 0.0
    grad_lov = (p * (dprob - (dprob * p).sum(1, keepdims=True))).astype(np.float32)
    pred = (z[:, 1] > z[:, 0]).astype(np.int64) if n_classes == 2 else z.argmax(1)
    return {'bce_loss': float(bce), 'lovasz_loss': lovasz, 'metric': compute_iou(pred, y, n_classes, ignore_index),
            'grad_bce': grad_bce, 'grad_lovasz': grad_lov}


def offset_loss(input_points, time_indice, inst_labels, fb_labels, ego_motion_gt, inst_motion_gt, transformed_points, offset_est):
    """libs/loss.py:194-250: ground-truth reconstruction per sample (ego compensation, then the instance motions), instance
    centres = mean of the reconstructed points per label, offsets of the foreground points to their centre (x, y) against the
    estimate: L1 term (mean per coordinate, summed), direction term (1 - cosine with 1e-20 in the norms), mean L2 error; plus
    the gradients of the first two w.r.t. offset_est and the ground-truth offsets (predictions['offset_gt'])."""
    pts = np.asarray(input_points, np.float32)
    tidx = np.asarray(time_indice).astype(np.int64)
    lab = np.asarray(inst_labels).reshape(-1).astype(np.int64)
    ego = np.asarray(ego_motion_gt, np.float32)
    n_frames = ego.shape[1]
    centres = np.zeros((pts.shape[0], 2), np.float32)
    for b in range(len(inst_motion_gt)):
        sel = tidx[:, 0] == b
        comp = ego_motion_compensation(pts[sel], tidx[sel, 1], ego[b]).astype(np.float32)
        rec = reconstruct_sequence(comp, tidx[sel, 1], lab[sel], np.asarray(inst_motion_gt[b], np.float32), n_frames).astype(np.float32)
        k = int(lab[sel].max()) + 1
        sums = np.zeros((k, 3), np.float64)
        np.add.at(sums, lab[sel], rec)
        cnt = np.maximum(np.bincount(lab[sel], minlength=k), 1)[:, None]
        centres[sel] = (sums / cnt).astype(np.float32)[lab[sel]][:, :2]
    fb = np.asarray(fb_labels).reshape(-1) == 1
    gt = (centres - np.asarray(transformed_points, np.float32)[:, :2])[fb]
    est = np.asarray(offset_est, np.float32)[fb]
    m = gt.shape[0]
    diff = (gt - est).astype(np.float64)
    norm_loss = np.abs(diff).mean(0).sum()
    l2 = np.linalg.norm(diff, axis=1).mean()
    gn, en = np.linalg.norm(gt.astype(np.float64), axis=1, keepdims=True), np.linalg.norm(est.astype(np.float64), axis=1, keepdims=True)
    ngt, nest = gt / (gn + 1e-20), est / (en + 1e-20)
    dir_loss = (1 - (ngt * nest).sum(1)).mean()
    grad_norm, grad_dir = np.zeros((pts.shape[0], 2), np.float32), np.zeros((pts.shape[0], 2), np.float32)
    grad_norm[fb] = -np.sign(diff) / m
    with np.errstate(invalid='ignore', divide='ignore'):
        d_norm = np.where(en > 0, est / en, 0.0)                       # torch.norm's backward at 0 is 0
        g = ngt / (en + 1e-20) - (ngt * est).sum(1, keepdims=True) / (en + 1e-20) ** 2 * d_norm
    grad_dir[fb] = -g / m
    return {'offset_norm_loss': float(norm_loss), 'offset_dir_loss': float(dir_loss), 'offset_l2_error': float(l2),
            'offset_gt': gt, 'grad_norm': grad_norm, 'grad_dir': grad_dir}
