"""ctypes binding of the CPU twin of the C ABI (oracle/csrc/pcacc_twin.c) -- TEST INFRASTRUCTURE.

`orc_<op>` has the argument meaning and layouts of `pcacc_<op>` in include/pcacc.h, on host pointers.  numpy arrays in, numpy arrays
out; used by the tests (twin against the numpy restatement, HIP against twin) and by oracle/cpu_backend.py."""
import ctypes

import numpy as np

from . import lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _i64(v):
    return ctypes.c_int64(int(v))


def _f(v):
    return ctypes.c_float(float(v))


def _d(v):
    return ctypes.c_double(float(v))


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype)


def _ok(rc, what):
    if rc != 0:
        raise ValueError('%s: error %d' % (what, rc))


def cell_index(coords, nx, ny, nt, n_batch):
    is64 = coords.dtype == np.float64
    coords = _c(coords, np.float64 if is64 else np.int32)
    m = coords.shape[0]
    cell = np.empty(m, np.int32)
    c2p = np.empty(n_batch * nt * ny * nx, np.int32)
    _ok(lib().orc_cell_index(_p(coords), int(is64), _i64(m), nx, ny, nt, n_batch, _p(cell), _p(c2p)), 'cell_index')
    return cell, c2p


def csr_build(p2v, m):
    p2v = _c(p2v, np.int32)
    offs, order = np.empty(m + 1, np.int32), np.empty(p2v.shape[0], np.int32)
    _ok(lib().orc_csr_build(_p(p2v), _i64(p2v.shape[0]), _i64(m), _p(offs), _p(order)), 'csr_build')
    return offs, order


def segment_mean3_maxlabel(points, labels, offs, order, m):
    points = _c(points, np.float32)
    labels = _c(labels, np.int64) if labels is not None else None
    mean, lab = np.empty((m, 3), np.float32), np.empty(m, np.int64)
    _ok(lib().orc_segment_mean3_maxlabel(_p(points), _p(labels), _p(_c(offs, np.int32)), _p(_c(order, np.int32)), _i64(m), _p(mean),
                                         _p(lab)), 'segment_mean3_maxlabel')
    return mean, (lab if labels is not None else None)


def segment_max(src, offs, order, m):
    src = _c(src, np.float32)
    n, c = src.shape
    out, arg = np.empty((m, c), np.float32), np.empty((m, c), np.int32)
    _ok(lib().orc_segment_max(_p(src), c, _p(_c(offs, np.int32)), _p(_c(order, np.int32)), _i64(n), _i64(m), _p(out), _p(arg)), 'segment_max')
    return out, arg


def segment_max_backward(grad_out, arg, p2v, n):
    g = _c(grad_out, np.float32)
    c = g.shape[1]
    out = np.empty((n, c), np.float32)
    _ok(lib().orc_segment_max_backward(_p(g), _p(_c(arg, np.int32)), _p(_c(p2v, np.int32)), _i64(n), c, _p(out)), 'segment_max_backward')
    return out


def segment_sum(src, offs, order, m):
    src = _c(src, np.float32)
    n, c = src.shape
    out = np.empty((m, c), np.float32)
    _ok(lib().orc_segment_sum(_p(src), c, _p(_c(offs, np.int32)), _p(_c(order, np.int32)), _i64(n), _i64(m), _p(out)), 'segment_sum')
    return out


def pfn_features(points, p2v, pillar_mean, coords, time_indice, vx, vy, x_offset, y_offset, scale, n_frames):
    points, p2v, pm = _c(points, np.float32), _c(p2v, np.int32), _c(pillar_mean, np.float32)
    is64 = coords.dtype == np.float64
    coords = _c(coords, np.float64 if is64 else np.int32)
    ti = _c(time_indice, np.float64)
    n = points.shape[0]
    out = np.empty((n, 9), np.float32)
    tcol = ctypes.c_void_p(ti.ctypes.data + 8 * (ti.shape[1] - 1))
    _ok(lib().orc_pfn_features(_p(points), _p(p2v), _p(pm), _p(coords), int(is64), tcol, _i64(ti.shape[1]), _i64(n), _d(vx), _d(vy),
                               _d(x_offset), _d(y_offset), _f(scale), _f(n_frames), _p(out)), 'pfn_features')
    return out


def pillar_scatter(feats, cell2pillar):
    feats, c2p = _c(feats, np.float32), _c(cell2pillar, np.int32)
    canvas = np.empty((c2p.shape[0], feats.shape[1]), np.float32)
    _ok(lib().orc_pillar_scatter(_p(feats), _p(c2p), _i64(c2p.shape[0]), feats.shape[1], _p(canvas)), 'pillar_scatter')
    return canvas


def gather_rows(src, idx):
    src, idx = np.ascontiguousarray(src), _c(idx, np.int32)
    out = np.empty((idx.shape[0],) + src.shape[1:], src.dtype)
    row_bytes = src.strides[0] if src.shape[0] > 0 else int(np.prod(src.shape[1:])) * src.itemsize
    _ok(lib().orc_gather_rows(_p(src), int(row_bytes), _p(idx), _i64(idx.shape[0]), _p(out)), 'gather_rows')
    return out


def bilinear_gather(fmap, points, map_idx, x_scale, y_scale):
    fmap, points, map_idx = _c(fmap, np.float32), _c(points, np.float32), _c(map_idx, np.int32)
    n_maps, h, w, c = fmap.shape
    out = np.empty((points.shape[0], c), np.float32)
    _ok(lib().orc_bilinear_gather(_p(fmap), n_maps, h, w, c, _p(points), _p(map_idx), _i64(points.shape[0]), _f(x_scale), _f(y_scale),
                                  _p(out)), 'bilinear_gather')
    return out


def bev_warp(bev_cl, inv_pose, x_reso, y_reso, x_min, y_min):
    bev, inv = _c(bev_cl, np.float32), _c(inv_pose, np.float32)
    b, t, h, w, c = bev.shape
    out = np.empty_like(bev)
    _ok(lib().orc_bev_warp(_p(bev), _p(inv), b, t, h, w, c, _f(x_reso), _f(y_reso), _f(x_min), _f(y_min), _p(out)), 'bev_warp')
    return out


def rigid_transform(points, frame_idx, tsfm):
    points, idx, tsfm = _c(points, np.float32), _c(frame_idx, np.int32), _c(tsfm, np.float32).reshape(-1, 16)
    out = np.empty_like(points)
    _ok(lib().orc_rigid_transform(_p(points), _p(idx), _p(tsfm), _i64(points.shape[0]), _p(out)), 'rigid_transform')
    return out


def frames_max(x):
    x = _c(x, np.float32)
    s, t = x.shape[0], x.shape[1]
    plane = int(np.prod(x.shape[2:]))
    out, arg = np.empty((s,) + x.shape[2:], np.float32), np.empty((s,) + x.shape[2:], np.uint8)
    _ok(lib().orc_frames_max(_p(x), _i64(s), t, _i64(plane), _p(out), _p(arg)), 'frames_max')
    return out, arg
