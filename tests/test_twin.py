"""The CPU twin of the C ABI (oracle/csrc/pcacc_twin.c, OpenMP) against the numpy restatement it shadows -- both are test
infrastructure; the numpy functions are the ones pinned by the reference-generated golden vectors (test_oracle_golden.py).
The GPU leg (test_hip_ops.py::test_hip_against_cpu_twin) diffs the HIP library against the twin on raw buffers."""
import numpy as np

import oracle
from oracle import twin
from helpers import small_cfg, make_batch


def _scene():
    cfg = small_cfg('val')
    inp = make_batch(cfg, [3, 4], 3, 1500)
    return cfg, {k: (v.numpy() if hasattr(v, 'numpy') else v) for k, v in inp.items()}


def test_twin_index_structures_and_segments():
    cfg, inp = _scene()
    nx, ny, nz, nt = (int(v) for v in inp['shape'][0])
    coords, p2v = inp['coordinates'], inp['point_to_voxel_map'][:, 0]
    m, n = coords.shape[0], p2v.shape[0]
    cell, c2p = twin.cell_index(coords, nx, ny, nt, 2)
    want = ((coords[:, 0] * nt + coords[:, 4]) * ny + coords[:, 2]) * nx + coords[:, 3]
    assert np.array_equal(cell, want.astype(np.int32)) and np.array_equal(c2p[cell], np.arange(m))
    assert (c2p >= 0).sum() == m
    offs, order = twin.csr_build(p2v, m)
    assert np.array_equal(order, np.argsort(p2v, kind='stable')) and np.array_equal(np.diff(offs), np.bincount(p2v, minlength=m))
    pts = inp['input_points'].astype(np.float32)
    mean, lab = twin.segment_mean3_maxlabel(pts, inp['fb_labels'][:, 0], offs, order, m)
    assert np.array_equal(mean, oracle.segment_mean(pts, p2v.astype(np.int64), m))
    assert np.array_equal(lab, oracle.segment_max_label(inp['fb_labels'], p2v.astype(np.int64), m)[:, 0])
    src = np.random.RandomState(0).randn(n, 32).astype(np.float32)
    src[::7] = src[3::7][:src[::7].shape[0]]                                        # ties: lowest point index must win
    out, arg = twin.segment_max(src, offs, order, m)
    ro, ra = oracle.segment_max(src, p2v.astype(np.int64), m)
    assert np.array_equal(out, ro) and np.array_equal(arg, ra)
    g = np.random.RandomState(1).randn(m, 32).astype(np.float32)
    gs = twin.segment_max_backward(g, arg, p2v, n)
    want = np.where(arg[p2v] == np.arange(n)[:, None], g[p2v], 0)
    assert np.array_equal(gs, want)
    ssum = twin.segment_sum(src, offs, order, m)
    ref = np.zeros((m, 32), np.float32)
    for i in np.argsort(p2v, kind='stable'):
        ref[p2v[i]] += src[i]
    assert np.array_equal(ssum, ref)


def test_twin_pfn_features_scatter_gather():
    cfg, inp = _scene()
    vg = cfg['voxel_generator']
    nx, ny, nz, nt = (int(v) for v in inp['shape'][0])
    coords, p2v = inp['coordinates'], inp['point_to_voxel_map'][:, 0]
    m = coords.shape[0]
    pts = inp['input_points'].astype(np.float32)
    mean = oracle.segment_mean(pts, p2v.astype(np.int64), m)
    vx, vy = vg['voxel_size'][0], vg['voxel_size'][1]
    got = twin.pfn_features(pts, p2v, mean, coords, inp['time_indice'], vx, vy, vx / 2 + vg['range'][0], vy / 2 + vg['range'][1],
                            abs(vg['range'][0]), nt)
    want = oracle.pfn_features(pts, p2v.astype(np.int64), coords, mean, inp['time_indice'], vg['voxel_size'], vg['range'], nt)
    assert np.array_equal(got, want)
    cell, c2p = twin.cell_index(coords, nx, ny, nt, 2)
    feats = np.random.RandomState(2).randn(m, 8).astype(np.float32)
    canvas = twin.pillar_scatter(feats, c2p)
    ref = oracle.scatter_point_pillar(feats, coords, 2, inp['shape'][0])            # [B,C,nt,ny,nx]
    assert np.array_equal(canvas.reshape(2, nt, ny, nx, 8).transpose(0, 4, 1, 2, 3), ref)
    assert np.array_equal(twin.gather_rows(canvas, cell), feats)
    idx = np.array([2, -1, 0], np.int32)
    assert np.array_equal(twin.gather_rows(feats, idx), np.stack([feats[2], np.zeros(8, np.float32), feats[0]]))
    lab = np.arange(canvas.shape[0], dtype=np.int64)[:, None]
    assert np.array_equal(twin.gather_rows(lab, cell)[:, 0], cell.astype(np.int64))


def test_twin_bilinear_warp_transform_frames_max():
    cfg, inp = _scene()
    vg = cfg['voxel_generator']
    rng = np.random.RandomState(3)
    fmap = rng.randn(2, 16, 64, 64).astype(np.float32)                              # [B,C,H,W]
    pts = inp['input_points'].astype(np.float32)
    pts[:5, :2] = [[-8.5, 0], [8.5, 3], [0, -9], [7.99, 7.99], [-8, -8]]             # beyond / on the border
    ti = inp['time_indice']
    got = twin.bilinear_gather(fmap.transpose(0, 2, 3, 1), pts, ti[:, 0].astype(np.int32), abs(vg['range'][0]), abs(vg['range'][1]))
    want = oracle.ungrid(fmap, pts, vg['range'], ti)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    bev = rng.randn(2, 3, 8, 64, 64).astype(np.float32)                              # [B,T,C,H,W]
    pose = np.tile(np.eye(4, dtype=np.float32), (2, 3, 1, 1))
    for b in range(2):
        for t in range(1, 3):
            a = 0.02 * t * (b + 1)
            pose[b, t, :2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            pose[b, t, :2, 3] = [0.7 * t, -0.3 * b]
    inv = np.linalg.inv(pose).astype(np.float32)
    got = twin.bev_warp(bev.transpose(0, 1, 3, 4, 2), inv, vg['voxel_size'][0], vg['voxel_size'][1], vg['range'][0], vg['range'][1])
    want = oracle.warp_feats(bev, pose, vg['voxel_size'], vg['range'])
    np.testing.assert_allclose(got.transpose(0, 1, 4, 2, 3), want, rtol=0, atol=2e-5)
    assert np.array_equal(got[:, 0], bev.transpose(0, 1, 3, 4, 2)[:, 2])               # trap 1: slot 0 = last frame un-warped
    tp = twin.rigid_transform(pts, (ti[:, 0] * 3 + ti[:, 1]).astype(np.int32), pose.reshape(-1, 16))
    np.testing.assert_allclose(tp, oracle.transform_points(pts, ti, pose), rtol=0, atol=2e-6)
    x = rng.randn(2, 5, 7, 9, 4).astype(np.float32)
    x[0, 3] = x[0, 1]                                                                 # ties: lowest frame wins
    out, arg = twin.frames_max(x)
    assert np.array_equal(out, x.max(1)) and np.array_equal(arg, x.argmax(1).astype(np.uint8))
