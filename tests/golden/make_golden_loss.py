"""Golden vectors for the loss row: the reference's own FuseLoss.get_seg_loss (libs/loss.py:110-137:
weighted cross entropy + Lovasz-Softmax + IoU counters) and FuseLoss.get_offset_loss (libs/loss.py:194-250) run on CPU on
seeded inputs, with the gradients its autograd gives.  Run: python tests/golden/make_golden_loss.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402


def _loss_fn():
    ref_harness.install()
    import libs.loss as L
    cfg = dict(default_config('waymo', 'train', n_sweeps=3)['loss'], save_dir='/tmp', min_p_cluster=15)
    try:
        return L.FuseLoss(cfg)
    except Exception:                                      # ClusterEvaluation wants extra keys; not part of this row
        L.ClusterEvaluation = lambda c: None
        return L.FuseLoss(cfg)


def seg_cases():
    """name -> (logits [n,2] f32, labels [n] i64).  Logit scales from soft to saturated (ties at error 0 / 1), labels with
    and without the ignore value, one class absent, a single row."""
    rng = np.random.RandomState(11)
    cases = {}
    z = (rng.randn(6000, 2) * 2).astype(np.float32)
    y = (rng.rand(6000) < 0.2).astype(np.int64)
    cases['mixed'] = (z, y)
    y2 = y.copy()
    y2[rng.rand(6000) < 0.1] = -1
    cases['ignore'] = (z, y2)
    cases['saturated'] = ((rng.randn(3000, 2) * 40).astype(np.float32), (rng.rand(3000) < 0.5).astype(np.int64))
    cases['only_bg'] = ((rng.randn(500, 2)).astype(np.float32), np.zeros(500, dtype=np.int64))
    cases['only_fg'] = ((rng.randn(500, 2)).astype(np.float32), np.ones(500, dtype=np.int64))
    cases['one_row'] = (np.array([[0.3, -0.2]], dtype=np.float32), np.array([1], dtype=np.int64))
    cases['rare'] = ((rng.randn(4097, 2) * 3).astype(np.float32), (rng.rand(4097) < 0.0005).astype(np.int64))
    return cases


def offset_case(seed, n_samples=2, n_frames=3, pts=900):
    """Random scenes: per sample a few instances with rigid motions per frame, ego poses, estimated offsets."""
    rng = np.random.RandomState(seed)

    def pose(scale):
        from scipy.spatial.transform import Rotation as R
        m = np.eye(4)
        m[:3, :3] = R.from_euler('z', rng.randn() * scale).as_matrix()
        m[:3, 3] = rng.randn(3) * scale * 4
        return m
    points, tidx, inst, fb, bbox = [], [], [], [], []
    for b in range(n_samples):
        n_inst = 3 + b
        n = pts + 37 * b
        points.append(rng.randn(n, 3) * 10)
        t = rng.randint(0, n_frames, n)
        tidx.append(np.stack([np.full(n, b), t], 1))
        lab = np.where(rng.rand(n) < 0.3, rng.randint(1, n_inst + 1, n), 0)
        lab[:n_inst + 1] = np.arange(n_inst + 1)                  # every label present (the reference asserts max+1 rows)
        inst.append(lab)
        fb.append((lab > 0).astype(np.int64))
        m = np.stack([np.stack([pose(0.1) if i > 0 else np.eye(4) for _ in range(n_frames)]) for i in range(n_inst + 1)])
        bbox.append(m.astype(np.float32))
    ego = np.stack([np.stack([pose(0.05) for _ in range(n_frames)]) for _ in range(n_samples)]).astype(np.float32)
    points = np.concatenate(points).astype(np.float32)
    n = points.shape[0]
    d = {'input_points': points, 'time_indice': np.concatenate(tidx).astype(np.int64), 'inst_labels': np.concatenate(inst)[:, None].astype(np.int64),
         'fb_labels': np.concatenate(fb)[:, None], 'ego_motion_gt': ego,
         'transformed_points': (points + rng.randn(n, 3) * 0.05).astype(np.float32), 'offset_est': (rng.randn(n, 2) * 2).astype(np.float32)}
    d['offset_est'][5] = 0                                                # a zero estimate: norm backward at 0
    for b, m in enumerate(bbox):
        d['inst_motion_gt_%d' % b] = m
    d['n_samples'] = n_samples
    return d


def gen(save):
    loss_fn = _loss_fn()
    out = {}
    for name, (z, y) in seg_cases().items():
        est = torch.from_numpy(z).requires_grad_(True)
        stats = loss_fn.get_seg_loss(torch.from_numpy(y), est)
        g_bce, = torch.autograd.grad(stats['bce_loss'], est, retain_graph=True)
        g_lov, = torch.autograd.grad(stats['lovasz_loss'], est)
        out.update({'seg_%s_logits' % name: z, 'seg_%s_labels' % name: y, 'seg_%s_bce' % name: stats['bce_loss'].item(),
                    'seg_%s_lovasz' % name: stats['lovasz_loss'].item(), 'seg_%s_grad_bce' % name: g_bce.numpy(),
                    'seg_%s_grad_lovasz' % name: g_lov.numpy(),
                    'seg_%s_metric' % name: np.stack([stats['metric'][k] for k in ('intersection', 'union', 'pred_positives', 'gt_positives')])})
        print(name, stats['bce_loss'].item(), stats['lovasz_loss'].item(), stats['metric'])
    out['seg_names'] = np.array(list(seg_cases().keys()))
    for ci, seed in enumerate((3, 4)):
        d = offset_case(seed)
        inp = {k: torch.from_numpy(d[k]) for k in ('input_points', 'time_indice', 'inst_labels', 'fb_labels', 'ego_motion_gt')}
        inp['inst_motion_gt'] = [torch.from_numpy(d['inst_motion_gt_%d' % b]) for b in range(d['n_samples'])]
        est = torch.from_numpy(d['offset_est']).requires_grad_(True)
        pred = {'transformed_points': torch.from_numpy(d['transformed_points']), 'offset_est': est}
        o_norm, o_dir, o_l2 = loss_fn.get_offset_loss(inp, pred)
        g_norm, = torch.autograd.grad(o_norm, est, retain_graph=True)
        g_dir, = torch.autograd.grad(o_dir, est)
        for k, v in d.items():
            out['off%d_%s' % (ci, k)] = v
        out.update({'off%d_norm' % ci: o_norm.item(), 'off%d_dir' % ci: o_dir.item(), 'off%d_l2' % ci: o_l2,
                    'off%d_grad_norm' % ci: g_norm.numpy(), 'off%d_grad_dir' % ci: g_dir.numpy(), 'off%d_offset_gt' % ci: pred['offset_gt'].numpy()})
        print('offset', ci, o_norm.item(), o_dir.item(), o_l2)
    save('loss', **out)


if __name__ == '__main__':
    def save(name, **arrays):
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **arrays)
    gen(save)
