"""Golden vectors for test-mode clustering: the reference's models/cluster.py:Cluster run here on synthetic scenes
(real scikit-learn DBSCAN; torchsparse's sparse_quantize restated in ref_harness.py).  Run: python tests/golden/make_golden_cluster.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402


def synth_scene(rng, n_boxes, n_bg, n_noise, spread=40.0):
    """Moving 'vehicles' as surface samples of boxes (dense, many points per 5 cm voxel column), static clutter, and
    isolated moving-labelled points (noise / tiny clusters)."""
    pts, mos = [], []
    for _ in range(n_boxes):
        c = rng.uniform(-spread, spread, 2)
        size = rng.uniform([1.5, 3.0], [2.2, 5.5])
        yaw = rng.uniform(0, np.pi)
        n = int(rng.randint(8, 900))
        u = rng.uniform(-0.5, 0.5, (n, 2)) * size
        side = rng.randint(0, 4, n)
        u[side == 0, 0] = -size[0] / 2
        u[side == 1, 1] = size[1] / 2
        r = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        xy = u @ r.T + c
        z = rng.uniform(-1.5, 0.3, (n, 1))
        pts.append(np.concatenate([xy, z], 1))
        mos.append((rng.rand(n) < 0.93).astype(np.int64))
    pts.append(np.concatenate([rng.uniform(-spread, spread, (n_bg, 2)), rng.uniform(-2, 1, (n_bg, 1))], 1))
    mos.append(np.zeros(n_bg, np.int64))
    pts.append(np.concatenate([rng.uniform(-spread, spread, (n_noise, 2)), rng.uniform(-2, 1, (n_noise, 1))], 1))
    mos.append(np.ones(n_noise, np.int64))
    pts, mos = np.concatenate(pts).astype(np.float32), np.concatenate(mos)
    perm = rng.permutation(len(pts))
    return pts[perm], mos[perm]


def gen_cluster(save):
    ref_harness.install()
    from models.cluster import Cluster
    cfg = {'cluster': {'min_p_cluster': 15, 'voxel_size': 0.15, 'min_samples_dbscan': 5, 'cluster_metric': 'euclidean',
                       'eps_dbscan': 0.4}}
    rng = np.random.RandomState(5)
    scenes = [synth_scene(rng, 14, 3000, 60), synth_scene(rng, 1, 500, 4),     # second sample: maybe one cluster only
              synth_scene(rng, 0, 400, 9),                                       # third: <= min_p_cluster moving points -> gate
              synth_scene(rng, 25, 2000, 150, spread=25.0)]                      # crowded: touching boxes, border points
    pts = np.concatenate([s[0] for s in scenes])
    mos = np.concatenate([s[1] for s in scenes])
    batch = np.concatenate([np.full(len(s[0]), i) for i, s in enumerate(scenes)])
    offset = (rng.randn(len(pts), 2) * 0.05).astype(np.float32) * (mos[:, None] == 1)
    # exact duplicates and points on voxel-rounding ties (x/0.05 = k + 0.5)
    pts[10] = pts[3]
    offset[10] = offset[3]
    mos[10] = mos[3] = 1
    time_indice = np.stack([batch, rng.randint(0, 5, len(pts))], 1).astype(np.int64)
    out = {}
    for use_offset in (True, False):
        res = {}
        Cluster(cfg)(torch.from_numpy(pts), torch.from_numpy(mos), torch.from_numpy(offset), torch.from_numpy(time_indice),
                     res, use_offset=use_offset)
        out['labels_offset' if use_offset else 'labels_plain'] = res['inst_labels_est'].numpy()
    save('cluster', points=pts, mos=mos, offset=offset, time_indice=time_indice, eps=0.4, min_samples=5, min_p_cluster=15, **out)


if __name__ == '__main__':
    def save(name, **arrays):
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **arrays)
        print(name, {k: np.asarray(v).shape for k, v in arrays.items()})
    gen_cluster(save)
    from make_golden_model import gen_model_test
    gen_model_test(save)
