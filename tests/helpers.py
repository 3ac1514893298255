"""Shared builders for the parity tests (inputs are regenerated from seeds, never read from /root/reference)."""
import os

import numpy as np

import oracle
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.dataloader import collate_fn
from pcaccumulation_amd.synthetic import make_sequence, attach_voxels


def oracle_voxeliser(cfg):
    vg = cfg['voxel_generator']

    def f(points):
        out = oracle.voxelize(points, vg['voxel_size'], vg['range'], vg['n_sweeps'])
        out.pop('num_points_per_voxel')
        return out
    return f


def small_cfg(mode='val'):
    return default_config('waymo', mode, n_sweeps=3, xy_range=8)


def make_batch(cfg, seeds, n_frames, pts_per_frame, mode='uniform', voxeliser=None):
    vox = voxeliser or oracle_voxeliser(cfg)
    samples = [attach_voxels(make_sequence(s, n_frames, pts_per_frame, cfg, mode=mode), vox) for s in seeds]
    return collate_fn(samples)


def vox_points(seed, n, cfg, frac_out=0.1):
    """Same generator as tests/golden/make_golden.py:vox_points."""
    rng = np.random.RandomState(seed)
    r = np.asarray(cfg['voxel_generator']['range'], np.float64)
    T = cfg['voxel_generator']['n_sweeps']
    lo, hi = r[:3], r[3:]
    span = hi - lo
    p = lo + rng.uniform(-frac_out / 2, 1 + frac_out / 2, (n, 3)) * span
    t = rng.randint(0, T, n)
    edge = rng.randint(0, n, 32)
    p[edge[:8], 0] = lo[0]
    p[edge[8:16], 0] = hi[0]
    p[edge[16:24], 1] = lo[1] + 0.25 * rng.randint(0, int(span[1] / 0.25), 8)
    p[edge[24:], 2] = hi[2]
    return np.concatenate([p, t[:, None]], axis=1).astype(np.float32)


def raw_sample(seed, n_frames, ppf, cfg):
    """A raw sample as BaseDataset.__getitem__ reads it from disk (libs/dataset.py:206-214): float64 points with some beyond
    the crop box, below the ground threshold and above crop_z_max; per-point labels and frame index as 1-D arrays."""
    import numpy as np
    from pcaccumulation_amd.synthetic import make_sequence
    s = make_sequence(seed, n_frames, ppf, cfg)
    pts = s['input_points'].astype(np.float64)
    pts[::7] *= 1.6
    pts[::5, 2] -= 1.0
    pts[3::11, 2] += 9.0
    return {'raw_points': pts, 'time_indice': s['time_indice'][:, 0].astype(np.float64), 'sd_labels': s['sd_labels'][:, 0],
            'fb_labels': s['fb_labels'][:, 0], 'inst_labels': s['inst_labels'][:, 0], 'ego_motion_gt': s['ego_motion_gt'].astype(np.float64),
            'inst_motion_gt': s['inst_motion_gt'].astype(np.float64)}


def flow_error_scenes(root, n_scenes=3, seed=0):
    """A results folder as SegTrainer.test leaves it (libs/tester.py:95-107): <root>/<scene>/flow_error.npz with the reference's
    keys and narrow dtypes; errors spread around the 0.05 / 0.1 / 0.3 thresholds; the last scene has no moving point and no
    static foreground, and stores its frame index run-length coded ('length', toolbox/evaluation.py:37-46)."""
    import os
    rng = np.random.RandomState(seed)
    for s in range(n_scenes):
        n = 4000 + 777 * s
        t = np.sort(rng.randint(1, 5, n)).astype(np.int8)
        fb = rng.rand(n) < 0.2
        sd = fb & (rng.rand(n) < 0.5)
        if s == n_scenes - 1:
            fb[:] = False
            sd[:] = False
        epe = np.abs(rng.randn(n) * 0.15).astype(np.float16)
        rel = np.abs(rng.randn(n) * 0.2).astype(np.float16)
        d = os.path.join(root, 'scene_%03d' % s)
        os.makedirs(d, exist_ok=True)
        if s == n_scenes - 1:
            vals, counts = np.unique(t, return_counts=True)
            np.savez_compressed(os.path.join(d, 'flow_error'), fb_label=fb, sd_label=sd, epe_per_point=epe, relative_error=rel,
                                time_indice=vals.astype(np.int8), length=counts.astype(np.int64))
        else:
            np.savez_compressed(os.path.join(d, 'flow_error'), fb_label=fb, sd_label=sd, epe_per_point=epe, relative_error=rel, time_indice=t)
