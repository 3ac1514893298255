"""Time the device-side data step (crop / ground removal / augmentation / voxelisation, include/pcacc.h D1 + A1) against the
numpy path it replaces (oracle.prep_points + the oracle voxeliser) on one 5 x 160 k-point sample.
Usage: python tools/bench_prep.py [--points 160000] [--frames 5] [--iters 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import raw_sample  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.dataset import PrepInput  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=160000)
    ap.add_argument('--frames', type=int, default=5)
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    cfg = default_config('waymo', 'train', n_sweeps=a.frames)
    raw = raw_sample(0, a.frames, a.points, cfg)
    dev = torch.device('cuda:0')
    t = lambda x: torch.from_numpy(x).to(dev)
    args = (t(raw['raw_points']), t(raw['sd_labels']), t(raw['fb_labels']), t(raw['inst_labels']), t(raw['time_indice']),
            raw['ego_motion_gt'], raw['inst_motion_gt'])
    out = {'points': int(raw['raw_points'].shape[0])}
    for rng in ('device', 'reference'):
        prep = PrepInput(cfg, augmentation=True, rng=rng)
        d = prep(*args)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.iters):
            d = prep(*args)
        torch.cuda.synchronize()
        out['gpu_ms_rng_%s' % rng] = round((time.time() - t0) / a.iters * 1e3, 3)
    out['kept'] = int(d['input_points'].shape[0])
    import oracle
    vg, dd = cfg['voxel_generator'], cfg['data']
    p = dict(cfg['data_aug'], crop_xy=vg['crop_range'][0], crop_z_min=vg['crop_range'][1], crop_z_max=vg['crop_range'][2],
             remove_ground=dd['remove_ground'], ground_height=dd['ground_height'] + dd['ground_slack'], n_frames=a.frames)
    t0 = time.time()
    h = oracle.prep_points(raw['raw_points'], raw['sd_labels'], raw['fb_labels'], raw['inst_labels'], raw['time_indice'],
                           raw['ego_motion_gt'], raw['inst_motion_gt'], p, True)
    t1 = time.time()
    pts4 = np.concatenate((h['input_points'], h['time_indice']), axis=1).astype(np.float32)
    oracle.voxelize(pts4, vg['voxel_size'], vg['range'], vg['n_sweeps'])
    out['host_ms_numpy_prep'] = round((t1 - t0) * 1e3, 1)
    out['host_ms_voxelise_c'] = round((time.time() - t1) * 1e3, 1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
