#!/usr/bin/env python
"""The bf16 bound on TRAINED weights at BASELINE.json's config size (c3: Waymo train, 5 frames x 160 000 points, full BEV range): the
product trained in the fp32x3 mode (fp32-accurate, pinned to the reference at 1e-3 by tests/test_config_parity.py) for STEPS Adam steps of
two synthetic sequences each, then the frozen model evaluated in fp32x3 and in bf16 on SCENES held-out sequences with the same forward
seeds.  Metrics as tests/test_config_parity.py::test_gpu_bf16_against_fp32_on_trained_weights reports them for the tiny configuration:
validation-set means, median scene, worst scene, flipped foreground decisions.  Everything runs on the product path (device voxeliser).
Usage: python tools/trained_c3_bound.py [steps=400] [scenes=32] [points_mode=uniform|lidar_scan]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from pcaccumulation_amd import distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.loss import scene_flow_epe
from pcaccumulation_amd.motionnet import MotionNet
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
SCENES = int(sys.argv[2]) if len(sys.argv) > 2 else 32
MODE = sys.argv[3] if len(sys.argv) > 3 else 'uniform'
T, PPF = 5, 160000
dev = torch.device('cuda:0')


def config(dtype):
    cfg = default_config('waymo', 'train', n_sweeps=T)
    cfg['misc']['compute_dtype'] = dtype
    return cfg


cfg = config('fp32x3')
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=cfg['train']['grad_clip'], two_streams=False, pipelined=False)
scene = lambda seed: sample_to_device(make_sequence(seed, T, PPF, cfg, mode=MODE), dev)
t0 = time.time()
losses = []
for step in range(STEPS):
    stats = bench.train_step(stepper, batcher, [scene(1000 + 2 * step), scene(1001 + 2 * step)])
    if stats is not None and 'loss' in stats:
        losses.append(float(stats['loss'].detach()) if torch.is_tensor(stats['loss']) else float(stats['loss']))
torch.cuda.synchronize()
print('trained %d steps of 2 sequences in %.0f s; loss first / last five: %s / %s' % (STEPS, time.time() - t0, np.round(losses[:5], 3).tolist(),
                                                                                    np.round(losses[-5:], 3).tolist()), flush=True)

model.eval()
cfg16 = config('bf16')
model16 = MotionNet(cfg16).to(dev).channels_last_()
model16.load_state_dict(model.state_dict())
model16.eval()
batcher16 = DeviceBatcher(cfg16)                              # the batcher prepares dtype-specific inputs
rows = {'fp32x3': [], 'bf16': []}
for seed in range(5000, 5000 + SCENES):
    for tag, m, bt in (('fp32x3', model, batcher), ('bf16', model16, batcher16)):
        inp = bt([scene(seed)])                                # a device sample belongs to the batcher that consumed it: one per model
        torch.manual_seed(seed)
        with torch.no_grad():
            out = m(inp)
            st = loss_fn(out, inp)
        i, u = st['mos_metric']['intersection'], st['mos_metric']['union']
        sel = inp['time_indice'][:, 0] == 0
        s0 = {k: inp[k][sel] for k in ('input_points', 'time_indice', 'inst_labels')}
        s0['input_points'] = s0['input_points'].float()
        s0['ego_motion_gt'], s0['inst_motion_gt'] = inp['ego_motion_gt'], inp['inst_motion_gt']
        epe = scene_flow_epe({'rec_est': out['rec_est'][sel].detach()}, s0, T)
        rows[tag].append(dict(rot=float(out['ego_rot_error']), trans=float(out['ego_trans_error']), epe=float(epe.mean()),
                              mos_iou=float(np.mean(np.array(i, dtype=np.float64) / (np.array(u, dtype=np.float64) + 1e-20))), mos_i=np.array(i, dtype=np.float64), mos_u=np.array(u, dtype=np.float64), fb=out['fb_est_per_points'].clone()))
per_scene = lambda k: np.array([abs(a[k] - b[k]) for a, b in zip(rows['fp32x3'], rows['bf16'])])
set_mean = lambda k: abs(float(np.mean([a[k] for a in rows['fp32x3']])) - float(np.mean([b[k] for b in rows['bf16']])))
flips = np.array([float((a['fb'] != b['fb']).float().mean()) for a, b in zip(rows['fp32x3'], rows['bf16'])])
agg = {t: float(np.mean(sum(r['mos_i'] for r in rows[t]) / (sum(r['mos_u'] for r in rows[t]) + 1e-20))) for t in rows}
res = {'config': 'c3 size: waymo train, %d x %d points (%s), %d training steps of 2 sequences in fp32x3, %d held-out scenes' % (T, PPF, MODE, STEPS, SCENES),
       'fp32x3_level': {k: float(np.mean([r[k] for r in rows['fp32x3']])) for k in ('rot', 'trans', 'epe')} | {'mos_iou': agg['fp32x3']},
       'set_mean_abs_diff': {'rot_deg': set_mean('rot'), 'trans_m': set_mean('trans'), 'epe_m': set_mean('epe'), 'mos_iou': abs(agg['fp32x3'] - agg['bf16'])},
       'median_scene_abs_diff': {'rot_deg': float(np.median(per_scene('rot'))), 'trans_m': float(np.median(per_scene('trans'))),
                                 'epe_m': float(np.median(per_scene('epe'))), 'mos_iou': float(np.median(per_scene('mos_iou'))),
                                 'fb_flips': float(np.median(flips))},
       'worst_scene_abs_diff': {'rot_deg': float(per_scene('rot').max()), 'trans_m': float(per_scene('trans').max()), 'epe_m': float(per_scene('epe').max()),
                                'mos_iou': float(per_scene('mos_iou').max()), 'fb_flips': float(flips.max())}}
print(json.dumps(res, indent=1))
