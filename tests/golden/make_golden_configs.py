#!/usr/bin/env python
"""Whole-model golden vectors at BASELINE.json's full config sizes, produced by the REFERENCE itself (imported read-only from
/root/reference on the CPU of this container, see ref_harness.py).  One fixture per config:

    c2  nuScenes geometry (z in [-5,3), freq 20, max_speed 10), T = 5,  80 k points / frame, B = 1, misc.mode = val
    c3  Waymo geometry,                                          T = 5, 160 k points / frame, B = 1, train step (loss + backward)
    c4  Waymo geometry,                                          T = 10, 200 k points / frame, B = 1, misc.mode = val
    c5  nuScenes geometry,                                       T = 5,  80 k points / frame, B = 4, train step (loss + backward)
    nus11  nuScenes configuration as the reference ships it (configs/nuscene/nuscene.yaml:7-9: 11 sweeps), 30 k points / frame, val

    python tests/golden/make_golden_configs.py c2 c3 c4 c5 nus11 c3_lidar

Inputs are regenerated from seeds by pcaccumulation_amd.synthetic (byte-stable numpy RandomState); weights are the closed-form
fill of synthetic.fill_state_dict_ plus two stored head-bias offsets (the foreground one placed in a gap of the logit
distribution, see _gap_threshold).  Stored: seeds, the reference's scalar metrics and loss
terms, 2048-point samples of the per-point outputs, digests of the integer voxel structure, per-parameter gradient norms for the
train configs.  Nothing of the reference's source is stored.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.install()
from make_golden import save, sha  # noqa: E402
from make_golden_model import _run, _common  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402

CONFIGS = {
    # name: (dataset, T, points per frame, scene seeds, mode, forward seed)
    'c2': ('nuscene', 5, 80000, (31,), 'val', 322),
    'c3': ('waymo', 5, 160000, (32,), 'train', 323),
    'c4': ('waymo', 10, 200000, (33,), 'val', 324),
    'c5': ('nuscene', 5, 80000, (34, 35, 36, 37), 'train', 325),
    'nus11': ('nuscene', 11, 30000, (38,), 'val', 326),
    # c3 on LiDAR-distributed points (1/r range density, 64 beams, scan order within a frame -- synthetic mode 'lidar_scan'):
    # crowded pillars near the sensor, ~3 points per pillar on average, long CSR segments
    'c3_lidar': ('waymo', 5, 160000, (39,), 'train', 327, 'lidar_scan'),
}


def _gap_threshold(d, lo=0.90, hi=0.97):
    """A decision threshold for the logit difference d (fg - bg) that predicts 3-10 % of the pillars foreground AND sits in the middle
    of the widest gap between neighbouring sorted values in that quantile window.  Closed-form weights shifted to a quantile put the
    boundary through a continuum of 3e5 values a few 1e-5 apart: fp32 convolutions summed in another order (oneDNN here, MIOpen on
    the GPU) then flip a handful of pillars, the background count of a frame changes by one and `torch.randperm(n)`
    (models/egomotion.py:157) draws an entirely different key-point set -- the pose of a random-weight model moves by a degree.
    A trained head is confident; the gap (typically 1e-2, reported by the caller) stands in for that."""
    v = torch.sort(d.reshape(-1).double())[0]
    a, b = int(lo * v.numel()), int(hi * v.numel())
    gaps = v[a + 1:b + 1] - v[a:b]
    j = int(torch.argmax(gaps))
    return float(0.5 * (v[a + j] + v[a + j + 1])), float(gaps[j]), 1.0 - (a + j + 1) / v.numel()


def _probe(cfg, seeds, T, ppf, fwd_seed, train, points='uniform'):
    """The two head-bias offsets (make_golden_model._tweak_biases) with a gap-seeking foreground threshold; train configs are probed
    in train() mode (BatchNorm batch statistics, as the step itself normalises), no gradients."""
    from models.motionnet import MotionNet
    from pcaccumulation_amd.synthetic import make_sequence, attach_voxels, fill_state_dict_
    cfg = dict(cfg)
    cfg['misc'] = dict(cfg['misc'], mode='train' if train else 'val')
    vox = rh.voxeliser(cfg)
    inp = rh.collate([attach_voxels(make_sequence(s, T, ppf, cfg, mode=points), vox) for s in seeds])
    model = MotionNet(cfg)
    fill_state_dict_(model)
    model.train(train)
    tweaks = {}
    with torch.no_grad():
        torch.manual_seed(fwd_seed)
        out = model(inp)
        fs = out['fb_seg_est']
        d = (fs[:, :, 1] - fs[:, :, 0])[out['occ_map'][:, :, 0] > 0]
        thr, gap, frac = _gap_threshold(d)
        print('fb threshold %.5f in a gap of %.2e, %.3f of the pillars foreground' % (thr, gap, frac), flush=True)
        tweaks['semseg_head.seg_head.3.bias'] = np.array([thr, 0.0], np.float32)
        model.semseg_head.seg_head[3].bias += torch.tensor([thr, 0.0])
        torch.manual_seed(fwd_seed)
        out = model(inp)
        m = out['mos_est']
        fb = torch.logical_or(inp['fb_labels'][:, 0] == 1, out['fb_est_per_points'][:, 0] == 1)
        med2 = float(torch.median((m[:, 1] - m[:, 0])[fb]))
        tweaks['motionhead.mos_seg.seg_head.3.bias'] = np.array([med2, 0.0], np.float32)
    return tweaks, gap


def gen(name):
    dataset, T, ppf, seeds, mode, fwd_seed = CONFIGS[name][:6]
    points = CONFIGS[name][6] if len(CONFIGS[name]) > 6 else 'uniform'
    cfg = default_config(dataset, mode, n_sweeps=T)
    t0 = time.time()
    train = mode == 'train'
    # the two head-bias offsets are found with probe forwards (_probe) and then applied to a freshly built model for the run
    tweaks, gap = _probe(cfg, seeds, T, ppf, fwd_seed, train, points)
    model, inp, out, stats, _ = _run(cfg, seeds, T, ppf, mode, fwd_seed, train=train, tweaks=tweaks, points=points)
    d, epe = _common(out, stats, inp, T)
    n = inp['input_points'].shape[0]
    idx = np.linspace(0, n - 1, 2048).astype(np.int64)
    det = lambda t: t.detach().numpy()
    extra = {}
    if train:
        names, norms = [], []
        for k, p in model.named_parameters():
            names.append(k)
            norms.append(float(p.grad.norm()) if p.grad is not None else 0.0)
        extra = dict(grad_names=np.array(names), grad_norms=np.array(norms),
                     bn_running_mean=model.semseg_head.seg_head[1].running_mean.numpy())
    fb = out['fb_est_per_points']
    save('model_%s' % name, config=name, dataset=dataset, mode=mode, seeds=np.array(seeds), n_frames=T, pts_per_frame=ppf,
         fwd_seed=fwd_seed, points=points, tweak_keys=np.array(list(tweaks.keys())), tweak_vals=np.stack(list(tweaks.values())),
         sample_idx=idx, mos_est=det(out['mos_est'])[idx], offset_est=det(out['offset_est'])[idx], rec_est=det(out['rec_est'])[idx],
         transformed_points=det(out['transformed_points'])[idx], fb_est_per_points=fb.numpy()[idx], fb_est_sum=int(fb.sum()),
         mos1_sum=int(out['mos_est'].argmax(1).sum()),
         fb_seg_est_sample=det(out['fb_seg_est'])[0, :, :, ::8, ::8], num_voxels=inp['num_voxels'].numpy(),
         coordinates_sha=sha(inp['coordinates'].numpy()), p2v_sha=sha(inp['point_to_voxel_map'].numpy()),
         epe_sample=epe.detach().numpy()[::64], fb_gap=gap, **extra, **d)
    print('%s: N=%d M=%s fg %.3f mos1 %.3f rot %.4f trans %.4f mos_iou %.4f epe %.4f loss %.4f  (%.0f s)' % (
        name, n, inp['num_voxels'].tolist(), float(fb.float().mean()), float(out['mos_est'].argmax(1).float().mean()),
        d['ego_rot_error'], d['ego_trans_error'], d['mos_iou'], d['epe_mean'], d['loss'], time.time() - t0), flush=True)


if __name__ == '__main__':
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or list(CONFIGS)):
        gen(name)
