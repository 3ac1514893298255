#!/usr/bin/env python
"""Eval-only forward throughput of the hot path for the other BASELINE.json configs (not the contract bench):
c2 = 5 x 80 k points bf16 eval forward; c4 = 10 x 200 k points, bf16 convs + fp32 ego head.  Prints one JSON line each."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.motionnet import MotionNet
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence


def run(name, T, ppf, batch, steps=10, warmup=3):
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'val', n_sweeps=T)
    cfg['misc']['compute_dtype'] = 'bf16'
    cfg['pose_estimation']['kpt_sampler'] = 'device'
    torch.manual_seed(0)
    model = MotionNet(cfg).to(dev).channels_last_().eval()
    batcher = DeviceBatcher(cfg)
    scenes = [sample_to_device(make_sequence(i, T, ppf, cfg), dev) for i in range(batch)]
    with torch.no_grad():
        for _ in range(warmup):
            model(batcher(scenes))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(batcher(scenes))
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({'config': name, 'frames': T, 'pts_per_frame': ppf, 'sequences_per_step': batch, 'mode': 'eval forward (voxelise + MotionNet)',
                      'dtype': 'bf16', 'ms_per_step': dt * 1e3, 'lidar_frames_per_s': batch * T / dt}))


if __name__ == '__main__':
    run('c2: 5 x 80k, eval forward', 5, 80000, 4)
    run('c3 shape: 5 x 160k, eval forward', 5, 160000, 4)
    run('c4: 10 x 200k, eval forward', 10, 200000, 2)
