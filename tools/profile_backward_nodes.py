#!/usr/bin/env python
"""Autograd nodes of one backward pass: how often each runs and how many GPU launches / how much GPU time its kernels take.
Development aid for trimming the small-kernel tail."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = 'bf16'; cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
for it in range(2):
    bench.train_step(model, opt, loss_fn, batcher, scenes, None, 1.0)
inp = batcher(scenes); out = model(inp); stats = loss_fn(out, inp)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    stats['loss'].backward()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    if ('Backward' in e.key or e.key.startswith('autograd::engine::evaluate_function')) and e.count > 0:
        rows.append((e.device_time_total / 1e3, e.count, e.key))
seen = set()
print('autograd nodes by total GPU time of their kernels (ms), count:')
for t, n, k in sorted(rows, reverse=True)[:60]:
    name = k.replace('autograd::engine::evaluate_function: ', '')
    if name in seen:
        continue
    seen.add(name)
    print('%8.3f ms  n=%4d  %s' % (t, n, name[:80]))
print('\nnodes by count:')
for t, n, k in sorted(rows, key=lambda r: -r[1])[:30]:
    if k.startswith('autograd::engine'):
        print('%8.3f ms  n=%4d  %s' % (t, n, k.replace('autograd::engine::evaluate_function: ', '')[:80]))
