#!/usr/bin/env python
"""aten ops of one training step by GPU time, grouped by input shapes (finds the avoidable elementwise passes).  Development aid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16'); cfg['pose_estimation']['kpt_sampler'] = 'device'
from pcaccumulation_amd import distributed as pdist
model, opt, loss_fn = bench.build(cfg, dev)
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=cfg['train']['grad_clip'], catch=False)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
for it in range(3):
    bench.train_step(stepper, batcher, scenes)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.train_step(stepper, batcher, scenes)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.self_device_time_total > 0:
        rows.append((e.self_device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:110]))
tot = sum(r[0] for r in rows)
print('total self device time %.2f ms' % tot)
for r in sorted(rows, reverse=True)[:int(sys.argv[1]) if len(sys.argv) > 1 else 60]:
    print('%8.3f ms  n=%4d  %-38s %s' % r)
