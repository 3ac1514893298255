#!/usr/bin/env python
"""aten-level ops (count, GPU time) inside one stage's forward.  Usage: profile_stage_ops.py loss|tubenet|ego   Development aid."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
from torch.profiler import profile, ProfilerActivity

stage = sys.argv[1]
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = 'bf16'; cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
for it in range(2):
    inp = batcher(scenes)
    loss_fn(model(inp), inp)['loss'].backward()
    opt.zero_grad(set_to_none=True)


def report(prof):
    rows = []
    for e in prof.key_averages(group_by_stack_n=0):
        if e.self_device_time_total > 0 or e.count > 20:
            rows.append((e.self_device_time_total / 1e3, e.count, e.key))
    print('launch-level total %.2f ms' % sum(r[0] for r in rows))
    for r in sorted(rows, reverse=True)[:45]:
        print('%8.3f ms  n=%4d  %s' % (r[0], r[1], r[2][:120]))


inp = batcher(scenes)
if stage == 'loss':
    out = model(inp)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        stats = loss_fn(out, inp)
        torch.cuda.synchronize()
    report(prof)
else:
    target = model.ego_motion_head if stage == 'ego' else model.reconstructor
    name = 'forward_pillars' if stage == 'ego' else 'forward'
    orig = getattr(target, name)
    def wrapped(*a, **k):
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            r = orig(*a, **k)
            torch.cuda.synchronize()
        report(prof)
        return r
    setattr(target, name, wrapped)
    out = model(inp)
