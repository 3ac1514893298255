#!/usr/bin/env python
"""Library GEMM calls of one training step with their shapes and GPU time.  Development aid."""
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
    inp = batcher(scenes); out = model(inp); stats = loss_fn(out, inp); stats['loss'].backward(); opt.zero_grad(set_to_none=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    inp = batcher(scenes); out = model(inp); stats = loss_fn(out, inp); stats['loss'].backward()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ('aten::mm', 'aten::bmm', 'aten::addmm', 'aten::baddbmm', 'aten::matmul', 'aten::linear', 'aten::_scaled_mm') and e.device_time_total > 0:
        rows.append((e.device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:150]))
for r in sorted(rows, reverse=True)[:25]:
    print('%8.3f ms  n=%3d  %-12s %s' % r)
