#!/usr/bin/env python
"""Top GPU kernels (time, launches) of one stage's forward and of one loss term's backward.  Usage: profile_branch.py <stage> <loss_key>
stage in {ego, tubenet}.  Development aid."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
from torch.profiler import profile, ProfilerActivity

stage, key = sys.argv[1], sys.argv[2]
B = int(os.environ.get('BATCH', '4'))
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = 'bf16'; cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(B)]


def show(prof, title, n=22):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            agg[e.name][0] += 1
            agg[e.name][1] += e.device_time / 1e3
    tot_n, tot_t = sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())
    print('== %s: %d launches, %.2f ms' % (title, tot_n, tot_t))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
        print('   %6d  %8.3f ms  %s' % (v[0], v[1], k[:110]))


for it in range(2):
    inp = batcher(scenes); out = model(inp); stats = loss_fn(out, inp); stats['loss'].backward(); opt.zero_grad(set_to_none=True)
inp = batcher(scenes)
target = model.ego_motion_head if stage == 'ego' else model.reconstructor
name = 'forward_pillars' if stage == 'ego' else 'forward'
orig = getattr(target, name)
def wrapped(*a, **k):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        r = orig(*a, **k)
        torch.cuda.synchronize()
    show(prof, 'forward ' + stage)
    return r
setattr(target, name, wrapped)
out = model(inp)
stats = loss_fn(out, inp)
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    stats[key].backward(retain_graph=True)
    torch.cuda.synchronize()
show(prof, 'backward ' + key, 30)
