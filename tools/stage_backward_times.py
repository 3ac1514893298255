#!/usr/bin/env python
"""Backward GPU time and kernel-launch count of each loss term on its own (retain_graph), plus forward launch counts per stage:
shows which branch of the model owns the small-kernel tail.  Development aid."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
from torch.profiler import profile, ProfilerActivity

B = int(os.environ.get('BATCH', '4'))
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16'); cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(B)]


def kernels(fn):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        r = fn()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    return r, len(evs), sum(e.device_time for e in evs) / 1e3


for it in range(2):
    inp = batcher(scenes)
    out = model(inp)
    stats = loss_fn(out, inp)
    stats['loss'].backward()
    opt.zero_grad(set_to_none=True)
inp = batcher(scenes)
stage = collections.OrderedDict()
def wrap(name, fn):
    def w(*a, **k):
        r, n, t = kernels(lambda: fn(*a, **k))
        stage[name] = (n, t)
        return r
    return w
model.pillar_encoder.forward = wrap('fwd pillar_encoder', model.pillar_encoder.forward)
model.unet.forward = wrap('fwd unet', model.unet.forward)
model.ego_motion_head.forward_pillars = wrap('fwd ego_motion_head', model.ego_motion_head.forward_pillars)
model.motionhead.backbone = wrap('fwd stpn backbone', model.motionhead.backbone)
model._stpn_heads = wrap('fwd stpn point heads', model._stpn_heads)
model.reconstructor.forward = wrap('fwd tubenet', model.reconstructor.forward)
out = model(inp)
stats, n, t = kernels(lambda: loss_fn(out, inp))
stage['loss'] = (n, t)
for k, (n, t) in stage.items():
    print('%-28s %6d launches %8.2f ms GPU' % (k, n, t))
for key in ('fb_loss', 'mos_loss', 'offset_loss', 'obj_loss', 'perm_loss', 'ego_l1_loss', 'loss'):
    if key not in stats or not torch.is_tensor(stats[key]) or not stats[key].requires_grad:
        print('bwd %-24s -' % key)
        continue
    _, n, t = kernels(lambda: stats[key].backward(retain_graph=True))
    print('bwd %-24s %6d launches %8.2f ms GPU' % (key, n, t))
    opt.zero_grad(set_to_none=True)
