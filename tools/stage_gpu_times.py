#!/usr/bin/env python
"""GPU time per stage of one training step (events around each stage, sync'd), B sequences of 5 x 160k points. Development aid."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd import ops
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
from pcaccumulation_amd import motionnet, egomotion, stpn, alignnet, pillar_encoder, unet

B = int(os.environ.get('BATCH', '4'))
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16'); cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(B)]
times = collections.OrderedDict()

def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); r = fn(*a, **k); e.record(); torch.cuda.synchronize()
        times[name] = times.get(name, 0.0) + s.elapsed_time(e)
        return r
    return wrap

model.pillar_encoder.forward = timed('fwd pillar_encoder', model.pillar_encoder.forward)
model.unet.forward = timed('fwd unet', model.unet.forward)
model.semseg_head.forward = timed('fwd semseg_head', model.semseg_head.forward)
model.ego_feats_head.forward = timed('fwd ego_feats_head', model.ego_feats_head.forward)
model.ego_motion_head.forward_pillars = timed('fwd ego_motion_head', model.ego_motion_head.forward_pillars)
model.motionhead.backbone = timed('fwd stpn backbone', model.motionhead.backbone)
model._stpn_heads = timed('fwd stpn point heads', model._stpn_heads)
model.reconstructor.forward = timed('fwd tubenet', model.reconstructor.forward)
for it in range(4):
    times.clear()
    inp = timed('batcher (voxelise+collate)', batcher)(scenes)
    out = timed('forward total', model)(inp)
    stats = timed('loss', loss_fn)(out, inp)
    timed('backward', stats['loss'].backward)()
    def step():
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step(); opt.zero_grad(set_to_none=True)
    timed('clip+adam', step)()
sub = sum(v for k, v in times.items() if k.startswith('fwd '))
times['fwd glue (index, scatter, warp, gathers)'] = times['forward total'] - sub
for k, v in times.items():
    print('%-45s %8.2f ms' % (k, v))
