#!/usr/bin/env python
"""Where the torch (non-libpcacc) kernels of one training step come from: every aten op with device time, keyed by the innermost
pcaccumulation_amd / bench.py source line on its Python stack (ops run by built-in autograd nodes have no Python stack: they are keyed by
the node's name and input shapes).  Development aid for VERDICT r02 item 3 (torch_misc).  Usage: profile_torch_tail.py [rows=80]"""
import collections
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd import distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16')
cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=cfg['train']['grad_clip'], catch=False)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
# Python stacks are not recorded on this build (with_stack gives empty lists): forward ops are attributed to named ranges instead -- every module
# down to depth 3, the loss terms, the model's own phase methods and the batcher
from torch.profiler import record_function


def _ranged(fn, name):
    def wrapped(*a, **k):
        with record_function('mod:' + name):
            return fn(*a, **k)
    return wrapped


for name, m in model.named_modules():
    if name and name.count('.') <= 2:
        m.forward = _ranged(m.forward, name)
for obj, label in ((loss_fn, 'loss'), (model, 'model'), (batcher, 'batcher'), (model.ego_motion_head, 'ego'), (getattr(model, 'align_net', None), 'align')):
    if obj is None:
        continue
    for attr in dir(obj):
        if attr.startswith('__') or attr in ('forward', 'train', 'eval', 'to', 'cuda', 'cpu', 'float', 'half', 'double', 'type', 'apply', 'zero_grad', 'requires_grad_'):
            continue
        try:
            fn = getattr(obj, attr)
        except Exception:
            continue
        if callable(fn) and getattr(fn, '__self__', None) is obj and getattr(fn, '__func__', None) is not None and 'pcaccumulation_amd' in (getattr(fn.__func__, '__module__', '') or ''):
            try:
                setattr(obj, attr, _ranged(fn, label + '.' + attr))
            except Exception:
                pass
for it in range(3):
    bench.train_step(stepper, batcher, scenes)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    bench.train_step(stepper, batcher, scenes)
    torch.cuda.synchronize()

NATIVE = ('_Conv3x3', '_Rows', '_Pfn', '_Segment', '_Bilinear', '_BatchNormRows', '_Sinkhorn', '_SegLoss', '_Head', '_UpConv', '_Pool', '_Frames',
          '_Offset', '_Tube', '_Kabsch', '_Chamfer', '_Cluster', '_Scatter', '_BevWarp', '_Canvas')
agg = collections.OrderedDict()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or e.self_device_time_total <= 0:
        continue
    if e.key.startswith(NATIVE):
        continue                       # a libpcacc call: its kernels are the product's own
    where = None
    for fr in (e.stack or []):
        if 'pcaccumulation_amd/' in fr or 'bench.py' in fr:
            where = fr.split('pcaccumulation_amd/')[-1].strip()
            break
    if where is None:
        par, depth = e.cpu_parent, 0
        while par is not None and depth < 40:
            if 'Backward' in par.key or 'evaluate_function' in par.key:
                where = par.key.replace('autograd::engine::evaluate_function: ', 'autograd node ')
                break
            if par.key.startswith('mod:'):
                where = par.key
                break
            par, depth = par.cpu_parent, depth + 1
        where = where or '(no python frame)'
    key = (where[:90], e.key, str(e.input_shapes)[:70])
    d = agg.setdefault(key, [0.0, 0])
    d[0] += e.self_device_time_total / 1e3
    d[1] += 1
tot = sum(v[0] for v in agg.values())
print('torch-side ops with device time: %.2f ms, %d op calls' % (tot, sum(v[1] for v in agg.values())))
by_where = collections.OrderedDict()
for (w, k, s), v in agg.items():
    d = by_where.setdefault(w, [0.0, 0])
    d[0] += v[0]
    d[1] += v[1]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
print('--- by source line')
for w, v in sorted(by_where.items(), key=lambda x: -x[1][0])[:n]:
    print('%7.3f ms n=%4d  %s' % (v[0], v[1], w))
print('--- by (line, op, shapes)')
for (w, k, s), v in sorted(agg.items(), key=lambda x: -x[1][0])[:n]:
    print('%7.3f ms n=%4d  %-28s %-60s %s' % (v[0], v[1], k[:28], w[:60], s))
