#!/usr/bin/env python
"""Every call into libpcacc_hip.so during one training step: shapes, event-timed duration, bytes of its tensor arguments and
results (an upper bound of the algorithmic traffic: every operand once) and the resulting GB/s.  Shows which calls sit far from
the 8 TB/s HBM roofline at their real shapes.  Development aid.  Usage: [PCACC_DTYPE=bf16|fp32|fp32x3] python tools/native_call_table.py [min_us]"""
import os, sys, collections, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pcaccumulation_amd import native, distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence

min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16'); cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=1.0)
for _ in range(3):
    bench.train_step(stepper, batcher, scenes)

calls = []


def tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from tensors(o)


def wrap(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **k)
        e.record()
        torch.cuda.synchronize()
        ins = list(tensors(a)) + list(tensors(list(k.values())))
        outs = list(tensors(r))
        nbytes = sum(t.numel() * t.element_size() for t in ins + outs)
        desc = ' '.join('%s%s' % (str(t.dtype).replace('torch.', '')[:4], tuple(t.shape)) for t in ins[:4])
        calls.append((name, desc, s.elapsed_time(e) * 1e3, nbytes))
        return r
    return w


skip = {'lib', 'upload_small'}
for name, fn in list(vars(native).items()):
    if isinstance(fn, types.FunctionType) and not name.startswith('_') and name not in skip and not name.endswith('_supported'):
        setattr(native, name, wrap(name, fn))
bench.train_step(stepper, batcher, scenes)
torch.cuda.synchronize()

agg = collections.OrderedDict()
for name, desc, us, nb in calls:
    key = (name, desc)
    a = agg.setdefault(key, [0, 0.0, nb])
    a[0] += 1
    a[1] += us
print('%d native calls, %.2f ms in total (each timed alone, queue drained)' % (len(calls), sum(c[2] for c in calls) / 1e3))
print('%8s %4s %8s %8s %7s  %s' % ('tot us', 'n', 'avg us', 'MB', 'GB/s', 'call'))
for (name, desc), (n, us, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us / n < min_us:
        continue
    print('%8.0f %4d %8.1f %8.1f %7.0f  %s %s' % (us, n, us / n, nb / 1e6, nb / (us / n) / 1e3, name, desc[:150]))
