#!/usr/bin/env python
"""Every call into libpcacc_hip.so during one training step: shapes, event-timed duration, bytes of its tensor arguments and
results (an upper bound of the algorithmic traffic: every operand once) and the resulting GB/s.  Shows which calls sit far from
the 8 TB/s HBM roofline at their real shapes.  Each call is timed alone between two events; a spin kernel queued in front of the first
event lets the host finish issuing the call (allocations, ctypes, several launches) before the GPU reaches it, so the interval is the
call's kernels back to back and not the host's issue latency (about 25 us per call before round 3's end, which put a 48 us kernel at
77 us).  Development aid.  Usage: [PCACC_DTYPE=bf16|fp32|fp32x3] [PCACC_POINTS=lidar] python tools/native_call_table.py [min_us]"""
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
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg, mode='lidar_scan' if os.environ.get('PCACC_POINTS') == 'lidar' else 'uniform'), dev) for i in range(4)]
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


depth = [0]


def wrap(name, fn):
    def w(*a, **k):
        if depth[0]:                                            # a call made by another native wrapper (bilinear backward -> csr_build): part of the outer call
            return fn(*a, **k)
        depth[0] += 1
        try:
            return timed(name, fn, a, k)
        finally:
            depth[0] -= 1
    return w


def timed(name, fn, a, k):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(600_000)                              # ~0.3 ms of spinning: the host issues the whole call behind it
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
print('%d native calls, %.2f ms in total (each timed alone behind a spin kernel: kernel time, no host issue latency)' % (len(calls), sum(c[2] for c in calls) / 1e3))
print('%8s %4s %8s %8s %7s  %s' % ('tot us', 'n', 'avg us', 'MB', 'GB/s', 'call'))
for (name, desc), (n, us, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us / n < min_us:
        continue
    print('%8.0f %4d %8.1f %8.1f %7.0f  %s %s' % (us, n, us / n, nb / 1e6, nb / (us / n) / 1e3, name, desc[:150]))
