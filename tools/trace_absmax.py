#!/usr/bin/env python
"""Which tensors of an fp32x3 training step still get a standalone absmax256 pass (no tag carried from their producer), and from where."""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd import distributed as pdist, native
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence

dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'fp32x3')
cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=cfg['train']['grad_clip'], catch=False)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
for it in range(2):
    bench.train_step(stepper, batcher, scenes)
calls = collections.Counter()
orig = native.absmax256
def spy(x):
    fr = [f for f in traceback.extract_stack()[:-1] if 'pcaccumulation_amd' in f.filename]
    where = ' < '.join('%s:%d %s' % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(fr[-4:]))
    calls[(tuple(x.shape), where)] += 1
    return orig(x)
native.absmax256 = spy
bench.train_step(stepper, batcher, scenes)
torch.cuda.synchronize()
for (shape, where), n in sorted(calls.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][0]).numel()):
    print('%3d x %-24s %s' % (n, shape, where))
