"""Dispatched (non-view) aten ops of one stage per source line of the package, or of a whole train step by op name (forward and
backward).  Development aid.  Usage: python tools/count_dispatched_ops.py tubenet|ego|loss|forward|step"""
import os, sys, collections, traceback
sys.path.insert(0, '/root/repo')
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
stage = sys.argv[1]
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = 'bf16'; cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
from pcaccumulation_amd import distributed as pdist
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=1.0)
bench.train_step(stepper, batcher, scenes)
inp = batcher(scenes)
per_line = collections.Counter()
class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        only = os.environ.get('PCACC_COUNT_ONLY')
        if only and only not in name:
            return func(*args, **(kwargs or {}))
        if only or not any(s in name for s in ('view', 'permute', 'select', 'slice', 'expand', 'unsqueeze', 'squeeze', 'detach', 'alias', 'aten.t.default', 'transpose', 'as_strided', 'size', 'stride', 'numel', 'is_', 'sym_')):
            for fr in reversed(traceback.extract_stack(limit=14)):
                if 'pcaccumulation_amd' in fr.filename:
                    per_line['%s:%d' % (os.path.basename(fr.filename), fr.lineno)] += 1
                    break
        return func(*args, **(kwargs or {}))
per_name = collections.Counter()
class CountNames(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
        per_name['%s %s' % (func, shapes)] += 1
        return func(*args, **(kwargs or {}))
if stage == 'step':
    with CountNames():
        bench.train_step(stepper, batcher, scenes)
    skip = ('view', 'permute', 'select', 'slice', 'expand', 'unsqueeze', 'squeeze', 'detach', 'alias', 'aten.t.', 'transpose', 'as_strided', 'sym_', 'reshape')
    rows = [(v, k) for k, v in per_name.items() if not any(s in k for s in skip)]
    print('total', sum(v for v, _ in rows))
    for v, k in sorted(rows, reverse=True)[:90]:
        print('%4d  %s' % (v, k[:150]))
    sys.exit(0)
target = {'tubenet': (model.reconstructor, 'forward'), 'ego': (model.ego_motion_head, 'forward_pillars'), 'forward': (model, 'forward')}[stage] if stage != 'loss' else None
if target:
    orig = getattr(*target)
    def wrapped(*a, **k):
        with Count():
            return orig(*a, **k)
    setattr(target[0], target[1], wrapped)
    out = model(inp)
else:
    out = model(inp)
    with Count():
        loss_fn(out, inp)
print('total', sum(per_line.values()))
for k2, v in per_line.most_common(50):
    print('%4d  %s' % (v, k2))
