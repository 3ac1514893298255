"""Source lines of the package that dispatch aten copy / index / cat / arithmetic ops on tensors of at least MIN elements during one
training step (forward + backward of the custom Functions; built-in autograd nodes have no Python frame).  Development aid.
Usage: python tools/where_big_copies.py [min_elements=200000]"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from pcaccumulation_amd import distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
MIN = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16'); cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=1.0, two_streams=False)
for _ in range(2):
    bench.train_step(stepper, batcher, scenes)
rows = collections.Counter()
SKIP = ('view', 'permute', 'select', 'slice', 'expand', 'unsqueeze', 'squeeze', 'detach', 'alias', 'aten.t.default', 'transpose', 'as_strided', 'sym_',
        'empty', 'reshape', '_unsafe_view', 'split', 'unbind', 'narrow', 'lift_fresh', 'is_', 'stride', 'size', 'numel')
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(s in name for s in SKIP):
            return out
        n = max([a.numel() for a in args if torch.is_tensor(a)] + [out.numel() if torch.is_tensor(out) else 0])
        if n >= MIN:
            where = 'autograd (no python frame)'
            for fr in reversed(traceback.extract_stack(limit=18)):
                if 'pcaccumulation_amd' in fr.filename or fr.filename.endswith('bench.py'):
                    where = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                    break
            shape = next((tuple(a.shape) for a in args if torch.is_tensor(a) and a.numel() == n), tuple(out.shape) if torch.is_tensor(out) else ())
            rows[(where, name.replace('aten.', ''), shape)] += 1
        return out
with Mode():
    bench.train_step(stepper, batcher, scenes)
torch.cuda.synchronize()
for (where, name, shape), c in sorted(rows.items(), key=lambda kv: -kv[1] * max(1, torch.Size(kv[0][2]).numel())):
    print('%3d x %-34s %-28s %s' % (c, name[:34], str(shape)[:28], where))
