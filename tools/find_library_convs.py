"""Round 5, verdict item 8: which calls of one training step still reach the convolution library?  Every dispatched aten op whose name contains
`convolution` / `miopen` / `cudnn` / `batch_norm`, with operand shapes and the package line that issued it.
Usage: python tools/find_library_convs.py [mixed|bf16|fp32x3] [batch]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402
from pcaccumulation_amd import distributed as pdist  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device  # noqa: E402
from pcaccumulation_amd.synthetic import make_sequence  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'mixed'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = mode
cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(batch)]
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=1.0, early_thread=False)
for _ in range(2):
    bench.train_step(stepper, batcher, scenes)
seen = collections.Counter()


class Watch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(s in name for s in ('convolution', 'miopen', 'cudnn', 'batch_norm')):
            shapes = [(tuple(a.shape), str(a.dtype).replace('torch.', '')) for a in args if torch.is_tensor(a)][:3]
            where = '?'
            for fr in reversed(traceback.extract_stack(limit=24)):
                if 'pcaccumulation_amd' in fr.filename:
                    where = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                    break
            seen['%s %s @ %s' % (name, shapes, where)] += 1
        return func(*args, **(kwargs or {}))


with Watch():
    bench.train_step(stepper, batcher, scenes)
torch.cuda.synchronize()
print('mode %s, %d sequences: %d library-convolution / batch-norm dispatches in one step' % (mode, batch, sum(seen.values())))
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print('%3d  %s' % (v, k))
