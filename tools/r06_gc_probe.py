"""[r6] What does one staged training step leave to the CYCLIC garbage collector?  (Freezing / disabling the collector made the step 2 ms slower, profiles/r06_gc_ab.txt:
if reference cycles hold device tensors, only a collection returns their memory.)  Runs a few steps with the collector off, then collects with DEBUG_SAVEALL and lists
what was unreachable: object types, tensors, bytes, and for the largest tensors one referrer chain."""
import collections
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import make_batch  # noqa: E402
from pcaccumulation_amd import distributed as pdist  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.loss import FuseLoss  # noqa: E402
from pcaccumulation_amd.motionnet import MotionNet  # noqa: E402
from pcaccumulation_amd.synthetic import fill_state_dict_  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'train', n_sweeps=5)
    cfg['misc']['compute_dtype'] = 'mixed'
    torch.manual_seed(0)
    model = MotionNet(cfg)
    fill_state_dict_(model)
    model = model.to(dev).train().channels_last_()
    inp = make_batch(cfg, [21, 22], 5, 30000, mode='lidar')
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-5, fused=True)
    step = pdist.DataParallelStep(model, opt, FuseLoss(cfg['loss']), iter_size=1, grad_clip=1.0, catch=False, pipelined=True, two_streams=True, early_thread=True)
    for _ in range(3):
        step(dict(inp))
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    gc.set_debug(gc.DEBUG_SAVEALL)
    base = torch.cuda.memory_allocated()
    for k in range(3):
        step(dict(inp))
        torch.cuda.synchronize()
        print('after step %d with the collector off: %.1f MB allocated beyond the start' % (k, (torch.cuda.memory_allocated() - base) / 1e6))
    n = gc.collect()
    torch.cuda.synchronize()
    tens = [o for o in gc.garbage if torch.is_tensor(o)]
    print('unreachable objects after three steps: %d; tensors among them: %d, %.1f MB on the device' % (
        n, len(tens), sum(t.numel() * t.element_size() for t in tens if t.is_cuda) / 1e6))
    print(collections.Counter(type(o).__name__ for o in gc.garbage).most_common(30))
    for t in sorted(tens, key=lambda t: -t.numel())[:8]:
        refs = [type(r).__name__ for r in gc.get_referrers(t) if r is not tens and r is not gc.garbage][:6]
        print('  tensor %s %s  referrers: %s' % (tuple(t.shape), t.dtype, refs))
    cyc = [o for o in gc.garbage if type(o).__name__ in ('function', 'cell', 'frame', 'method')]
    for o in cyc[:12]:
        print('  %s %s' % (type(o).__name__, getattr(o, '__qualname__', repr(o))[:120]))


if __name__ == '__main__':
    main()
