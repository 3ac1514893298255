"""GPU time of every aten op the package dispatches during one training step, by source line (each op timed alone behind a spin kernel, as
tools/native_call_table.py does for the library calls; ops run by built-in autograd nodes have no package frame and are keyed by op name).
Calls into libpcacc_hip.so are not aten ops and do not appear.  Development aid for the torch tail.
Usage: [PCACC_DTYPE=bf16|fp32x3] python tools/torch_time_by_line.py [rows=60]"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from pcaccumulation_amd import distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16'); cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=1.0, two_streams=False, pipelined=False)
for _ in range(2):
    bench.train_step(stepper, batcher, scenes)
SKIP = ('view', 'permute', 'select', 'slice', 'expand', 'unsqueeze', 'squeeze', 'detach', 'alias', 'aten.t.default', 'transpose', 'as_strided', 'sym_',
        'empty', 'reshape', '_unsafe_view', 'split', 'unbind', 'narrow', 'lift_fresh', 'is_', 'stride', 'size', 'numel', 'record_stream', '_local_scalar')
rows = collections.OrderedDict()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(s in name for s in SKIP):
            return func(*args, **(kwargs or {}))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(300_000)
        s.record()
        out = func(*args, **(kwargs or {}))
        e.record()
        torch.cuda.synchronize()
        where = None
        for fr in reversed(traceback.extract_stack(limit=20)):
            if 'pcaccumulation_amd' in fr.filename or fr.filename.endswith('bench.py'):
                where = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                break
        key = where or ('autograd: ' + name.replace('aten.', ''))
        r = rows.setdefault(key, [0.0, 0])
        r[0] += s.elapsed_time(e) * 1e3
        r[1] += 1
        return out
with Mode():
    bench.train_step(stepper, batcher, scenes)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tot = sum(v[0] for v in rows.values())
print('aten ops: %.2f ms in %d dispatches (timed alone)' % (tot / 1e3, sum(v[1] for v in rows.values())))
for k, (us, c) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:n]:
    print('%8.1f us  n=%3d  %s' % (us, c, k))
