#!/usr/bin/env python
"""Where one data-parallel step spends its time with N ranks (host wall clock per phase, device drained at each boundary).
Development aid.  Launch: PCACC_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... tools/dist_phase_times.py [two|one]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from pcaccumulation_amd import distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence

mode = sys.argv[1] if len(sys.argv) > 1 else 'two'
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group(os.environ.get('PCACC_DIST_BACKEND', 'nccl'))
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = 'bf16'; cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(10 * rank + i, 5, 160000, cfg), dev) for i in range(2)]
step = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=1.0, two_streams=(mode == 'two'))
r = step.reducer
marks = []
def mark(name):
    torch.cuda.synchronize()
    marks.append((name, time.time()))
# wrap the reducer's phases
for name in ('prepare', 'finish', 'agree', '_launch'):
    orig = getattr(r, name)
    def make(orig, name):
        def w(*a, **k):
            t0 = time.time()
            out = orig(*a, **k)
            acc[name] = acc.get(name, 0.0) + time.time() - t0
            cnt[name] = cnt.get(name, 0) + 1
            return out
        return w
    setattr(r, name, make(orig, name))
for it in range(4):
    acc, cnt = {}, {}
    inp = batcher(scenes)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.time()
    stats = step(inp)
    t_issue = time.time() - t0
    torch.cuda.synchronize()
    t_all = time.time() - t0
    if rank == 0:
        print('step %d mode %s: host issue %.1f ms, until drained %.1f ms; reducer host time %s calls %s buckets %d early %d' % (
            it, mode, t_issue * 1e3, t_all * 1e3, {k: round(v * 1e3, 1) for k, v in acc.items()}, cnt, len(r.buckets), r._n_early), flush=True)
dist.destroy_process_group()
