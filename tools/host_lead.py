#!/usr/bin/env python
"""How far the host runs ahead of the GPU at the phase boundaries of one training step (bench.py's step, mixed, four sequences): at each boundary
the host time it was reached and the time the GPU reached the event recorded there.  GPU time minus host time ~ 0: the GPU was waiting for the host
(host-bound phase); several ms: the host is ahead (GPU-bound phase).  Usage: host_lead.py [steps=6]   (PCACC_DTYPE, PCACC_BATCH, PCACC_EARLY_THREAD)"""
import os
import sys
import time

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd import distributed as pdist
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence

dev = torch.device('cuda:0')
pdist.bind_to_l3_domain(0, 1)
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'mixed')
cfg['pose_estimation']['kpt_sampler'] = 'device'
B = int(os.environ.get('PCACC_BATCH', '4'))
model, opt, loss_fn = bench.build(cfg, dev)
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=cfg['train']['grad_clip'], catch=False,
                                 early_thread={'0': False, '1': True}.get(os.environ.get('PCACC_EARLY_THREAD', '1')))
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(2 * B)]
feed = bench.BatchFeed(batcher, lambda i: [scenes[(i * B + j) % len(scenes)] for j in range(B)], True)
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()                                             # on the stream that is current where the boundary is crossed
    marks.append((name, time.perf_counter(), ev))


early_terms, call = loss_fn.early_terms, type(loss_fn).__call__


def early_terms_marked(results, **k):
    mark('sync + ego head issued (early backward starts)')
    return early_terms(results, **k)


class Marked(type(loss_fn)):
    def __call__(self, *a, **k):
        mark('upper half of the forward issued (side stream)')
        out = call(self, *a, **k)
        mark('loss issued (side stream)')
        return out


loss_fn.early_terms = early_terms_marked
loss_fn.__class__ = Marked
lower = model.pillar_encoder.forward


def lower_marked(*a, **k):
    mark('batch ready, forward starts')
    return lower(*a, **k)


model.pillar_encoder.forward = lower_marked
ego_fp = model.ego_motion_head.forward_pillars


def ego_marked(*a, **k):
    mark('host sync returned, ego head starts')
    return ego_fp(*a, **k)


model.ego_motion_head.forward_pillars = ego_marked
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(4):
    bench.train_step(stepper, batcher, feed)
rows = []
for it in range(n):
    torch.cuda.synchronize()
    base_ev = torch.cuda.Event(enable_timing=True)
    base_ev.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()                                # host and GPU clocks aligned here (to the sync's return latency)
    del marks[:]
    mark('step starts')
    inp = feed.next()
    stepper(inp, before_sync=lambda: (mark('lower half issued, host starts waiting'), feed.prefetch()), after_forward=feed.prefetch).resolve()
    mark('backward + clip + Adam issued')
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    rows.append([(name, (th - t0) * 1e3, base_ev.elapsed_time(ev)) for name, th, ev in marks] + [('all work done (host sees it)', (t_end - t0) * 1e3, float('nan'))])
print('step time %s ms; per boundary: host reached it at / GPU reached it at (ms from step start), mean over %d steps' % (' '.join('%.1f' % r[-1][1] for r in rows), n))
for i, (name, _, _) in enumerate(rows[0]):
    h = sum(r[i][1] for r in rows) / n
    g = sum(r[i][2] for r in rows) / n
    print('  %-52s host %6.2f   gpu %6.2f   gpu - host %6.2f' % (name, h, g, g - h))
