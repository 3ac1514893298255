#!/usr/bin/env python
"""Every host<->device synchronisation of one training step, with the Python line that causes it
(torch.cuda.set_sync_debug_mode('warn')).  Development aid."""
import os, sys, warnings, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence

dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = 'bf16'; cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(4)]
for it in range(2):
    bench.train_step(model, opt, loss_fn, batcher, scenes, None, 1.0)
torch.cuda.synchronize()
hits = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' in str(message):
        frames = [f for f in traceback.extract_stack() if '/root/repo/' in f.filename or 'repo/pcacc' in f.filename or 'bench.py' in f.filename]
        key = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(frames[-3:])) if frames else '%s:%d' % (filename, lineno)
        hits[key] += 1
warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
bench.train_step(model, opt, loss_fn, batcher, scenes, None, 1.0)
torch.cuda.set_sync_debug_mode('default')
print('syncs in one step:', sum(hits.values()))
for k, v in hits.most_common():
    print('%3d  %s' % (v, k))
