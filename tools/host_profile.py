#!/usr/bin/env python
"""cProfile of the host side of train steps on the GPU (which Python functions keep the GPU waiting). Development aid."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence

B = int(os.environ.get('BATCH', '4'))
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = os.environ.get('PCACC_DTYPE', 'bf16'); cfg['pose_estimation']['kpt_sampler'] = 'device'
model, opt, loss_fn = bench.build(cfg, dev)
batcher = DeviceBatcher(cfg)
scenes = [sample_to_device(make_sequence(i, 5, 160000, cfg), dev) for i in range(B)]
from pcaccumulation_amd import distributed as pdist
stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=1.0)
for _ in range(3):
    bench.train_step(stepper, batcher, scenes)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
for _ in range(3):
    bench.train_step(stepper, batcher, scenes)
torch.cuda.synchronize()
pr.disable()
print('ms/step', (time.time() - t0) / 3 * 1e3)
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(25)
