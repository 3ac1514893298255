#!/usr/bin/env python
"""Wall-clock per stage of one train step (sync after each stage), printed as it goes. Development aid."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
from pcaccumulation_amd.synthetic import make_sequence

T0 = time.time()
def log(msg):
    torch.cuda.synchronize()
    print('[%7.2fs] %s' % (time.time() - T0, msg), flush=True)

def main():
    ppf = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'train', n_sweeps=5)
    cfg['misc']['compute_dtype'] = dtype
    model, opt, loss_fn = bench.build(cfg, dev)
    batcher = DeviceBatcher(cfg)
    log('model built')
    scene = sample_to_device(make_sequence(1, 5, ppf, cfg), dev)
    log('scene on device')
    for it in range(steps):
        inp = batcher([scene] * int(os.environ.get("BATCH", "1"))); log('it%d batcher M=%d' % (it, inp['coordinates'].shape[0]))
        out = model(inp); log('it%d forward' % it)
        stats = loss_fn(out, inp); log('it%d loss=%.4f' % (it, float(stats['loss'])))
        stats['loss'].backward(); log('it%d backward' % it)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step(); opt.zero_grad(set_to_none=True); log('it%d optimizer' % it)

if __name__ == '__main__':
    main()
