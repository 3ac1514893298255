#!/bin/bash
# how close to its 1 % bound does test_mixed_gradients_against_fp32x3 sit on `pillar_encoder.blocks.0.fc_0.bias`?  8 evaluations per build of the library
for v in new prev; do
  if [ $v = prev ]; then export PCACC_LIB=$PWD/build/libpcacc_hip_prev.so; else unset PCACC_LIB; fi
  python - <<PY
import sys, torch
sys.path.insert(0, 'tests')
import test_mixed as tm
from helpers import make_batch
from pcaccumulation_amd.config import default_config
dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
inp = make_batch(cfg, [51, 52], 3, 6000)
inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
k = 'pillar_encoder.blocks.0.fc_0.bias'
out = []
for i in range(8):
    _, _, ref = tm._step('fp32x3', inp)
    _, _, got = tm._step('mixed', inp)
    n = float(ref[k].norm())
    out.append(round(abs(float(got[k].norm()) - n) / n, 5))
print('$v', out)
PY
done
