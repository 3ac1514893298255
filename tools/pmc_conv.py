#!/usr/bin/env python
"""Workload for PMC passes over the MFMA conv kernel: one layer shape, a few launches.  Usage: pmc_conv.py c_in c_out H n_img [kt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402

ci, co, h, n = (int(v) for v in sys.argv[1:5])
kt = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device('cuda:0')
x = torch.randn(n, h, h, ci, device=dev).to(torch.bfloat16)
w = torch.randn(*((co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)), device=dev) / (3 * ci ** 0.5)
b = torch.randn(co, device=dev)
wp = native.conv3x3_prepare_weights(w)
for _ in range(5):
    y = native.conv3x3(x, wp, b, 5 if kt == 3 else 1, True)
torch.cuda.synchronize()
print('done')
