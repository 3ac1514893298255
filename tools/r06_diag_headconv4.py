#!/usr/bin/env python
"""Is pcacc_head_conv3x3_wgrad sensitive to what else runs on the device?  The same call on the same operands, repeated while a second stream keeps the device busy
with (a) large fills / copies, (b) the library's own convolution kernels; every result compared with the result on a quiet device."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcaccumulation_amd import native
dev = torch.device('cuda:0')
torch.manual_seed(0)
n_img, h, w, ci, co = 10, 288, 288, 32, 2
dy = torch.randn(n_img, h, w, co, device=dev) * (torch.rand(n_img, h, w, 1, device=dev) < 0.3)
x = torch.relu(torch.randn(n_img, h, w, ci, device=dev))
quiet = native.head_conv3x3_wgrad(dy, x)[0].clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
big = torch.empty(64 * 1024 * 1024, device=dev)
a16 = torch.randn(20, 288, 288, 32, device=dev).to(torch.bfloat16)
xs = torch.randn(4, 288, 288, 64, device=dev)
for name in ('quiet', 'fills on a second stream', 'gathers + row kernels on a second stream', 'quiet again'):
    bad = 0
    for it in range(200):
        if 'fills' in name:
            with torch.cuda.stream(side):
                big.fill_(float(it)); big.mul_(1.0001)
        elif 'gathers' in name:
            with torch.cuda.stream(side):
                idx = torch.randint(0, xs.numel() // 64, (300000,), device=dev, dtype=torch.int32)
                r = native.gather_rows(xs.view(-1, 64), idx)
                m = native.maxpool2x2(a16)
        dw = native.head_conv3x3_wgrad(dy, x)[0]
        bad += int(not torch.equal(dw, quiet))
    torch.cuda.synchronize()
    print('%-45s %d of 200 results differ from the quiet one' % (name, bad), flush=True)
