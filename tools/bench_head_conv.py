"""Time the SegHead2D output convolution kernels (csrc/head_conv.hip) on the model's shape: 32 -> 2 @ 288^2 x 20, f32 and bf16 rows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native
from bench_conv import timeit
dev = torch.device('cuda:0')
w = torch.randn(2, 32, 3, 3, device=dev) / 17
b = torch.randn(2, device=dev)
dy = torch.randn(20, 288, 288, 2, device=dev)
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(20, 288, 288, 32, device=dev).to(dt)
    mb = x.numel() * x.element_size() / 1e6
    tf = timeit(lambda: native.head_conv3x3_forward(x, w, b), iters=50)
    td = timeit(lambda: native.head_conv3x3_dgrad(dy, w, 32, dt), iters=50)
    tw = timeit(lambda: native.head_conv3x3_wgrad(dy, x), iters=50)
    print('%-8s fwd %6.1f us (%4.0f GB/s)  dgrad %6.1f us (%4.0f GB/s)  wgrad %6.1f us (%4.0f GB/s)' % (
        str(dt).split('.')[-1], tf, mb / tf * 1e3, td, mb / td * 1e3, tw, mb / tw * 1e3))
