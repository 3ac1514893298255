"""Time the fp32x3 row-linear kernel on the model's wide layers; PCACC_ROWS_FM_OFF=1 selects the weights-in-LDS kernel for all of them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native
from bench_conv import timeit
dev = torch.device('cuda:0')
for rows, k, n in [(429567, 128, 128), (320000, 128, 64), (320000, 64, 128), (3200000, 32, 32), (3200000, 64, 32)]:
    x = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    am = native.absmax256(x)
    t = timeit(lambda: native.rows_linear_split(x, am, w, b, want_amax=True), iters=30)
    mb = rows * (k + n) * 4 / 1e6
    print('%8d x %3d -> %3d  %7.1f us  %5.0f GB/s' % (rows, k, n, t, mb / t * 1e3))
