"""Time the resident fp32x3 conv kernel (32 -> 32 @ 288^2 x 20) of the library named by PCACC_LIB: one number per run."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native
from bench_conv import timeit
dev = torch.device('cuda:0')
x = torch.randn(20, 288, 288, 32, device=dev)
wt = torch.randn(32, 32, 3, 3, device=dev) / 17
wf, wb = native.conv3x3_split_prepare_weights(wt)
bias = torch.randn(32, device=dev)
am = native.absmax256(x)
t = timeit(lambda: native.conv3x3_split(x, wf, bias, 1, True, amax=am), iters=50)
t2 = timeit(lambda: native.conv3x3_split(x, wf, bias, 1, True, amax=am, want_amax=True), iters=50)
print('%-40s %7.1f us  (with amax out %7.1f us)' % (os.environ.get('PCACC_LIB', 'default').split('/')[-1], t, t2))
