"""Time of the 9 -> 64 position layer (3.2 M rows): fp32 output, and fp32 + bf16 + maxima."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native
dev = torch.device('cuda:0')
x = torch.randn(3_200_000, 9, device=dev); w = torch.randn(64, 9, device=dev) / 3; b = torch.randn(64, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print('fp32 out       %.1f us' % t(lambda: native.rows_linear(x, w, b, None, False, False, out_dtype=torch.float32)))
print('fp32+bf16+max  %.1f us' % t(lambda: native.rows_linear_few_dual(x, w, b)))
print('bf16 out       %.1f us' % t(lambda: native.rows_linear(x, w, b, None, False, False, out_dtype=torch.bfloat16)))
