#!/usr/bin/env python
"""Workload for the PMC passes over the MFMA kernels (rocprofv3 --pmc <set> --kernel-trace, one counter set per run):
  conv3x3_resident_kernel   64 -> 64 @ 288^2, 20 images      (csrc/conv.hip, weights resident in LDS)
  conv3x3_strip_kernel      256 -> 256 @ 36^2, 20 images     (csrc/conv_deep.hip, forward / data gradient of the deep layers)
  conv3x3_wgrad_kernel      32 -> 32 @ 288^2, 20 images      (csrc/conv.hip)
  conv3x3_wgrad_strip_kernel 256 x 256 @ 36^2, 20 images     (csrc/conv_deep.hip)
  rows_wgrad_bf16_kernel    3.2 M rows, 32 -> 32             (csrc/mlp_mfma.hip, the pillar encoder's weight gradients)
  rows_linear_bf16_kernel   3.2 M rows, 64 -> 32
three launches each.  tools/pmc_kernels_summary.py turns the counter files into profiles/rNN_pmc_kernels_summary.json."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
bf = lambda *shape: torch.randn(*shape, device=dev).to(torch.bfloat16)


def conv(n, h, ci, co):
    x = bf(n, h, h, ci)
    w = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    b = torch.randn(co, device=dev)
    wp = native.conv3x3_prepare_weights(w)
    for _ in range(3):
        native.conv3x3(x, wp, b, 1, True)


def wgrad(n, h, ci, co, deep):
    x, gy = bf(n, h, h, ci), bf(n, h, h, co)
    for _ in range(3):
        (native.conv3x3_wgrad_deep if deep else native.conv3x3_wgrad)(gy, x)


conv(20, 288, 64, 64)
conv(20, 36, 256, 256)
wgrad(20, 288, 32, 32, False)
wgrad(20, 36, 256, 256, True)
rows = 3_200_000
x, dy = bf(rows, 32), bf(rows, 32)
for _ in range(3):
    native.rows_wgrad(dy, x)
x64 = bf(rows, 64)
w = torch.randn(32, 64, device=dev) / 8
for _ in range(3):
    native.rows_linear(x64, w, None, None, False, True, out_dtype=torch.bfloat16)
# ---- fp32x3 kernels (round 3): fp32 operands, scaled fp16 hi / lo products
f32 = lambda *shape: torch.randn(*shape, device=dev)


def conv_split(n, h, ci, co, kt=1, frames=1):
    x, gy = f32(n, h, h, ci), f32(n, h, h, co)
    w = torch.randn(*((co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)), device=dev) / (3 * (ci * kt) ** 0.5)
    b = torch.randn(co, device=dev)
    wf, wb = native.conv3x3_split_prepare_weights(w)
    ax, ag = native.absmax256(x), native.absmax256(gy)
    for _ in range(3):
        native.conv3x3_split(x, wf, b, frames, True, amax=ax)
    for _ in range(3):
        native.conv3x3_wgrad_split(gy, x, frames, 0, dy_amax=ag, x_amax=ax)


conv_split(20, 288, 32, 32)               # conv3x3_split_res_kernel<32,1,2,5> + conv3x3_wgrad_split_kernel<1,1,9>
conv_split(20, 36, 256, 256)              # conv3x3_split_kernel<64,2,2,2,9> + conv3x3_wgrad_split_kernel<2,2,9>
xr, dyr = f32(rows, 32), f32(rows, 32)
ar, ad = native.absmax256(xr), native.absmax256(dyr)
w32 = torch.randn(32, 32, device=dev) / 6
for _ in range(3):
    native.rows_linear_split(xr, ar, w32, None, None, False, True)
for _ in range(3):
    native.rows_wgrad_split(dyr, ad, xr, ar)
for _ in range(3):
    native.absmax256(xr)
torch.cuda.synchronize()
print('done')
