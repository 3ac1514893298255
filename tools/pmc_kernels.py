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
# fused pillar-encoder block (two-piece rows), wide row layer with weights in registers, fg/bg head kernels
m = 1_169_433
xa, pooled = f32(rows, 32), f32(m, 32)
p2v = torch.randint(0, m, (rows,), device=dev, dtype=torch.int32)
w0, ws, w1 = torch.randn(32, 64, device=dev) / 8, torch.randn(32, 64, device=dev) / 8, torch.randn(32, 32, device=dev) / 6
b0, b1 = torch.randn(32, device=dev), torch.randn(32, device=dev)
aa, ap = native.absmax256(xa), native.absmax256(pooled)
for _ in range(3):
    out, hr, xm, hm, oa, ha = native.pfn_block_split_forward(xa, aa, pooled, ap, p2v, w0, b0, ws, w1, b1)
g = f32(rows, 32)
ag = native.absmax256(g)
for _ in range(3):
    native.pfn_block_split_dgrad(g, ag, xm, hm, w0, ws, w1, True)
x128 = f32(429_567, 128)
a128 = native.absmax256(x128)
w128 = torch.randn(128, 128, device=dev) / 11
for _ in range(3):
    native.rows_linear_split(x128, a128, w128, None, None, False, True)      # rows_linear_split_fm_kernel<128, 4>
xh = bf(20, 288, 288, 32)
wh = torch.randn(2, 32, 3, 3, device=dev) / 17
dyh = f32(20, 288, 288, 2)
for _ in range(3):
    native.head_conv3x3_forward(xh, wh, torch.zeros(2, device=dev))
    native.head_conv3x3_dgrad(dyh, wh, 32, torch.bfloat16)
    native.head_conv3x3_wgrad(dyh, xh)
# ---- round 4: the 'mixed' mode's kernels -- fp32x3 forward with a bf16 second output, pool backward with fp32 winners, batched collate + voxelise
# (the bf16 second outputs of the split kernels share their kernels' names with the plain launches above: one extra 2-byte store per element;
#  they are timed in tools/native_call_table.py, not here)
yp = torch.relu(f32(20, 288, 288, 32))
gp, gs = bf(20, 144, 144, 32), bf(20, 288, 288, 64)[..., 32:]
for _ in range(3):
    native.pool_skip_relu_backward(yp, gp, gs)                                                                          # pool_skip_relu_bwd_y32_kernel
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.pipeline import sample_to_device  # noqa: E402
from pcaccumulation_amd.synthetic import make_sequence  # noqa: E402
cfgv = default_config('waymo', 'train', n_sweeps=5)
smp = [sample_to_device(make_sequence(i, 5, 160000, cfgv), dev) for i in range(4)]
vg = cfgv['voxel_generator']
import numpy as np  # noqa: E402
grid = np.round((np.array(vg['range'][3:], np.float32) - np.array(vg['range'][:3], np.float32)) / np.array(vg['voxel_size'], np.float32)).astype(int).tolist()
for _ in range(3):
    native.collate_voxelize(smp, vg['voxel_size'], vg['range'], grid, 5)                                               # vox_batch_keys / vox_batch_assign / vox_p2v
torch.cuda.synchronize()
print('done')
