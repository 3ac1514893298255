#!/usr/bin/env python
"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per pass): the pillar-scatter launch of
bench.py's roofline object -- the canvas of a 4-sequence step (20 x 288 x 288 cells, 32 channels) filled from M = 1 169 433 pillar
rows in cell order -- in its three forms (f32 rows -> f32 canvas, f32 rows -> bf16 canvas, bf16 rows -> bf16 canvas = the bf16
compute mode), plus a plain 128 MiB device copy whose byte count is known, to calibrate the counters
(MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950; WRITE_SIZE must be calibrated).
tools/pmc_summary.py turns the two counter files into profiles/rNN_pmc_scatter_summary.json."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
n_cells, m, c = 20 * 288 * 288, 1_169_433, 32
occupied = torch.randperm(n_cells, device=dev)[:m].sort().values
c2p = torch.full((n_cells,), -1, dtype=torch.int32, device=dev)
c2p[occupied] = torch.arange(m, dtype=torch.int32, device=dev)
feats = torch.randn(m, c, device=dev)
feats16 = feats.to(torch.bfloat16)
# [r6] the fused form ('mixed' mode): the encoder's last max-pooling writes both canvases -- 3.2 M point rows in the same M pillars (every pillar >= 1 point)
n = 3_200_000
p2v = torch.cat([torch.arange(m, device=dev), torch.randint(0, m, (n - m,), device=dev)])[torch.randperm(n, device=dev)].to(torch.int32)
rows = torch.randn(n, c, device=dev)
offs, order = native.csr_build(p2v, m)
big = torch.empty(512 * 1024 * 1024 // 4, device=dev)          # 512 MiB: evict the 256 MiB Infinity Cache between launches
src = torch.randn(32 * 1024 * 1024, device=dev)                # 128 MiB
dst = torch.empty_like(src)
for it in range(5):
    big.fill_(float(it))
    dst.copy_(src)                                             # calibration: reads 128 MiB, writes 128 MiB
    big.fill_(float(it) + 0.5)
    native.pillar_scatter(feats, c2p, torch.float32)
    big.fill_(float(it) + 0.25)
    native.pillar_scatter(feats, c2p, torch.bfloat16)
    big.fill_(float(it) + 0.125)
    native.pillar_scatter(feats16, c2p, torch.bfloat16)
    big.fill_(float(it) + 0.0625)
    native.segment_max_canvas(rows, offs, order, m, c2p)
torch.cuda.synchronize()
print('done M=%d cells=%d' % (m, n_cells))
