#!/usr/bin/env python
"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass): the pillar-scatter kernel
at c3 size (fp32 and bf16 canvas) plus a plain 128 MiB device copy whose byte count is known, used to calibrate the counters
(MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950; WRITE_SIZE must be calibrated)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402

dev = torch.device('cuda:0')
T, nx, ny = 5, 288, 288
rng = np.random.RandomState(0)
n_cells = T * ny * nx
m = 292250
occ = np.sort(rng.permutation(n_cells)[:m]).astype(np.int64)
coords = np.zeros((m, 5), np.int32)
perm = rng.permutation(m)                     # pillar ids are in first-touch order, not cell order
occ = occ[perm]
coords[:, 4] = occ // (ny * nx)
coords[:, 2] = (occ % (ny * nx)) // nx
coords[:, 3] = occ % nx
cell, c2p = native.cell_index(torch.from_numpy(coords).to(dev), nx, ny, T, 1)
feats = torch.randn(m, 32, device=dev)
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
torch.cuda.synchronize()
print('done M=%d cells=%d' % (m, n_cells))
