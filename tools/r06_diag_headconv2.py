#!/usr/bin/env python
"""Does pcacc_head_conv3x3_wgrad depend on what its workspace held before?  Same inputs, workspace block pre-filled with NaN / with large values / zeros."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcaccumulation_amd import native
dev = torch.device('cuda:0')
torch.manual_seed(0)
for n_img, h, w, ci, co in ((10, 288, 288, 32, 2), (20, 288, 288, 32, 2), (10, 288, 288, 64, 2), (3, 50, 70, 32, 4)):
    dy = torch.randn(n_img, h, w, co, device=dev)
    x = torch.randn(n_img, h, w, ci, device=dev)
    ref = None
    for fill in (0.0, float('nan'), 1e30, 0.0):
        import ctypes
        need = ctypes.c_size_t(0)
        native.lib().pcacc_head_conv3x3_wgrad_workspace_bytes(n_img, h, w, ci, co, ctypes.byref(need))
        junk = torch.full((max(need.value, 256) // 4 + 64,), fill, device=dev)
        ptr = junk.data_ptr()
        del junk
        probe = native._ws(need.value, dev)
        same_block = probe.data_ptr() == ptr
        del probe
        dw, db = native.head_conv3x3_wgrad(dy, x)
        torch.cuda.synchronize()
        if ref is None:
            ref = (dw.clone(), db.clone())
        print((n_img, h, w, ci, co), 'workspace pre-filled with', fill, 'same block reused:', same_block, '| finite:', bool(torch.isfinite(dw).all()), bool(torch.isfinite(db).all()),
              '| equal to the first call:', torch.equal(dw, ref[0]), torch.equal(db, ref[1]), '| max diff', float((dw - ref[0]).abs().max()))
