#!/usr/bin/env python
"""Catch the race behind the fg/bg head's last weight gradient in the staged step: around every native.head_conv3x3_wgrad call take stream-ordered clones of
its operands BEFORE the call, run the call, then a device-wide synchronisation, and compare: operands now against their earlier clones (somebody was still
writing them), and the call repeated on the quiescent device against the first result."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from helpers import make_batch
from pcaccumulation_amd import distributed as pdist, native
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.loss import FuseLoss
from pcaccumulation_amd.motionnet import MotionNet
from pcaccumulation_amd.synthetic import fill_state_dict_

dev = torch.device('cuda:0')
cfg = default_config('waymo', 'train', n_sweeps=5)
cfg['misc']['compute_dtype'] = 'mixed'
torch.manual_seed(0)
model = MotionNet(cfg); fill_state_dict_(model)
model = model.to(dev).train().channels_last_()
inp = make_batch(cfg, [21, 22], 5, 30000, mode='lidar')
inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
opt = torch.optim.SGD(model.parameters(), lr=0.0)
orig = native.head_conv3x3_wgrad
events = []


import ctypes


def raw_wgrad(gy, x):
    """native.head_conv3x3_wgrad with a workspace this script keeps: -> (dw, partial slots as a float tensor)."""
    n, h, w, c_out = gy.shape
    c_in = x.shape[3]
    dw = torch.empty((c_out, c_in, 3, 3), dtype=torch.float32, device=gy.device)
    db = torch.empty((c_out,), dtype=torch.float32, device=gy.device)
    need = ctypes.c_size_t(0)
    native.lib().pcacc_head_conv3x3_wgrad_workspace_bytes(int(n), int(h), int(w), int(c_in), int(c_out), ctypes.byref(need))
    ws = torch.empty(need.value // 4, dtype=torch.float32, device=gy.device)
    rc = native.lib().pcacc_head_conv3x3_wgrad(ctypes.c_void_p(gy.data_ptr()), ctypes.c_void_p(x.data_ptr()), 0, ctypes.c_void_p(dw.data_ptr()), ctypes.c_void_p(db.data_ptr()),
                                               int(n), int(h), int(w), int(c_in), int(c_out), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(need.value), native._stream())
    assert rc == 0
    return dw, ws


def spy(gy, x, want_bias=True):
    g0, x0 = gy.clone(), x.clone()                     # stream-ordered: what the operands hold when the kernel is queued
    out = orig(gy, x, want_bias=want_bias)
    dw1, ws1 = raw_wgrad(gy, x)                        # in flight, beside whatever the other stream runs
    dwc, wsc = raw_wgrad(g0, x0)                       # the same on the private copies of the operands
    g1 = gy.clone()                                    # what a copy kernel sees of dy in the same window
    p1 = ws1.clone()
    torch.cuda.synchronize()                           # the whole device is quiet
    dw2, ws2 = raw_wgrad(gy, x)
    torch.cuda.synchronize()
    copies = int((dwc != dw2).sum())
    elems = gy.shape[3] * x.shape[3] * 9 + gy.shape[3]
    d = (p1 != ws2).nonzero()[:, 0]
    slots = torch.unique(d // elems).tolist()
    explain = []
    if d.numel():
        # which pixel's dy explains the difference?  delta[c][ci][tap] = k_c * x[pixel + tap][ci] for the pixel whose gradient the kernel saw differently
        n_img, H, W, co = gy.shape
        ci_n = x.shape[3]
        tiles_y, tiles_x = (H + 7) // 8, (W + 31) // 32
        n_tiles = n_img * tiles_y * tiles_x
        grid = p1.numel() // elems
        xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))                       # [n, H+2, W+2, ci]
        for slot in slots[:3]:
            delta = (p1.view(grid, elems)[slot] - ws2.view(grid, elems)[slot])[:co * ci_n * 9].view(co, ci_n, 9).double()
            best = None
            for t in range(slot, n_tiles, grid):
                img, r = divmod(t, tiles_y * tiles_x)
                y0, x0_ = (r // tiles_x) * 8, (r % tiles_x) * 32
                patch = xp[img, y0:y0 + 10, x0_:x0_ + 34]                          # [10, 34, ci]
                win = torch.stack([patch[dy_:dy_ + 8, dx_:dx_ + 32] for dy_ in range(3) for dx_ in range(3)], dim=-1).double()   # [8, 32, ci, 9]
                for c in range(co):
                    num = (win * delta[c]).sum(dim=(2, 3))
                    den = (win * win).sum(dim=(2, 3)).clamp(min=1e-30)
                    k = num / den
                    res = ((delta[c] - k[..., None, None] * win) ** 2).sum(dim=(2, 3)) / (delta[c] ** 2).sum().clamp(min=1e-30)
                    j = int(res.argmin())
                    yy, xx = divmod(j, 32)
                    cand = (float(res.view(-1)[j]), c, t, img, y0 + yy, x0_ + xx, float(k.view(-1)[j]))
                    if float((delta[c] ** 2).sum()) > 0 and (best is None or cand[0] < best[0]):
                        best = cand
            if best:
                _, c, t, img, yy, xx, k = best
                explain.append(dict(slot=slot, channel=c, tile=t, pixel=(img, yy, xx), k=k, residual=best[0], dy_at_pixel=gy[img, min(yy, H - 1), min(xx, W - 1)].tolist(),
                                    dy_next_in_thread=gy[img, min(yy + 1, H - 1), min(xx, W - 1)].tolist() if yy + 1 < H else None))
    events.append(dict(explain=explain, on_copies_vs_quiet=copies, gy_inflight_copy_col0=int((g1[..., 0] != gy[..., 0]).sum()), gy_inflight_copy_col1=int((g1[..., 1] != gy[..., 1]).sum()), gy_changed=int((g0 != gy).sum()), x_changed=int((x0 != x).sum()), dw_inflight_vs_quiet=int((dw1 != dw2).sum()), api_vs_quiet=int((out[0] != dw2).sum()),
                       partial_elems_differ=int(d.numel()), slots=slots[:8], n_elems_in_slot=len(set((d % elems).tolist())),
                       max_rel=float(((p1 - ws2).abs().max() / ws2.abs().max()).item()) if d.numel() else 0.0))
    return out


native.head_conv3x3_wgrad = spy
for trial in range(4):
    step = pdist.DataParallelStep(model, opt, FuseLoss(cfg['loss']), iter_size=1, grad_clip=None, catch=False, pipelined=True, two_streams=True, early_thread=True)
    for r in range(8):
        torch.manual_seed(5)
        step(dict(inp))
        torch.cuda.synchronize()
bad = [e for e in events if e['gy_changed'] or e['x_changed'] or e['dw_inflight_vs_quiet'] or e['api_vs_quiet'] or e['partial_elems_differ'] or e['on_copies_vs_quiet']]
print('%d calls, %d with a difference' % (len(events), len(bad)))
for e in bad[:12]:
    print(e)
