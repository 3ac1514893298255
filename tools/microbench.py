#!/usr/bin/env python
"""Per-kernel timing at BASELINE sizes (HIP events on the launch stream), printed as algorithmic GB/s.
Development aid; bench.py is the contract benchmark."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = torch.device('cuda:0')
    T = int(os.environ.get('T', 5))
    ppf = int(os.environ.get('PPF', 160000))
    nx = ny = 288
    n = T * ppf
    rng = np.random.RandomState(0)
    pts = np.concatenate([rng.uniform(-31.9, 31.9, (n, 2)), rng.uniform(-1.6, 5.9, (n, 1)),
                          np.repeat(np.arange(T), ppf)[:, None]], axis=1).astype(np.float32)
    p4 = torch.from_numpy(pts).to(dev)
    res = {}
    vs, rg = [0.25, 0.25, 8], [-36, -36, -2, 36, 36, 6]
    coords, p2v, num = native.voxelize(p4, vs, rg, (nx, ny, 1), T, nx * ny * T)
    m = int(num.item())
    n_cells = T * ny * nx
    t = timeit(lambda: native.voxelize(p4, vs, rg, (nx, ny, 1), T, nx * ny * T))
    res['voxelize'] = dict(ms=t * 1e3, GBps=(16 * n + 4 * n + 16 * m + 4 * n_cells) / t / 1e9)
    c5 = torch.cat([torch.zeros(m, 1, dtype=torch.int32, device=dev), coords[:m]], 1).contiguous()
    cell, c2p = native.cell_index(c5, nx, ny, T, 1)
    t = timeit(lambda: native.cell_index(c5, nx, ny, T, 1))
    res['cell_index'] = dict(ms=t * 1e3)
    t = timeit(lambda: native.csr_build(p2v, m))
    res['csr_build'] = dict(ms=t * 1e3)
    offs, order = native.csr_build(p2v, m)
    p3 = p4[:, :3].contiguous()
    lab = torch.zeros(n, dtype=torch.int64, device=dev)
    t = timeit(lambda: native.segment_mean3_maxlabel(p3, lab, offs, order, m))
    res['segment_mean3_maxlabel'] = dict(ms=t * 1e3, GBps=((12 + 4 + 8) * n + 20 * m) / t / 1e9)
    for c in (32, 64):
        src = torch.randn(n, c, device=dev)
        t = timeit(lambda: native.segment_max(src, offs, order, m))
        res['segment_max_c%d' % c] = dict(ms=t * 1e3, GBps=(4 * c * n + 4 * n + 8 * c * m) / t / 1e9)
    feats = torch.randn(m, 32, device=dev)
    for dt, s in ((torch.float32, 4), (torch.bfloat16, 2)):
        t = timeit(lambda: native.pillar_scatter(feats, c2p, dt), iters=50)
        alg = 32 * s * n_cells + 32 * 4 * m + 4 * m
        res['pillar_scatter_%s' % str(dt).split('.')[-1]] = dict(ms=t * 1e3, GBps=alg / t / 1e9, alg_MB=alg / 1e6, M=m)
        cv = native.pillar_scatter(feats, c2p, dt)
        t = timeit(lambda: native.gather_rows(cv, cell), iters=50)
        res['pillar_gather_%s' % str(dt).split('.')[-1]] = dict(ms=t * 1e3, GBps=(32 * s * m * 2 + 4 * m) / t / 1e9)
    k = n // 10
    fm = torch.randn(1, ny, nx, 64, device=dev)
    kp = p3[:k].contiguous()
    bi = torch.zeros(k, dtype=torch.int32, device=dev)
    t = timeit(lambda: native.bilinear_gather(fm, kp, bi, 36.0, 36.0))
    res['bilinear_gather_c64'] = dict(ms=t * 1e3, GBps=(12 * k + 4 * 64 * 4 * k + 64 * 4 * k) / t / 1e9)
    go = torch.randn(k, 64, device=dev)
    t = timeit(lambda: native.bilinear_gather_backward(go, (1, ny, nx, 64), kp, bi, 36.0, 36.0))
    res['bilinear_gather_bwd_c64'] = dict(ms=t * 1e3)
    bev = torch.randn(1, T, ny, nx, 32, device=dev)
    inv = torch.eye(4, device=dev).repeat(1, T, 1, 1).contiguous()
    inv[:, :, 0, 3] = 0.7
    for dt, s in ((torch.float32, 4), (torch.bfloat16, 2)):
        b2 = bev.to(dt)
        t = timeit(lambda: native.bev_warp(b2, inv, 0.25, 0.25, -36.0, -36.0))
        res['bev_warp_%s' % str(dt).split('.')[-1]] = dict(ms=t * 1e3, GBps=2 * 32 * s * n_cells / t / 1e9)
    fi = p4[:, 3].to(torch.int32).contiguous()
    ts = inv.reshape(-1, 16).contiguous()
    t = timeit(lambda: native.rigid_transform(p3, fi, ts))
    res['rigid_transform'] = dict(ms=t * 1e3, GBps=28 * n / t / 1e9)
    for nn in (16384, 160000):
        a = torch.rand(1, nn, 3, device=dev) * 60
        b = torch.rand(1, nn, 3, device=dev) * 60
        t = timeit(lambda: native.chamfer_forward(a, b), iters=3, warmup=1)
        res['chamfer_%d' % nn] = dict(ms=t * 1e3, TFLOPs=2 * nn * nn * 8 / t / 1e12)
    for k_, v in res.items():
        print(k_, json.dumps({a: round(b, 4) if isinstance(b, float) else b for a, b in v.items()}))


if __name__ == '__main__':
    main()
