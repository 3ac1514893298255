#!/usr/bin/env python
"""pcacc_segment_max_canvas (pooling + pillar scatter in one pass) at the step's size: variants (PCACC_SCATTER_VARIANT: cells in flight per lane group, cache
policy of the row loads) on uniform and on LiDAR-shaped pillar sizes (a few crowded pillars, two thirds of the cells empty), warm and behind 1 GiB of
streamed lines; results compared bit for bit with the default.  Usage: python tools/bench_fused_canvas.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pcaccumulation_amd import native  # noqa: E402


def case(kind, dev):
    n_cells, c = 20 * 288 * 288, 32
    g = torch.Generator(device=dev).manual_seed(1)
    if kind == 'uniform':
        m, n = 1_169_433, 3_200_000
        p2v = torch.cat([torch.arange(m, device=dev), torch.randint(0, m, (n - m,), device=dev, generator=g)])
    else:                                                        # LiDAR-shaped: a third of the cells occupied, pillar sizes ~ 1 / r (a long tail)
        m, n = 560_000, 3_200_000
        w = 1.0 / (torch.rand(m, device=dev, generator=g) * 0.98 + 0.02)
        extra = torch.multinomial(w / w.sum(), n - m, replacement=True, generator=g)
        p2v = torch.cat([torch.arange(m, device=dev), extra])
    p2v = p2v[torch.randperm(n, device=dev, generator=g)].to(torch.int32)
    c2p = torch.full((n_cells,), -1, dtype=torch.int32, device=dev)
    c2p[torch.randperm(n_cells, device=dev, generator=g)[:m].sort().values] = torch.arange(m, dtype=torch.int32, device=dev)
    src = torch.randn(n, c, device=dev, generator=g)
    offs, order = native.csr_build(p2v, m)
    return src, offs, order, m, c2p, n_cells, c, n


def main():
    dev = torch.device('cuda:0')
    flush = torch.zeros(256 * 1024 * 1024, device=dev)
    for kind in ('uniform', 'lidar'):
        src, offs, order, m, c2p, n_cells, c, n = case(kind, dev)
        alg = bench.fused_alg_bytes(n_cells, c, m, n)
        ref = None
        for variant in ('', 'c', 'b', 'e', 'd'):
            if variant:
                os.environ['PCACC_SCATTER_VARIANT'] = variant
            else:
                os.environ.pop('PCACC_SCATTER_VARIANT', None)
            native.reload_switches()
            out = native.segment_max_canvas(src, offs, order, m, c2p)
            if ref is None:
                ref = out
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            res = {}
            for cold in (False, True):
                native.scatter_timer = []
                for _ in range(10):
                    if cold:
                        flush.add_(1.0)
                    native.segment_max_canvas(src, offs, order, m, c2p)
                torch.cuda.synchronize()
                us = sorted(t[0].elapsed_us() for t in native.scatter_timer)
                native.scatter_timer = None
                res['cold' if cold else 'warm'] = us[len(us) // 2]
            print('%-8s variant %-7s warm %6.1f us (%.3f of 8 TB/s)  cold %6.1f us (%.3f)  %s' % (
                kind, repr(variant) if variant else 'default', res['warm'], alg / res['warm'] / 1e3 / 8000.0, res['cold'], alg / res['cold'] / 1e3 / 8000.0,
                'identical' if same else 'RESULTS DIFFER'), flush=True)
    os.environ.pop('PCACC_SCATTER_VARIANT', None)
    native.reload_switches()


if __name__ == '__main__':
    main()
