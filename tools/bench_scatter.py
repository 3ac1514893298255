"""The pillar-scatter launch of bench.py's roofline object in isolation: the canvas of a 4-sequence step (20 x 288 x 288 cells,
32 channels, bf16) filled from 1.17 M bf16 feature rows, timed with events attached to the dispatch, 50 launches back to back --
first with the features in pillar-id order as the model has them (random with respect to the cells on uniform synthetic
input), then with pillar ids in cell order (sequential reads).  Usage: python tools/bench_scatter.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402


def run(c2p, feats, n=50, flush=None):
    """feats: bf16 rows (the bf16 compute mode) or f32 rows (converted on the way)."""
    for _ in range(5):
        native.pillar_scatter(feats, c2p, torch.bfloat16)
    native.scatter_timer = []
    try:
        for _ in range(n):
            if flush is not None:
                flush.add_(1.0)                                # 1 GiB read + written in between: nothing of the table stays cached
            native.pillar_scatter(feats, c2p, torch.bfloat16)
        torch.cuda.synchronize()
        us = sorted(t[0].elapsed_us() for t in native.scatter_timer)
    finally:
        native.scatter_timer = None
    return us


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    n_cells, m, c = 20 * 288 * 288, 1_169_433, 32
    feats = torch.randn(m, c, device=dev).to(torch.bfloat16)
    occupied = torch.randperm(n_cells, device=dev)[:m]
    alg = n_cells * c * 2 + m * c * 2 + 4 * m               # SURVEY 8d with s = 2
    for name, ids in (('pillar ids in first-touch (random) order', torch.arange(m, dtype=torch.int32, device=dev)),
                      ('pillar ids in cell order', None)):
        c2p = torch.full((n_cells,), -1, dtype=torch.int32, device=dev)
        if ids is None:
            c2p[occupied.sort().values] = torch.arange(m, dtype=torch.int32, device=dev)
        else:
            c2p[occupied] = ids
        for flush in (None, torch.zeros(256 * 1024 * 1024, device=dev)):
            us = run(c2p, feats, flush=flush)
            med = us[len(us) // 2]
            print(json.dumps({'case': name + (', caches flushed between launches' if flush is not None else ', back to back'), 'launches': len(us), 'median_us': round(med, 2), 'min_us': round(us[0], 2), 'max_us': round(us[-1], 2),
                              'algorithmic_MB': round(alg / 1e6, 1), 'GBps': round(alg / med / 1e3, 1), 'frac_of_8TBps': round(alg / med / 1e3 / 8000, 3)}))


if __name__ == '__main__':
    main()
