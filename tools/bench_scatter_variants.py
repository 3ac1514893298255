"""Round 5, verdict item 6: variants of the bf16 pillar-scatter launch (pieces per lane, streaming cache policy, workgroups per CU) at the 4-sequence
size, pillar ids in cell order (as the model numbers them), three cache states: back to back (warm), after 1 GiB WRITTEN (round 4's `cold_cache`:
cold operands AND 256 MiB of dirty lines in the Infinity Cache that this kernel's traffic has to push out), after 1 GiB READ (cold operands, clean caches).
Usage: python tools/bench_scatter_variants.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402


def run(c2p, feats, flush, n=30):
    for _ in range(5):
        native.pillar_scatter(feats, c2p, torch.bfloat16)
    native.scatter_timer = []
    try:
        for _ in range(n):
            if flush is not None:
                flush()
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
    c2p = torch.full((n_cells,), -1, dtype=torch.int32, device=dev)
    c2p[occupied.sort().values] = torch.arange(m, dtype=torch.int32, device=dev)
    big = torch.zeros(256 * 1024 * 1024, device=dev)
    ref = native.pillar_scatter(feats, c2p, torch.bfloat16).clone()
    flushes = (('warm', None), ('written-flush', lambda: big.add_(1.0)), ('read-flush', lambda: big.sum()))
    variants = [('0', 0)] + [(v, b) for v in '2567' for b in (8, 16)]
    for v, b in variants:
        os.environ['PCACC_SCATTER_VARIANT'] = v
        if b:
            os.environ['PCACC_SCATTER_BLOCKS'] = str(b)
        else:
            os.environ.pop('PCACC_SCATTER_BLOCKS', None)
        native.reload_switches()
        assert torch.equal(native.pillar_scatter(feats, c2p, torch.bfloat16), ref)
        row = {'variant': v, 'blocks_per_cu': b or 8}
        for name, fl in flushes:
            us = run(c2p, feats, fl)
            med = us[len(us) // 2]
            row[name] = {'median_us': round(med, 2), 'min_us': round(us[0], 2), 'frac_of_8TBps': round(alg / med / 1e3 / 8000, 3)}
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
