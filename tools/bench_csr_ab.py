"""A/B of the CSR build (segment.hip: pcacc_csr_build) between two builds of libpcacc_hip.so (PCACC_LIB or the in-tree one): the step's point -> pillar map
(3.2 M points, ~1.17 M pillars), the bilinear backward's cell keys (336 k points uniform / in 20 boxes per map over 4 x 288^2 cells) and a short-segment case.
Digests of offsets and order (order is ascending inside segments of <= 64 points: deterministic there).  Usage: [PCACC_LIB=...] python tools/bench_csr_ab.py"""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402
from bench_conv import timeit  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12]
    cases = []
    m = 1170000
    cases.append(('3.2 M points -> 1.17 M pillars (uniform)', torch.randint(0, m, (3200000,), device=dev, dtype=torch.int32), m))
    cells = 4 * 288 * 288
    cases.append(('336 k points -> 4 x 288^2 cells (uniform)', torch.randint(0, cells, (336000,), device=dev, dtype=torch.int32), cells + 1))
    box = torch.randint(0, 80, (336000,), device=dev)
    centre = torch.randint(20, 260, (80, 2), device=dev)
    xy = centre[box] + torch.randint(0, 16, (336000, 2), device=dev)[:, :] // torch.tensor([1, 2], device=dev)
    key = ((box // 20) * 288 + xy[:, 1]) * 288 + xy[:, 0]
    cases.append(('336 k points -> 4 x 288^2 cells (20 boxes per map: ~30 points per cell)', key.to(torch.int32), cells + 1))
    cases.append(('800 k points -> 300 k pillars', torch.randint(0, 300000, (800000,), device=dev, dtype=torch.int32), 300000))
    for name, p2v, mm in cases:
        f = lambda: native.csr_build(p2v, mm)
        offs, order = f()
        cnt = torch.bincount(p2v.long(), minlength=mm)
        row = {'case': name, 'us': round(min(timeit(f) for _ in range(3)), 1), 'offsets_sha': sha(offs), 'max_segment': int(cnt.max())}
        if int(cnt.max()) <= 64:
            row['order_sha'] = sha(order)
        else:                                                       # arbitrary order inside the long segments: compare the sorted segments
            row['order_sorted_ok'] = bool((torch.sort(p2v[order.long()].long(), stable=True).values == p2v[order.long()].long()).all()) and \
                sorted(order.cpu().tolist()) == list(range(p2v.numel()))
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
