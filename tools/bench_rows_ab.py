"""A/B of the fp32x3 row-linear kernels (mlp_split.hip) between two builds of libpcacc_hip.so: this process loads ONE library (PCACC_LIB or the in-tree one).
Usage: [PCACC_LIB=...] python tools/bench_rows_ab.py"""
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
    for rows, k, n, relu in ((335444, 128, 128, True), (320000, 64, 128, True), (335444, 128, 64, False), (3200000, 32, 32, True), (1169433, 64, 32, True), (429567, 128, 128, True)):
        x = torch.randn(rows, k, device=dev)
        ax = native.absmax256(x)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev)
        f = lambda: native.rows_linear_split(x, ax, w, b, None, False, relu)
        y = f()
        y = y[0] if isinstance(y, (tuple, list)) else y
        ts = sorted(timeit(f) for _ in range(3))
        print(json.dumps({'layer': '%d x %d -> %d' % (rows, k, n), 'us': [round(t, 1) for t in ts], 'GBps': round(rows * (k + n) * 4 / ts[0] / 1e3), 'sha': hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]}), flush=True)


if __name__ == '__main__':
    main()
