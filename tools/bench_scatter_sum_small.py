"""[r6] The few-row sums of the TubeNet (pcacc_scatter_sum_small: 320 k rows into 400 slots, c = 1 / 4 / 16) -- time per call and the result against a float64
sum, for the library in PCACC_LIB or the in-tree one.  Usage: [PCACC_LIB=...] python tools/bench_scatter_sum_small.py"""
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
    for n, c, m in ((320000, 16, 400), (320000, 4, 400), (320000, 1, 400), (332420, 16, 84), (1000, 16, 400)):
        x = torch.randn(n, c, device=dev) * torch.rand(n, 1, device=dev) * 30
        idx = torch.randint(0, m, (n,), device=dev, dtype=torch.int32)
        idx[::97] = -1
        ref = torch.zeros(m + 1, c, dtype=torch.float64, device=dev).index_add_(0, torch.where(idx < 0, m, idx).long(), x.double())[:m]
        out = native.scatter_sum_small(x, idx, m)
        again = native.scatter_sum_small(x, idx, m)
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        ts = sorted(timeit(lambda: native.scatter_sum_small(x, idx, m), iters=50) for _ in range(3))
        print(json.dumps({'n': n, 'c': c, 'm': m, 'us': [round(t, 1) for t in ts], 'rel_err_vs_f64': err, 'same_bits_twice': bool(torch.equal(out, again))}), flush=True)


if __name__ == '__main__':
    main()
