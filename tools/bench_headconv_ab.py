"""A/B of the head convolution kernels (head_conv.hip) between two builds of libpcacc_hip.so (PCACC_LIB or the in-tree one): the fg / bg head of the step
(32 -> 2 at 20 x 288^2, bf16 and fp32 rows) and a 64-channel / 4-output shape.  Usage: [PCACC_LIB=...] python tools/bench_headconv_ab.py"""
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
    sha = lambda t: hashlib.sha256(t.float().cpu().numpy().tobytes()).hexdigest()[:12]
    for n, hw, ci, co in ((20, 288, 32, 2), (4, 288, 64, 2), (20, 288, 32, 4)):
        for dt in (torch.bfloat16, torch.float32):
            x = torch.randn(n, hw, hw, ci, device=dev).to(dt)
            w = torch.randn(co, ci, 3, 3, device=dev) * 0.1
            b = torch.randn(co, device=dev)
            dy = torch.randn(n, hw, hw, co, device=dev)
            row = {'shape': '%d x %d^2, %d -> %d, %s' % (n, hw, ci, co, str(dt).split('.')[-1])}
            f = lambda: native.head_conv3x3_forward(x, w, b)
            g = lambda: native.head_conv3x3_wgrad(dy, x)
            d = lambda: native.head_conv3x3_dgrad(dy, w, ci, dt)
            row['fwd_us'] = round(min(timeit(f) for _ in range(3)), 1)
            row['fwd_sha'] = sha(f())
            row['wgrad_us'] = round(min(timeit(g) for _ in range(3)), 1)
            row['wgrad_sha'] = sha(g()[0]) + '/' + sha(g()[1])
            row['dgrad_us'] = round(min(timeit(d) for _ in range(3)), 1)
            print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
