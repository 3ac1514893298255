"""A/B of the BatchNorm row passes (bn.hip) between two builds of libpcacc_hip.so (PCACC_LIB or the in-tree one): training-mode forward (statistics pass +
apply pass) on the step's shapes, fp32 and bf16 rows, plain and with the bf16 copy ('dual').  Usage: [PCACC_LIB=...] python tools/bench_bn_ab.py"""
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
    for rows, c in ((20 * 288 * 288, 32), (336852, 128), (4 * 288 * 288, 64)):
        for dt in (torch.float32, torch.bfloat16):
            x = (torch.randn(rows, c, device=dev) * 2 + 0.3).to(dt)
            g = torch.rand(c, device=dev) + 0.5
            b = torch.randn(c, device=dev)
            f = lambda: native.bn_rows_forward(x, g, b, 1e-5, 0.1, None, None, relu=True)
            y, mean, invstd = f()
            row = {'rows': '%d x %d %s' % (rows, c, str(dt).split('.')[-1]), 'fwd_us': round(min(timeit(f) for _ in range(3)), 1),
                   'sha': sha(y) + '/' + sha(mean) + '/' + sha(invstd)}
            if dt == torch.float32:
                d = lambda: native.bn_rows_forward_dual(x, g, b, 1e-5, 0.1, None, None, relu=True)
                out = d()
                row['dual_us'] = round(min(timeit(d) for _ in range(3)), 1)
                row['dual_sha'] = sha(out[0]) + '/' + sha(out[1]) + '/' + sha(out[3])
            print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
