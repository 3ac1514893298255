"""A/B of the few-input row kernel (mlp.hip: rows_linear_fewk_kernel) between two builds of libpcacc_hip.so (PCACC_LIB or the in-tree one): the pillar encoder's
position layer of the 'mixed' mode (3.2 M x 9 -> 64 with the bf16 second output and maxima) and the plain fp32 / bf16-output calls.
Usage: [PCACC_LIB=...] python tools/bench_fewk_ab.py"""
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
    for rows, k, n in ((3200000, 9, 64), (320000, 3, 32), (335444, 2, 128), (3200000, 4, 32)):
        x = torch.randn(rows, k, device=dev)
        w = torch.randn(n, k, device=dev)
        b = torch.randn(n, device=dev)
        f1 = lambda: native.rows_linear_few_dual(x, w, b, None, False, True)
        f2 = lambda: native.rows_linear(x, w, b, None, False, True, out_dtype=torch.bfloat16)
        f3 = lambda: native.rows_linear(x, w, b, None, False, True)
        row = {'layer': '%d x %d -> %d' % (rows, k, n)}
        for name, f in (('dual', f1), ('bf16_out', f2), ('f32', f3)):
            y = f()
            y = y[0] if isinstance(y, (tuple, list)) else y
            row[name + '_us'] = round(min(timeit(f) for _ in range(3)), 1)
            row[name + '_sha'] = sha(y)
        print(json.dumps(row), flush=True)
    # the weight gradients of the same layers (mlp.hip: rows_wgrad_few_kernel): dY bf16 + ReLU mask (the 'mixed' / bf16 backward), dY fp32 unmasked
    for rows, k, n in ((3200000, 9, 64), (320000, 3, 32), (335444, 128, 2), (3200000, 4, 32)):
        x = torch.randn(rows, k, device=dev)
        dy = torch.randn(rows, n, device=dev)
        mk = torch.randn(rows, n, device=dev)
        row = {'wgrad': '%d x (%d <- %d)' % (rows, k, n)}
        cases = (('bf16_masked', lambda: native.rows_wgrad(dy.bfloat16(), x, dy_mask=mk.bfloat16())),
                 ('f32', lambda: native.rows_wgrad(dy, x)),
                 ('f32_masked', lambda: native.rows_wgrad(dy, x, dy_mask=mk)))
        for name, f in cases:
            if name == 'bf16_masked':
                d16, m16 = dy.bfloat16(), mk.bfloat16()
                f = lambda: native.rows_wgrad(d16, x, dy_mask=m16)
            try:
                y = f()
            except Exception as e:
                row[name] = repr(e)[:60]
                continue
            row[name + '_us'] = round(min(timeit(f) for _ in range(3)), 1)
            row[name + '_sha'] = sha(y)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
