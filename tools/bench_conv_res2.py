"""Round 5, verdict item 3: the c_out = 32 fp32x3 layers as two 256-thread workgroups per CU (default) against the one-workgroup resident kernel of
round 4 (PCACC_CONV_RES2=0), isolated launches at the 4-sequence size with the 'mixed' mode's second (bf16) output.
Usage: python tools/bench_conv_res2.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402
from bench_conv import timeit  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    n, T, h, w = 20, 5, 288, 288
    cases = [('32->32 9 taps', 32, 1), ('64->32 two inputs, 9 taps', 64, 1), ('32->32 27 taps', 32, 3)]
    for name, ci, kt in cases:
        x = torch.randn(n, h, w, ci, device=dev)
        wshape = (32, ci, 3, 3, 3) if kt == 3 else (32, ci, 3, 3)
        wt = torch.randn(*wshape, device=dev) / (3 * (ci * kt) ** 0.5)
        bias = torch.randn(32, device=dev)
        wf, _ = native.conv3x3_split_prepare_weights(wt)
        amax = native.absmax256(x)
        frames = T if kt == 3 else 1
        row = {'layer': name}
        out = {}
        for two in ('1', '0', '1', '0'):
            os.environ['PCACC_CONV_RES2'] = two
            native.reload_switches()
            if ci == 64:
                a, b = x[..., :32].contiguous(), x[..., 32:].contiguous()
                f = lambda: native.conv3x3_split_cat(a, b, amax, wf, bias, True, want_bf16=True)
            else:
                f = lambda: native.conv3x3_split(x, wf, bias, frames, True, amax=amax, want_amax=True, want_bf16=True)
            out[two] = f()[0]
            row.setdefault('two_per_cu_us' if two == '1' else 'one_per_cu_us', []).append(round(timeit(f), 1))
        row['bit_identical'] = bool(torch.equal(out['1'], out['0']))
        flops = 2.0 * n * h * w * 32 * ci * 9 * kt * (13.0 / 15.0 if kt == 3 else 1.0)
        row['mfma_frac_two_per_cu'] = round(3 * flops / min(row['two_per_cu_us']) / 1e6 / 2500, 3)
        row['algorithmic_MB'] = round(n * h * w * (ci * 4 + 32 * 4 + 32 * 2) / 1e6, 1)
        row['GBps_two_per_cu'] = round(n * h * w * (ci * 4 + 32 * 4 + 32 * 2) / min(row['two_per_cu_us']) / 1e3, 0)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
