"""A/B of the full-resolution fp32x3 convolution layers between two builds of libpcacc_hip.so: this process loads ONE library (PCACC_LIB or the in-tree
one) and prints one JSON row per layer; tools/gpu_r05_conv_ab.sh runs it for both builds and prints them side by side.  Layers: the 'mixed' mode's
forward calls at the 4-sequence size (second bf16 output, maxima).  Usage: [PCACC_LIB=...] python tools/bench_conv_ab.py"""
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
    n, T = 20, 5
    cases = [('32->32 9 taps @288', 32, 32, 1, 288, False), ('64->32 two inputs @288', 64, 32, 1, 288, True), ('32->32 27 taps @288', 32, 32, 3, 288, False),
             ('32->64 @288', 32, 64, 1, 288, False), ('64->64 @144', 64, 64, 1, 144, False), ('128->128 @72', 128, 128, 1, 72, False)]
    for name, ci, co, kt, hw, two in cases:
        x = torch.randn(n, hw, hw, ci, device=dev)
        wshape = (co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)
        wt = torch.randn(*wshape, device=dev) / (3 * (ci * kt) ** 0.5)
        bias = torch.randn(co, device=dev)
        wf, _ = native.conv3x3_split_prepare_weights(wt)
        amax = native.absmax256(x)
        frames = T if kt == 3 else 1
        if two:
            a, b = x[..., :32].contiguous(), x[..., 32:].contiguous()
            f = lambda: native.conv3x3_split_cat(a, b, amax, wf, bias, True, want_bf16=True)
        else:
            f = lambda: native.conv3x3_split(x, wf, bias, frames, True, amax=amax, want_amax=True, want_bf16=True)
        y = f()[0]
        ts = sorted(timeit(f) for _ in range(3))
        print(json.dumps({'layer': name, 'us': [round(t, 1) for t in ts], 'sha': hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]}), flush=True)


if __name__ == '__main__':
    main()
