"""[r6] Does the Infinity Cache carry a layer's result to the next layer when the batch is walked in chunks of images?  Two full-resolution fp32x3 convolutions
(32 -> 32, ReLU, bf16 shadow, maxima -- the 'mixed' mode's forward calls) on 20 x 288^2: layer A then layer B over the whole batch (B reads 212 MB that A wrote
212 + 106 MB ago: from HBM) against chunk by chunk (A, B on 10 / 5 / 4 / 2 / 1 images at a time: B's input is 106 .. 10 MB old).  Also a chain of four layers.
Prints us per chain.  Usage: python tools/bench_chunked_chain.py"""
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
    n, hw, c = 20, 288, 32
    x = torch.randn(n, hw, hw, c, device=dev)
    ws = []
    for _ in range(4):
        wt = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
        ws.append((native.conv3x3_split_prepare_weights(wt)[0], torch.randn(c, device=dev) * 0.1))
    amax_x = native.absmax256(x)

    def chain(xs, amax, depth):
        y = xs
        for d in range(depth):
            y, amax, _ = native.conv3x3_split(y, ws[d][0], ws[d][1], 1, True, amax=amax, want_amax=True, want_bf16=True)
        return y

    for depth in (2, 4):
        for chunk in (20, 10, 5, 4, 2, 1):
            def f():
                for i in range(0, n, chunk):
                    chain(x[i:i + chunk], amax_x, depth)
            ts = sorted(timeit(f, iters=10) for _ in range(3))
            print(json.dumps({'layers': depth, 'images_per_chunk': chunk, 'launches': depth * (n // chunk), 'us': [round(t, 1) for t in ts],
                              'us_per_layer': round(ts[0] / depth, 1)}), flush=True)


if __name__ == '__main__':
    main()
