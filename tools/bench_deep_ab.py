"""A/B of the deep bf16 layers (conv_deep.hip: strip forward / masked data gradient, strip weight gradient) between two builds of libpcacc_hip.so: this
process loads ONE library (PCACC_LIB or the in-tree one) and prints one JSON row per layer with a digest of the results.
Usage: [PCACC_LIB=...] python tools/bench_deep_ab.py"""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402
from bench_conv import timeit  # noqa: E402


def sha(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.float().cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for name, n, hw, ci, co in (('128->128 @72', 20, 72, 128, 128), ('256->256 @36', 20, 36, 256, 256), ('512->512 @18', 20, 18, 512, 512),
                                ('256->128 @72', 20, 72, 256, 128), ('128->64 @144', 20, 144, 128, 64), ('128->128 @36 x4', 4, 36, 128, 128)):
        x = torch.randn(n, hw, hw, ci, device=dev).to(torch.bfloat16)
        gy = torch.randn(n, hw, hw, co, device=dev).to(torch.bfloat16)
        y = torch.randn(n, hw, hw, co, device=dev).to(torch.bfloat16)
        wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
        bias = torch.randn(co, device=dev)
        wp, wpt = native.conv3x3_prepare_weights(wt), native.conv3x3_prepare_weights(wt, transpose=True)
        f_fwd = lambda: native.conv3x3(x, wp, bias, 1, True)
        f_dg = lambda: native.conv3x3(gy, wpt, None, 1, False, mask=y)
        row = {'layer': name, 'fwd_us': round(min(timeit(f_fwd) for _ in range(3)), 1), 'masked_dgrad_us': round(min(timeit(f_dg) for _ in range(3)), 1)}
        outs = [f_fwd(), f_dg()]
        if native.conv3x3_wgrad_deep_supported(hw, hw, ci, co):
            f_wg = lambda: native.conv3x3_wgrad_deep(gy, x, mask=y)
            row['masked_wgrad_us'] = round(min(timeit(f_wg) for _ in range(3)), 1)
            f_wg0 = lambda: native.conv3x3_wgrad_deep(gy, x)
            row['wgrad_us'] = round(min(timeit(f_wg0) for _ in range(3)), 1)
            outs += list(f_wg()) + list(f_wg0())
        row['sha'] = sha(*outs)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
