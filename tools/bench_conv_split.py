"""Time the fp32x3 (split-bf16) 3x3 convolution kernels -- forward, data gradient, weight gradient -- on the layer shapes of the model,
next to the bf16 kernels on the same shapes.  Usage: python tools/bench_conv_split.py [--batch 4] [--bf16 1]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402
from bench_conv import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--bf16', type=int, default=1)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    B, T = a.batch, 5
    shapes = [  # (name, n_img, frames, H, W, c_in, c_out, kt)
        ('unet 32->32 @288', B * T, 1, 288, 288, 32, 32, 1), ('stpn temporal 3x32->32 @288', B * T, T, 288, 288, 32, 32, 3),
        ('unet 32->64 @144', B * T, 1, 144, 144, 32, 64, 1), ('unet 64->64 @144', B * T, 1, 144, 144, 64, 64, 1),
        ('unet 64->128 @72', B * T, 1, 72, 72, 64, 128, 1), ('unet 128->128 @72', B * T, 1, 72, 72, 128, 128, 1),
        ('unet 256->256 @36', B * T, 1, 36, 36, 256, 256, 1), ('unet 512->512 @18', B * T, 1, 18, 18, 512, 512, 1),
        ('unet up 64->32 @288', B * T, 1, 288, 288, 64, 32, 1), ('ego head 32->64 @288', B * T, 1, 288, 288, 32, 64, 1),
        ('ego head 64->64 @288', B * T, 1, 288, 288, 64, 64, 1),
        ('stpn 32->64 @288', B, 1, 288, 288, 32, 64, 1), ('stpn 64->64 @288', B, 1, 288, 288, 64, 64, 1),
        ('stpn up 128->64 @288', B, 1, 288, 288, 128, 64, 1), ('stpn 256->128 @72', B, 1, 72, 72, 256, 128, 1),
        ('unet up 512->256 @36', B * T, 1, 36, 36, 512, 256, 1), ('unet up 256->128 @72', B * T, 1, 72, 72, 256, 128, 1),
        ('unet up 128->64 @144', B * T, 1, 144, 144, 128, 64, 1), ('unet 256->512 @18', B * T, 1, 18, 18, 256, 512, 1),
        ('stpn 128->128 @36', B, 1, 36, 36, 128, 128, 1), ('stpn 256->256 @18', B, 1, 18, 18, 256, 256, 1)]
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    for name, n, frames, h, w, ci, co, kt in shapes:
        x = torch.randn(n, h, w, ci, device=dev)
        gy = torch.randn(n, h, w, co, device=dev)
        y = torch.randn(n, h, w, co, device=dev)
        wshape = (co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)
        wt = torch.randn(*wshape, device=dev) / (3 * (ci * kt) ** 0.5)
        bias = torch.randn(co, device=dev)
        wf, wb = native.conv3x3_split_prepare_weights(wt)
        flops = 2.0 * n * h * w * co * ci * 9 * kt
        t_f = timeit(lambda: native.conv3x3_split(x, wf, bias, frames, True))
        t_d = timeit(lambda: native.conv3x3_split(gy, wb, None, frames, False, mask=y))
        if kt == 3:
            t_w = timeit(lambda: [native.conv3x3_wgrad_split(gy, x, frames, dt, mask=y) for dt in (-1, 0, 1)])
        else:
            t_w = timeit(lambda: native.conv3x3_wgrad_split(gy, x, mask=y))
        row = {'layer': name, 'fwd_us': round(t_f, 1), 'dgrad_us': round(t_d, 1), 'wgrad_us': round(t_w, 1),
               'fwd_TF_eff': round(flops / t_f / 1e6, 1), 'fwd_mfma_frac': round(3 * flops / t_f / 1e6 / 2500, 3),
               'wgrad_mfma_frac': round(3 * flops / t_w / 1e6 / 2500, 3), 'hbm_floor_us': round(n * h * w * (ci + co) * 4.0 / 8e6, 1)}
        if a.bf16:
            xb = x.to(torch.bfloat16)
            wp = native.conv3x3_prepare_weights(wt)
            row['bf16_fwd_us'] = round(timeit(lambda: native.conv3x3(xb, wp, bias, frames, True)), 1)
        tot['fwd'] += t_f
        tot['dgrad'] += t_d
        tot['wgrad'] += t_w
        print(json.dumps(row), flush=True)
    print(json.dumps({'sum_us': {k: round(v, 1) for k, v in tot.items()}}))


if __name__ == '__main__':
    main()
