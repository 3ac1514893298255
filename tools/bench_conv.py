"""Time the MFMA 3x3 convolution against the library (MIOpen through torch) on the layer shapes of the model.
Usage: python tools/bench_conv.py [--batch 4]"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native, ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3            # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--lib', type=int, default=1)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    B, T = a.batch, 5
    shapes = [  # (name, n_img, frames, H, W, c_in, c_out, kt)
        ('unet 32->32 @288', B * T, 1, 288, 288, 32, 32, 1), ('unet 32->64 @144', B * T, 1, 144, 144, 32, 64, 1),
        ('unet 64->64 @144', B * T, 1, 144, 144, 64, 64, 1), ('unet 128->128 @72', B * T, 1, 72, 72, 128, 128, 1),
        ('unet 256->256 @36', B * T, 1, 36, 36, 256, 256, 1), ('unet 512->512 @18', B * T, 1, 18, 18, 512, 512, 1),
        ('unet up 64->32 @288', B * T, 1, 288, 288, 64, 32, 1), ('ego head 32->64 @288', B * T, 1, 288, 288, 32, 64, 1),
        ('ego head 64->64 @288', B * T, 1, 288, 288, 64, 64, 1), ('stpn temporal 3x32->32 @288', B * T, T, 288, 288, 32, 32, 3),
        ('stpn 32->64 @288', B, 1, 288, 288, 32, 64, 1), ('stpn 64->64 @288', B, 1, 288, 288, 64, 64, 1),
        ('stpn up 128->64 @288', B, 1, 288, 288, 128, 64, 1), ('stpn 256->128 @72', B, 1, 72, 72, 256, 128, 1),
        ('unet up 512->256 @36', B * T, 1, 36, 36, 512, 256, 1), ('unet up 256->128 @72', B * T, 1, 72, 72, 256, 128, 1),
        ('unet up 128->64 @144', B * T, 1, 144, 144, 128, 64, 1), ('unet 256->512 @18', B * T, 1, 18, 18, 256, 512, 1),
        ('stpn 128->128 @36', B, 1, 36, 36, 128, 128, 1), ('stpn 256->256 @18', B, 1, 18, 18, 256, 256, 1)]
    for name, n, frames, h, w, ci, co, kt in shapes:
        x = torch.randn(n, h, w, ci, device=dev).to(torch.bfloat16)
        wshape = (co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)
        wt = torch.randn(*wshape, device=dev) / (3 * (ci * kt) ** 0.5)
        bias = torch.randn(co, device=dev)
        wp = native.conv3x3_prepare_weights(wt)
        t_k = timeit(lambda: native.conv3x3(x, wp, bias, frames, True))
        flops = 2.0 * n * h * w * co * ci * 9 * kt
        byts = n * h * w * (ci + co) * 2.0
        row = {'layer': name, 'mfma_us': round(t_k, 1), 'TFLOPs': round(flops / t_k / 1e6, 1), 'hbm_floor_us': round(byts / 6.3e6, 1)}
        if a.lib:
            if kt == 3:
                xs = ops._stack_frames(x, frames).permute(0, 3, 1, 2)
                w2 = wt.permute(0, 2, 1, 3, 4).reshape(co, 3 * ci, 3, 3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                t_l = timeit(lambda: torch.relu_(F.conv2d(ops._stack_frames(x, frames).permute(0, 3, 1, 2), w2, bias.to(torch.bfloat16), padding=1)))
            else:
                xs = x.permute(0, 3, 1, 2)
                w2 = wt.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                t_l = timeit(lambda: torch.relu_(F.conv2d(xs, w2, bias.to(torch.bfloat16), padding=1)))
            row['lib_us'] = round(t_l, 1)
        if kt == 1 and native.conv3x3_wgrad_deep_supported(h, w, ci, co):
            gy = torch.randn(n, h, w, co, device=dev).to(torch.bfloat16)
            row['wgrad_us'] = round(timeit(lambda: native.conv3x3_wgrad_deep(gy, x)), 1)
            row['wgrad_TFLOPs'] = round(flops / row['wgrad_us'] / 1e6, 1)
            if a.lib:
                w2 = wt.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                row['lib_wgrad_us'] = round(timeit(lambda: torch.ops.aten.convolution_backward(
                    gy.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w2, [co], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])), 1)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
