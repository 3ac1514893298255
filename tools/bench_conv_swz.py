"""27-tap 32 -> 32 / 64 temporal convolution (models/stpn.py Conv3D layers) at the bench shape: padded LDS rows (96 KB: one 8-wave workgroup
per CU) against unpadded XOR-swizzled rows (77 KB: two per CU), tiles frame by frame against frame-fastest; results compared bit for bit.
Usage: python tools/bench_conv_swz.py [batch=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcaccumulation_amd import native
from bench_conv import timeit

dev = torch.device('cuda:0')
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 5
KNOBS = ('PCACC_CONV_SWZ_OFF', 'PCACC_CONV_FRAME_MAJOR')
for kt, co in ((3, 32), (3, 64)):
    x = torch.randn(B * T, 288, 288, 32, device=dev).to(torch.bfloat16)
    wt = torch.randn(*((co, 32, 3, 3, 3) if kt == 3 else (co, 32, 3, 3)), device=dev) / (3 * (32 * kt) ** 0.5)
    bias = torch.randn(co, device=dev)
    wp = native.conv3x3_prepare_weights(wt)
    fr = T if kt == 3 else 1
    ref = None
    variants = [('padded rows (1 WG/CU), frame-major tiles', {'PCACC_CONV_SWZ_OFF': '1', 'PCACC_CONV_FRAME_MAJOR': '1'}),
                ('swizzled rows (2 WG/CU), frame-major tiles', {'PCACC_CONV_FRAME_MAJOR': '1'}),
                ('swizzled rows (2 WG/CU), frame-fastest tiles', {})]
    for name, env in variants + variants:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        native.reload_switches()
        y = native.conv3x3(x, wp, bias, fr, True)
        ts = [round(timeit(lambda: native.conv3x3(x, wp, bias, fr, True), iters=50), 1) for _ in range(3)]
        ref = y if ref is None else ref
        print('%d-tap 32->%d @288 x %d frames  %-46s %s us  identical=%s' % (9 * kt, co, B * T, name, ts, torch.equal(y, ref)), flush=True)
    for k in KNOBS:
        os.environ.pop(k, None)
