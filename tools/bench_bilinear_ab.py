"""A/B of the sorted bilinear-gather backward (bilinear.hip) between two builds of libpcacc_hip.so (PCACC_LIB or the in-tree one), WARM (a benchmark loop over
the same buffers: they sit in the 256 MB Infinity Cache) and COLD (1 GiB streamed through the caches before every launch: what the kernel meets in the step).
Times are of the whole native call (base cells, CSR build, preparation pass, the per-cell sums); the last kernel alone is in the rocprofv3 traces.
Usage: [PCACC_LIB=...] python tools/bench_bilinear_ab.py"""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    sha = lambda t: hashlib.sha256(t.float().cpu().numpy().tobytes()).hexdigest()[:12]
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    shapes = ((4, 288, 64, 335444, torch.float32, torch.bfloat16), (4, 288, 64, 335444, torch.bfloat16, torch.bfloat16), (4, 288, 32, 320000, torch.float32, torch.float32))
    for n_maps, hw, c, k, gdt, odt in shapes:
        for boxes in (0, 20):
            pts = torch.rand(k, 3, device=dev) * 2 - 1                              # normalised positions, uniform over the map
            if boxes:                                                               # the step's case: the points of `boxes` objects of 4 m x 2 m per map (72 m maps)
                centre = torch.rand(n_maps * boxes, 2, device=dev) * 1.6 - 0.8
                which = torch.randint(0, n_maps * boxes, (k,), device=dev)
                pts[:, :2] = centre[which] + (torch.rand(k, 2, device=dev) - 0.5) * torch.tensor([4.0 / 36, 2.0 / 36], device=dev)
                midx = (which // boxes).to(torch.int32)
            else:
                midx = torch.randint(0, n_maps, (k,), device=dev, dtype=torch.int32)
            g = torch.randn(k, c, device=dev).to(gdt)
            f = lambda: native.bilinear_gather_backward_sorted(g, (n_maps, hw, hw, c), pts, midx, 1.0, 1.0, out_dtype=odt)
            row = {'case': '%d x %d^2 x %d, %d points %s, %s -> %s' % (n_maps, hw, c, k, ('in %d boxes per map' % boxes) if boxes else 'uniform',
                                                                       str(gdt).split('.')[-1], str(odt).split('.')[-1]), 'sha': sha(f())}
            for name, cold in (('warm', False), ('cold', True)):
                ts = []
                for _ in range(12):
                    if cold:
                        flush.add_(1.0)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    f()
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) * 1e3)
                ts = sorted(ts[2:])
                row[name + '_us_min'] = round(ts[0], 1)
                row[name + '_us_med'] = round(ts[len(ts) // 2], 1)
            print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
