"""Time the fused loss passes (include/pcacc.h L1, L2) at the sizes of one 4-sequence Waymo-geometry step -- 600 k occupied
pillars for the fg/bg loss, 250 k supervised points for the motion loss, 3.2 M points / 400 k foreground rows for the offset
loss -- forward + backward with HIP events, next to the element-wise torch formulation of the same terms on the same GPU and
the numpy oracle on the host.  Prints one JSON line per term with the achieved HBM rate against the algorithmic bytes.
Usage: python tools/bench_loss.py [--iters 20] [--no-host]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import ops  # noqa: E402
from pcaccumulation_amd.loss import lovasz_softmax_flat  # noqa: E402


def gpu_ms(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def torch_seg(z, y):
    """The element-wise formulation this row replaced (weighted CE written out + torch.sort Lovasz + counters)."""
    counts = torch.stack([(y == c).sum() for c in range(2)]).float() + 1e-20
    w = torch.clamp(torch.sqrt(counts.sum() / counts), 0, 50)
    keep = y != -1
    safe = torch.where(keep, y, torch.zeros_like(y))
    picked = torch.log_softmax(z, dim=1).gather(1, safe[:, None])[:, 0]
    wi = w[safe] * keep
    ce = -(wi * picked).sum() / wi.sum()
    return ce + lovasz_softmax_flat(torch.softmax(z, 1), y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--no-host', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for name, n_total, n in (('fb_seg (occupied pillars of 20 BEV maps)', 20 * 288 * 288, 600_000), ('mos_seg (foreground points)', 3_200_000, 250_000)):
        z = (torch.randn(n_total, 2, device=dev) * 2).requires_grad_(True)
        y = (torch.rand(n_total, device=dev) < 0.1).long()
        rows = torch.randperm(n_total, device=dev)[:n].sort().values
        wv = torch.tensor([1.0, 1.0], device=dev)

        def fused():
            terms, _ = ops.seg_loss(z, y, rows)
            z.grad = None
            torch.dot(terms, wv).backward()

        def plain():
            z.grad = None
            torch_seg(z.index_select(0, rows), y[rows]).backward()
        out = {'term': name, 'rows': n, 'fused_fwd_bwd_ms': round(gpu_ms(fused, a.iters), 3), 'torch_fwd_bwd_ms': round(gpu_ms(plain, a.iters), 3)}
        # algorithmic bytes: forward reads 8 B logits + 8 B label + 8 B row index, writes and sorts 2 x (key, payload) = 4 radix
        # passes x 16 B x 2 (read + write) per class-row, scan pass reads 16 B and writes 8 B per row; backward reads 16 + 8 + 8
        # and writes 8 B per row (+ the zero fill of the full gradient)
        fwd_bytes = n * (24 + 16 + 2 * 4 * 16 + 24)
        bwd_bytes = n * 40 + n_total * 8
        out['algorithmic_MB'] = round((fwd_bytes + bwd_bytes) / 1e6, 1)
        out['achieved_GBps'] = round((fwd_bytes + bwd_bytes) / out['fused_fwd_bwd_ms'] / 1e6, 1)
        if not a.no_host:
            import oracle
            zs, ys = z.detach()[rows].cpu().numpy(), y[rows].cpu().numpy()
            t0 = time.time()
            oracle.seg_loss(zs, ys)
            out['host_numpy_ms'] = round((time.time() - t0) * 1e3, 1)
        print(json.dumps(out), flush=True)
    n, k, T = 3_200_000, 240, 5
    rng = np.random.RandomState(0)
    pts = torch.randn(n, 3, device=dev) * 20
    tidx = torch.stack([torch.arange(n, device=dev) // (n // 4), torch.randint(0, T, (n,), device=dev)], 1)
    lab = torch.where(torch.rand(n, device=dev) < 0.12, torch.randint(1, k // 4, (n,), device=dev), torch.zeros(n, dtype=torch.long, device=dev))
    base = torch.arange(4, device=dev) * (k // 4)
    ego = torch.eye(4, device=dev).repeat(4, T, 1, 1)
    motion = torch.eye(4, device=dev).repeat(k, T, 1, 1)
    motion[:, :, :3, 3] = torch.randn(k, T, 3, device=dev)
    est = torch.randn(n, 2, device=dev, requires_grad=True)
    rows = torch.nonzero(lab > 0)[:, 0]
    wv = torch.tensor([1.0, 1.0, 0.0], device=dev)

    def fused_off():
        out, _ = ops.offset_loss(est, pts, tidx, lab, base, ego, motion, pts, rows)
        est.grad = None
        torch.dot(out, wv).backward()
    m = int(rows.numel())
    out = {'term': 'offset (GT reconstruction of all points, centres, foreground rows)', 'points': n, 'rows': m,
           'fused_fwd_bwd_ms': round(gpu_ms(fused_off, a.iters), 3)}
    total = n * (12 + 16 + 8) + m * (8 + 24 + 12 + 8 + 8) + m * (8 + 8 + 8 + 8) + n * 8
    out['algorithmic_MB'] = round(total / 1e6, 1)
    out['achieved_GBps'] = round(total / out['fused_fwd_bwd_ms'] / 1e6, 1)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
