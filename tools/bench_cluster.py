"""Time test-mode clustering (include/pcacc.h C1) against the host path it replaces (sparse_quantize + scikit-learn DBSCAN via
the oracle).  Synthetic scene: B samples x `--points` points, vehicles as dense boxes, `--moving` share predicted moving.
Usage: python tools/bench_cluster.py [--batch 4] [--points 160000] [--boxes 40] [--iters 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcaccumulation_amd import native  # noqa: E402


def scene(rng, n, boxes, spread=70.0):
    per = n // 3 // max(boxes, 1)
    pts, mos = [], []
    for _ in range(boxes):
        c = rng.uniform(-spread, spread, 2)
        size = rng.uniform([1.6, 3.5], [2.2, 5.5])
        u = rng.uniform(-0.5, 0.5, (per, 2)) * size
        side = rng.randint(0, 4, per)
        u[side == 0, 0] = -size[0] / 2
        u[side == 1, 1] = size[1] / 2
        pts.append(np.concatenate([u + c, rng.uniform(-1.5, 0.3, (per, 1))], 1))
        mos.append((rng.rand(per) < 0.95).astype(np.int64))
    rest = n - per * boxes
    pts.append(np.concatenate([rng.uniform(-spread, spread, (rest, 2)), rng.uniform(-2, 1, (rest, 1))], 1))
    mos.append((rng.rand(rest) < 0.01).astype(np.int64))
    return np.concatenate(pts).astype(np.float32), np.concatenate(mos)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--points', type=int, default=160000 * 5)
    ap.add_argument('--boxes', type=int, default=40)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--cpu', type=int, default=1)
    a = ap.parse_args()
    rng = np.random.RandomState(0)
    sc = [scene(rng, a.points, a.boxes) for _ in range(a.batch)]
    pts = np.concatenate([s[0] for s in sc])
    mos = np.concatenate([s[1] for s in sc])
    batch = np.concatenate([np.full(len(s[0]), i, np.int32) for i, s in enumerate(sc)])
    off = (rng.randn(len(pts), 2) * 0.03).astype(np.float32)
    dev = torch.device('cuda:0')
    t_pts, t_off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
    t_sel, t_b = torch.from_numpy((mos == 1).astype(np.uint8)).to(dev), torch.from_numpy(batch).to(dev)
    run = lambda: native.cluster(t_pts, t_off, t_sel, t_b, a.batch, 0.05, 0.4, 5, 15)
    lab = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    gpu_ms = e0.elapsed_time(e1) / a.iters
    out = {'points': int(len(pts)), 'moving': int(mos.sum()), 'instances': int(lab.max()), 'gpu_ms': round(gpu_ms, 3)}
    if a.cpu:
        import oracle
        from sklearn.cluster import DBSCAN
        ti = np.stack([batch.astype(np.int64), np.zeros(len(pts), np.int64)], 1)
        t = time.time()
        want = oracle.cluster_forward(pts, mos, off, ti, 0.4, 5, 15, True, estimator=DBSCAN(min_samples=5, eps=0.4))
        out['host_ms'] = round((time.time() - t) * 1e3, 1)
        out['identical'] = bool(np.array_equal(want, lab.cpu().numpy()))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
