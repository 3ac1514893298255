#!/usr/bin/env python
"""Spread of the pose-driven metrics of a golden config over forward seeds (= key-point draws), fp32 and bf16: is a bf16 deviation
from the golden value inside what another draw does to the fp32 model?  Development aid.  Usage: seed_spread.py c3 [n_seeds]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import test_config_parity as t

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = dict(np.load(os.path.join(os.path.dirname(t.__file__), 'golden', 'model_%s.npz' % name), allow_pickle=True))
for dtype in ('fp32', 'bf16'):
    rows = []
    for off in range(n):
        model, inp, out, stats, T = t._run(g, dtype, seed_offset=off)
        got, ref = t._metrics(g, inp, out, stats, T)
        rows.append(got)
        del model, inp, out, stats
    for k in ('ego_rot_error', 'ego_trans_error', 'epe_mean', 'mos_iou'):
        v = np.array([r[k] for r in rows])
        print(dtype, k, 'ref %.3f' % float(g[k]), 'mean %.3f sd %.3f min %.3f max %.3f' % (v.mean(), v.std(), v.min(), v.max()), np.round(v, 2).tolist())
