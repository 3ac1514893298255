#!/usr/bin/env python
"""kstats.py on the STEADY steps of a rocprofv3 kernel trace: the trace is cut at the fused-Adam launches (one group per step), the first
`skip` steps (warm-up: library solver searches, first-touch compiles) are dropped, the rest averaged per step.
Usage: kstats_steady.py bench_kernel_trace.csv [skip=3] [rows=60]"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
marks = sorted(e[1] for e in ev if 'multi_tensor_apply_kernel' in e[2] and 'Adam' in e[2])
bounds = []
for t in marks:
    if not bounds or t - bounds[-1] > 5e6:
        bounds.append(t)
    else:
        bounds[-1] = t
bounds = bounds[skip:]
n_steps = len(bounds) - 1
assert n_steps >= 1, 'not enough steps in the trace'
sel = [e for e in ev if e[0] >= bounds[0] and e[1] <= bounds[-1]]


def cat(name):
    if name.startswith('Cijk'): return 'hipblaslt_gemm'
    if 'igemm' in name or name.startswith('_ZN2ck') or 'ck::' in name or 'naive_conv' in name or 'SubTensorOp' in name or 'batched_transpose' in name or 'gridwise' in name.lower() or 'MIOpen' in name and 'BatchNorm' not in name: return 'miopen_conv'
    if 'BatchNorm' in name or 'batch_norm' in name: return 'batchnorm'
    if any(k in name for k in ['seg_', 'csr_', 'vox_', 'pillar_', 'gather_rows', 'bilinear', 'bev_warp', 'rigid_', 'scan_chunk', 'chunk_', 'cell_index', 'fp_', 'rows_linear', 'rows_wgrad', 'chamfer', 'conv3x3', 'conv_', 'cluster_', 'sinkhorn', 'kabsch', 'affinity', 'sample_subsets', 'upload_words', 'prep_points', 'pfn_', 'scatter_sum_small', 'offset_', 'absmax256', 'tube_', 'bn_', 'frames_max', 'svd3', 'inv4x4', 'maxpool', 'pool_skip', 'upconv', 'head_conv']): return 'pcacc_hip'
    if 'rocclr' in name: return 'memcpy/memset'
    if 'at::native' in name or 'rocprim' in name or 'indexing' in name: return 'torch_misc'
    return 'other'


agg, per = {}, collections.OrderedDict()
for s, e, n in sel:
    d = agg.setdefault(cat(n), [0.0, 0])
    d[0] += (e - s) / 1e6 / n_steps
    d[1] += 1.0 / n_steps
    p = per.setdefault(n, [0.0, 0])
    p[0] += (e - s) / 1e6 / n_steps
    p[1] += 1.0 / n_steps
tot = sum(v[0] for v in agg.values())
print('steady steps %d (of %d), wall %.2f ms/step, GPU-busy %.2f ms/step, %.0f launches/step' % (
    n_steps, n_steps + skip, (bounds[-1] - bounds[0]) / 1e6 / n_steps, tot, sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda x: -x[1][0]):
    print('  %-16s %7.2f ms/step %7.0f launches/step' % (k, v[0], v[1]))
for n, v in sorted(per.items(), key=lambda x: -x[1][0])[:n_rows]:
    print('%-100s n/step=%6.1f avg=%8.1fus tot/step=%6.2fms' % (n[:100], v[1], v[0] / v[1] * 1e3, v[0]))
