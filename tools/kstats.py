#!/usr/bin/env python
"""Summarise a rocprofv3 kernel_stats.csv: time and launches per category per step. Usage: kstats.py file.csv n_steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
def cat(name):
    if name.startswith('Cijk'): return 'hipblaslt_gemm'
    if 'igemm' in name or name.startswith('_ZN2ck') or 'ck::' in name or 'naive_conv' in name or 'SubTensorOp' in name or 'batched_transpose' in name or 'gridwise' in name.lower() or 'MIOpen' in name: return 'miopen_conv'
    if 'BatchNorm' in name or 'batch_norm' in name: return 'batchnorm'
    if any(k in name for k in ['seg_', 'csr_', 'vox_', 'pillar_', 'gather_rows', 'bilinear', 'bev_warp', 'rigid_', 'scan_chunk', 'chunk_', 'cell_index', 'fp_', 'rows_linear', 'rows_wgrad', 'chamfer', 'conv3x3', 'conv_', 'cluster_', 'sinkhorn', 'kabsch', 'affinity', 'sample_subsets', 'upload_words', 'prep_points', 'pfn_', 'scatter_sum_small', 'offset_']): return 'pcacc_hip'
    if 'rocclr' in name: return 'memcpy/memset'
    if 'at::native' in name or 'rocprim' in name or 'indexing' in name: return 'torch_misc'
    return 'other'
agg = {}
for r in rows:
    d = agg.setdefault(cat(r['Name']), [0.0, 0.0])
    d[0] += float(r['TotalDurationNs']) / 1e6 / steps
    d[1] += int(r['Calls']) / steps
tot = sum(v[0] for v in agg.values())
print('total GPU-busy %.2f ms/step, %.0f launches/step' % (tot, sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda x: -x[1][0]):
    print('  %-16s %7.2f ms/step %7.0f launches/step' % (k, v[0], v[1]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
for r in rows[:n]:
    print('%-95s n/step=%6.1f avg=%8.1fus tot/step=%6.2fms' % (r['Name'][:95], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / steps))
