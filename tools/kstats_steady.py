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
qcol = next((c for c in ('Stream_Id', 'Queue_Id') if rows and c in rows[0] and len({r[c] for r in rows}) > 1), None)    # [r6] which HIP stream (else: which hardware queue) a kernel ran on
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r[qcol] if qcol else '0') for r in rows), key=lambda e: e[0])
queue_of = {(e[0], e[1], e[2]): e[3] for e in ev}
ev = [e[:3] for e in ev]
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

# [r6] where the GPU waits for the host: the union of all kernel intervals (any stream) against the wall time, and the longest idle gaps with the kernels around them
iv = sorted((s, e, n) for s, e, n in sel)
active, idle_gaps, cur_s, cur_e, last_name = 0, [], None, None, None
for s, e, n in iv:
    if cur_e is None:
        cur_s, cur_e, last_name = s, e, n
        continue
    if s > cur_e:
        active += cur_e - cur_s
        idle_gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e, last_name = s, e, n
    elif e > cur_e:
        cur_e, last_name = e, n
active += cur_e - cur_s
idle = sum(g[0] for g in idle_gaps)
print('\nGPU active (union over streams) %.2f ms/step, idle between kernels %.2f ms/step in %.0f gaps/step; gaps > 20 us: %.2f ms/step in %.0f gaps/step; > 100 us: %.2f ms/step in %.1f gaps/step' % (
    active / 1e6 / n_steps, idle / 1e6 / n_steps, len(idle_gaps) / n_steps, sum(g[0] for g in idle_gaps if g[0] > 20e3) / 1e6 / n_steps,
    sum(1 for g in idle_gaps if g[0] > 20e3) / n_steps, sum(g[0] for g in idle_gaps if g[0] > 100e3) / 1e6 / n_steps, sum(1 for g in idle_gaps if g[0] > 100e3) / n_steps))
by_pair = collections.defaultdict(lambda: [0.0, 0])
for g, a, b in idle_gaps:
    if g > 20e3:
        k = (a[:60], b[:60])
        by_pair[k][0] += g / 1e3 / n_steps
        by_pair[k][1] += 1
print('idle gaps > 20 us by (kernel before, kernel after), us/step:')
for (a, b), v in sorted(by_pair.items(), key=lambda x: -x[1][0])[:25]:
    print('  %7.1f us/step n=%3d  %s  ->  %s' % (v[0], v[1], a, b))

# [r6] per stream / queue: kernel time, active time (union) and the part of the step it spans -- which stream the step's length hangs on
per_q = collections.defaultdict(list)
for s_, e_, n_ in sel:
    per_q[queue_of.get((s_, e_, n_), '0')].append((s_, e_, n_))
print('\nper %s (steady steps): launches/step, kernel time, active (union), idle inside its own span' % (qcol or 'queue'))
for q, lst in sorted(per_q.items(), key=lambda x: -sum(e - s for s, e, _ in x[1])):
    lst.sort()
    busy = sum(e - s for s, e, _ in lst)
    act, ce = 0, None
    cs = None
    for s_, e_, _ in lst:
        if ce is None or s_ > ce:
            if ce is not None:
                act += ce - cs
            cs, ce = s_, e_
        elif e_ > ce:
            ce = e_
    act += ce - cs
    print('  %-8s %7.0f launches/step  kernel time %6.2f ms/step  active %6.2f ms/step  (wall %.2f)' % (q, len(lst) / n_steps, busy / 1e6 / n_steps, act / 1e6 / n_steps, (bounds[-1] - bounds[0]) / 1e6 / n_steps))
# the main stream's critical stretch: time during which ONLY one stream has a kernel running, per stream
edges = []
for q, lst in per_q.items():
    for s_, e_, _ in lst:
        edges.append((s_, 1, q))
        edges.append((e_, -1, q))
edges.sort()
alone = collections.defaultdict(int)
live = collections.Counter()
prev = None
for t, d, q in edges:
    if prev is not None:
        on = [k for k, v in live.items() if v > 0]
        if len(on) == 1:
            alone[on[0]] += t - prev
    live[q] += d
    prev = t
print('time with kernels of ONE stream only: ' + ', '.join('%s %.2f ms/step' % (q, v / 1e6 / n_steps) for q, v in sorted(alone.items(), key=lambda x: -x[1])))
# per stream: where inside the step (0 = the previous step's Adam finished) its first kernel starts and its last kernel ends, averaged over the steady steps
span = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i in range(n_steps):
    lo_, hi_ = bounds[i], bounds[i + 1]
    for q, lst in per_q.items():
        mine = [(s_, e_) for s_, e_, _ in lst if s_ >= lo_ and e_ <= hi_]
        if mine:
            span[q][0] += (min(m[0] for m in mine) - lo_) / 1e6
            span[q][1] += (max(m[1] for m in mine) - lo_) / 1e6
            span[q][2] += 1
print('first kernel start / last kernel end inside the step (ms after the previous step\'s Adam): ' +
      ', '.join('%s %.2f .. %.2f' % (q, v[0] / v[2], v[1] / v[2]) for q, v in sorted(span.items(), key=lambda x: x[1][0] / max(x[1][2], 1))))
# the kernels of every stream but the busiest (the side chain: what the step's length hangs on behind the lower half's forward)
ranked = sorted(per_q.items(), key=lambda x: -sum(e - s for s, e, _ in x[1]))
for q, lst in ranked[1:2]:
    agg_q = collections.defaultdict(lambda: [0.0, 0])
    for s_, e_, n_ in lst:
        agg_q[n_][0] += (e_ - s_) / 1e6 / n_steps
        agg_q[n_][1] += 1.0 / n_steps
    print('\nkernels of %s %s, by time:' % (qcol or 'queue', q))
    for n_, v in sorted(agg_q.items(), key=lambda x: -x[1][0])[:70]:
        print('  %-96s n/step=%6.1f avg=%8.1fus tot/step=%6.2fms' % (n_[:96], v[1], v[0] / v[1] * 1e3, v[0]))
# [r6] PCACC_KSTATS_DUMP=<file>: the last steady step's launches in start order -- stream, start (us after the step began), duration, gap to the previous kernel of the SAME stream, name
if os.environ.get('PCACC_KSTATS_DUMP'):
    lo_, hi_ = bounds[-2], bounds[-1]
    last_end = {}
    with open(os.environ['PCACC_KSTATS_DUMP'], 'w') as f:
        for s_, e_, n_ in sorted(sel):
            if s_ < lo_ or e_ > hi_:
                continue
            q = queue_of.get((s_, e_, n_), '0')
            gap = (s_ - last_end[q]) / 1e3 if q in last_end else 0.0
            last_end[q] = e_
            f.write('%s %9.1f %7.1f %7.1f  %s\n' % (q, (s_ - lo_) / 1e3, (e_ - s_) / 1e3, gap, n_[:150]))
