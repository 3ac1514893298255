#!/usr/bin/env python
"""Per-queue timeline of the last training steps in a rocprofv3 kernel trace: busy time per queue, time with 1 / 2+ queues busy,
idle time, and where in the step the queues overlap.  Usage: trace_timeline.py gpurun_out/prof_bench/bench_kernel_trace.csv [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows), key=lambda e: e[0])
# step boundaries: the fused Adam kernel closes a step
ends = [e[1] for e in ev if 'FusedAdam' in e[3] or 'multi_tensor_apply_kernel' in e[3] and 'Adam' in e[3]]
marks = sorted(set(ends))
# group Adam launches that sit within 1 ms of each other
bounds = []
for t in marks:
    if not bounds or t - bounds[-1] > 5e6:
        bounds.append(t)
    else:
        bounds[-1] = t
bounds = bounds[-(steps + 1):]
print('steps found', len(bounds) - 1)
for a, b in zip(bounds[:-1], bounds[1:]):
    sel = [e for e in ev if e[0] >= a and e[1] <= b]
    per_q = collections.defaultdict(float)
    pts = []
    for s, e, q, n in sel:
        per_q[q] += (e - s) / 1e6
        pts.append((s, 1)); pts.append((e, -1))
    pts.sort()
    depth, last, cover = 0, a, collections.defaultdict(float)
    for t, d in pts:
        cover[min(depth, 2)] += (t - last) / 1e6
        depth += d
        last = t
    cover[0] += (b - last) / 1e6
    print('step %.2f ms  launches %d  busy per queue %s  idle %.2f  one %.2f  two+ %.2f' % (
        (b - a) / 1e6, len(sel), {q: round(v, 2) for q, v in sorted(per_q.items())}, cover[0], cover[1], cover[2]))
