#!/bin/bash
# the default command exactly as the driver runs it, with its wall time
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python bench.py > gpurun_out/bench_default_full.json 2> gpurun_out/bench_default_full.err
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -3 gpurun_out/bench_default_full.err | cut -c1-300
tail -1 gpurun_out/bench_default_full.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','n_gpus','steps','warmup','dtype','vs_baseline')})
print('roofline', {k:d['roofline'].get(k) for k in ('bound','achieved','peak','frac','traffic')})
print('cpu_baseline', d.get('cpu_baseline'))
print('matched_accuracy', d.get('matched_accuracy'))
"
