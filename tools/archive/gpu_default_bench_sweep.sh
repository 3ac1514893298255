#!/bin/bash
mkdir -p gpurun_out
echo "=== bench default (with cpu baseline)"; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.json | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-configs > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/rocprof_bench.log | cut -c1-300
python3 tools/kstats.py gpurun_out/prof_bench/bench_kernel_stats.csv 9 60 > gpurun_out/bench_summary.txt; head -12 gpurun_out/bench_summary.txt | cut -c1-170
for b in 1 2 6 8; do timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-configs --batch $b 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', $b, d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null
