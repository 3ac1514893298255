#!/bin/bash
# default bench line (find-db, prefetch), rocprofv3 kernel stats of the same command, a heuristic-mode summary for the per-category
# split (the find benchmark kernels of the warm-up pollute the default command's totals), batch sweep, eval forward
mkdir -p gpurun_out
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-330 gpurun_out/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_default -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-configs > $GRAFT_REPO_ROOT/gpurun_out/rocprof_default.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-configs --no-miopen-find > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats.py gpurun_out/prof_bench/bench_kernel_stats.csv 9 60 > gpurun_out/bench_summary.txt; head -10 gpurun_out/bench_summary.txt | cut -c1-170
grep pillar_scatter_vec4 gpurun_out/prof_default/bench_kernel_stats.csv | cut -c1-200
for b in 1 2 6 8; do timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-configs --batch $b 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', $b, d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
timeout 600 python tools/bench_eval.py 2>&1 | grep '^{' > gpurun_out/bench_eval.log; cut -c150-260 gpurun_out/bench_eval.log
tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null
