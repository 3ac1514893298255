#!/bin/bash
# round 2, first GPU pass: all GPU tests (bf16 deltas dumped), the default bench line, rocprofv3 kernel stats of the same workload,
# the pillar-scatter launch in isolation and its two PMC passes
mkdir -p gpurun_out
rm -f gpurun_out/bf16_deltas.jsonl
R=$GRAFT_REPO_ROOT
PCACC_DUMP_DELTAS=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -25 gpurun_out/pytest_gpu.log | cut -c1-300
echo "=== bf16 deltas"; cat gpurun_out/bf16_deltas.jsonl | cut -c1-600
echo "=== bench default"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-1500 gpurun_out/bench_default.json
echo "=== scatter isolated"; timeout 300 python tools/bench_scatter.py 2>&1 | grep '^{' | tee gpurun_out/bench_scatter.jsonl | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-miopen-find > $R/gpurun_out/rocprof_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python3 $R/tools/pmc_scatter.py > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python3 tools/kstats.py gpurun_out/prof_bench/bench_kernel_stats.csv 9 60 > gpurun_out/bench_summary.txt; head -45 gpurun_out/bench_summary.txt | cut -c1-170
python3 tools/pmc_summary.py $(ls gpurun_out/pmc_FETCH_SIZE/*/*counter_collection.csv gpurun_out/pmc_FETCH_SIZE/*counter_collection.csv 2>/dev/null | head -1) $(ls gpurun_out/pmc_WRITE_SIZE/*/*counter_collection.csv gpurun_out/pmc_WRITE_SIZE/*counter_collection.csv 2>/dev/null | head -1) gpurun_out/pmc_scatter_summary.json | tail -30
