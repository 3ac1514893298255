#!/bin/bash
mkdir -p gpurun_out
echo "=== bench with cpu baseline"
timeout 1500 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/bench_full.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o pmc -- python3 $GRAFT_REPO_ROOT/tools/pmc_scatter.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -type f | head; nproc; cat /proc/cpuinfo | grep "model name" | head -1
