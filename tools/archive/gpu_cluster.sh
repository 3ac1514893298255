#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cluster.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/pytest_cluster.log; tail -15 gpurun_out/pytest_cluster.log
echo "=== bench_cluster"; timeout 600 python tools/bench_cluster.py 2>&1 | tail -3 | tee gpurun_out/bench_cluster.log
timeout 300 python tools/bench_cluster.py --batch 1 --cpu 0 2>&1 | tail -1 | tee -a gpurun_out/bench_cluster.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cluster -o cluster -- python3 $GRAFT_REPO_ROOT/tools/bench_cluster.py --cpu 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_cluster.log 2>&1
cd $GRAFT_REPO_ROOT; head -20 gpurun_out/prof_cluster/*kernel_stats.csv 2>/dev/null | cut -c1-200
