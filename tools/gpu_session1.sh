#!/bin/bash
# first GPU session: parity tests, microbench, kernel trace
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_micro -o micro -- python3 $GRAFT_REPO_ROOT/tools/microbench.py > $GRAFT_REPO_ROOT/gpurun_out/rocprof_micro.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_micro -name "*stats*" | head
tail -5 gpurun_out/smoke.log; tail -40 gpurun_out/pytest_gpu.log; cat gpurun_out/microbench.log | tail -30
