#!/bin/bash
# round 4 counter passes (one counter set per run, --kernel-trace only beside --pmc): the pillar-scatter kernels re-taken, the kernel set incl. the round-4 kernels
mkdir -p gpurun_out/pmc_scatter
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_scatter/$c -o p -- python3 $R/tools/pmc_scatter.py > $R/gpurun_out/pmc_scatter/$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_scatter/$c.log
done
cd $R
F=$(find gpurun_out/pmc_scatter/FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/pmc_scatter/WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W gpurun_out/r04_pmc_scatter_summary.json | tail -30
bash tools/gpu_pmc_kernels.sh > gpurun_out/pmc_kernels.log 2>&1; tail -5 gpurun_out/pmc_kernels.log
cp gpurun_out/pmc_kernels_summary.json gpurun_out/r04_pmc_kernels_summary.json
