#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_step.py tests/test_mixed.py -q -m gpu -k "collate or batcher or prepared or mixed or Mixed or rounded" > gpurun_out/t_part.txt 2>&1; tail -8 gpurun_out/t_part.txt
for i in 1 2 3; do for e in "PCACC_BATCHED_COLLATE=0" "PCACC_BATCHED_COLLATE=1"; do
  ms=$(env $e timeout 900 python bench.py --dtype mixed --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_mixed.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$e $ms"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mixed -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype mixed --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_mixed.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats_steady.py gpurun_out/prof_mixed/bench_kernel_trace.csv 4 200 > gpurun_out/r04_mixed_steady_v3.txt; head -12 gpurun_out/r04_mixed_steady_v3.txt | cut -c1-200; grep -n "vox_\|Cat\|bfloat16_copy" gpurun_out/r04_mixed_steady_v3.txt | cut -c1-200
