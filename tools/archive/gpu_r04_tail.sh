#!/bin/bash
# round 4: where the mixed step's torch-side time and its standalone absmax passes come from; steady kernel stats; call table
mkdir -p gpurun_out
PCACC_DTYPE=mixed timeout 600 python tools/profile_torch_tail.py 60 > gpurun_out/r04_torch_tail_mixed.txt 2>&1; head -75 gpurun_out/r04_torch_tail_mixed.txt | cut -c1-230
PCACC_DTYPE=mixed timeout 600 python tools/trace_absmax.py > gpurun_out/r04_absmax_mixed.txt 2>&1; tail -30 gpurun_out/r04_absmax_mixed.txt | cut -c1-200
for i in 1 2; do for d in mixed fp32x3; do
  ms=$(timeout 900 python bench.py --dtype $d --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_$d.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$d $ms"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mixed -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype mixed --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_mixed.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats_steady.py gpurun_out/prof_mixed/bench_kernel_trace.csv 4 200 > gpurun_out/r04_mixed_steady_v2.txt; head -12 gpurun_out/r04_mixed_steady_v2.txt | cut -c1-200
PCACC_DTYPE=mixed timeout 600 python tools/native_call_table.py 20 > gpurun_out/r04_native_call_table_mixed.txt 2>&1; head -5 gpurun_out/r04_native_call_table_mixed.txt | cut -c1-200
