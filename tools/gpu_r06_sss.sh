#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_determinism.py -q -m gpu -x -k "scatter_sum or small or determin or reproducible or tube or csr" 2>&1 | tail -2
echo "== this tree"; timeout 300 python tools/bench_scatter_sum_small.py 2>&1 | grep '"n"'
echo "== before (256 threads, one float per lane, 256 tables)"; PCACC_LIB=$PWD/build/lines0/libpcacc_hip.so timeout 300 python tools/bench_scatter_sum_small.py 2>&1 | grep '"n"'
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2))
PY
}
for i in 1 2 3; do for v in new before; do
  if [ $v = before ]; then export PCACC_LIB=$PWD/build/lines0/libpcacc_hip.so; else unset PCACC_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r06_sss_ab_$v.json 2> gpurun_out/r06_sss_ab_$v.err
  show gpurun_out/r06_sss_ab_$v.json "mixed, library $v"
done; done
} 2>&1 | tee gpurun_out/r06_scatter_sum_small_ab.txt
