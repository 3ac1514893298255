#!/bin/bash
# round 5 call 6: probe; fp32 inputs of the forward convolutions read with the streaming policy (A/B); bf16-mode step with the cached / streaming bf16 canvas fill
bash tools/gpu_r05_probe.sh
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), "scatter us", round(d["roofline"]["avg_launch_us"], 1), "frac", round(d["roofline"]["frac"], 3))
PY
}
for i in 1 2 3; do for v in base in_nt; do
  if [ $v = in_nt ]; then export PCACC_LIB=$PWD/build/libpcacc_hip_in_nt.so; else unset PCACC_LIB; fi
  python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_in_$v.json 2> gpurun_out/r05_bench_in_$v.err
  show gpurun_out/r05_bench_in_$v.json "mixed, conv inputs $v"
done; done
unset PCACC_LIB
for i in 1 2; do for v in 0 5; do
  PCACC_SCATTER_VARIANT=$v python bench.py --dtype bf16 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_bf16_sv$v.json 2> gpurun_out/r05_bench_bf16_sv$v.err
  show gpurun_out/r05_bench_bf16_sv$v.json "bf16 mode, scatter variant $v"
done; done
