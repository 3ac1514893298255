#!/bin/bash
# round 5 call 5: fresh-box probe (now with the bring-up turnstile); bf16 shadows stored with the streaming policy vs plain stores (A/B, interleaved); launcher tests
bash tools/gpu_r05_probe.sh
for i in 1 2 3; do for v in nt plain; do
  if [ $v = plain ]; then export PCACC_LIB=$PWD/build/libpcacc_hip_shadow_plain.so; else unset PCACC_LIB; fi
  python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_sh_$v.json 2> gpurun_out/r05_bench_sh_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_bench_sh_$v.json").read().strip().splitlines()[-1])
print("shadow stores $v", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), "scatter", round(d["roofline"]["frac"], 3))
PY
done; done
unset PCACC_LIB
python -m pytest tests/test_bench_multirank.py tests/test_step.py -q -m gpu 2>&1 | tail -3
