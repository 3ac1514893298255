#!/bin/bash
# round 5 call 4: fresh-box probe; register-operand fp32x3 convolution: equality tests, isolated A/B, step A/B
bash tools/gpu_r05_probe.sh
python -m pytest tests/test_conv_split.py -q -m gpu -x 2>&1 | tail -5
python tools/bench_conv_reg.py 2>&1 | grep layer | tee gpurun_out/r05_conv_reg_ab.txt
for v in 1 0 1 0; do
  PCACC_CONV_REG=$v python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_cr$v.json 2> gpurun_out/r05_bench_cr$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_bench_cr$v.json").read().strip().splitlines()[-1])
print("PCACC_CONV_REG=$v", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), "scatter", round(d["roofline"]["frac"], 3))
PY
done
