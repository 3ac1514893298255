#!/bin/bash
# round 5 call 7: probe; XCD-contiguous block walk: equality test, conv tests, step A/B, isolated kernels A/B
bash tools/gpu_r05_probe.sh
python -m pytest tests/test_conv_split.py tests/test_conv.py tests/test_upconv_bf16.py -q -m gpu -x 2>&1 | tail -3
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), "scatter frac", round(d["roofline"]["frac"], 3))
PY
}
for i in 1 2 3; do for v in 1 0; do
  PCACC_XCD_REMAP=$v python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_xcd$v.json 2> gpurun_out/r05_bench_xcd$v.err
  show gpurun_out/r05_bench_xcd$v.json "mixed, PCACC_XCD_REMAP=$v"
done; done
for v in 1 0; do echo "== bench_conv_split PCACC_XCD_REMAP=$v"; PCACC_XCD_REMAP=$v python tools/bench_conv_split.py --bf16 1 2>&1 | grep -E "72|36|18|144|sum_us" | cut -c1-260; done
