#!/bin/bash
# round 5 call 13: probe; compact_mask test + model parity tests; step timing (3 runs) with per-step lists
bash tools/gpu_r05_probe.sh
python -m pytest tests/test_hip_ops.py tests/test_model_parity.py tests/test_config_parity.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do
  PCACC_BENCH_DEBUG=1 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_cm.json 2> gpurun_out/r05_bench_cm.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_bench_cm.json").read().strip().splitlines()[-1])
print("mixed", round(d["ms_per_step"], 2), "p10/p50/p90/max", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), round(d["ms_per_step_max"], 2), "scatter frac", round(d["roofline"]["frac"], 3))
PY
done
