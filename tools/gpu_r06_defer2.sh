#!/bin/bash
mkdir -p gpurun_out
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2))
PY
}
{
for cfg in "--dtype bf16" "--batch 1" "--batch 2" "--points lidar"; do
for i in 1 2 3; do for v in 1 0; do
  PCACC_EARLY_DEFER=$v timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model $cfg > gpurun_out/r06_defer2_$v.json 2> gpurun_out/r06_defer2_$v.err
  show gpurun_out/r06_defer2_$v.json "[$cfg] early backward deferred = $v"
done; done; done
} 2>&1 | tee gpurun_out/r06_early_defer_other_configs.txt
