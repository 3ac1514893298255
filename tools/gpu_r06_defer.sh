#!/bin/bash
mkdir -p gpurun_out
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2))
PY
}
{
for i in 1 2 3 4 5; do for v in 1 0; do
  PCACC_EARLY_DEFER=$v timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r06_defer_$v.json 2> gpurun_out/r06_defer_$v.err
  show gpurun_out/r06_defer_$v.json "mixed, early backward deferred = $v"
done; done
PCACC_EARLY_DEFER=1 timeout 900 python -m pytest tests/test_determinism.py tests/test_train_trajectory.py tests/test_distributed_gpu.py -q -m gpu -x 2>&1 | tail -2
} 2>&1 | tee -a gpurun_out/r06_early_defer_ab.txt
