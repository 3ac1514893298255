#!/bin/bash
mkdir -p gpurun_out
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90/max", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), round(d["ms_per_step_max"], 2))
PY
}
{
for i in 1 2 3 4; do for v in freeze off none; do
  if [ $v = none ]; then unset PCACC_GC; else export PCACC_GC=$v; fi
  timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 40 > gpurun_out/r06_gc.json 2> gpurun_out/r06_gc.err
  show gpurun_out/r06_gc.json "[PCACC_GC=$v, 40 steps]"
done; done
unset PCACC_GC
} 2>&1 | tee gpurun_out/r06_gc_ab.txt
