#!/bin/bash
# round 6: the fused pooling + canvas kernel as a software pipeline (PCACC_SCATTER_VARIANT=p / q) against the chain-per-cell form: alone, then in the step
mkdir -p gpurun_out
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), "roofline frac", round(d["roofline"]["frac"], 3), "avg us", round(d["roofline"].get("avg_launch_us") or 0, 1))
PY
}
{
timeout 600 python tools/bench_fused_canvas.py 2>&1 | grep variant
for v in p ""; do PCACC_SCATTER_VARIANT=$v timeout 300 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "canvas or pooling" 2>&1 | tail -1; done
for i in 1 2 3; do for v in p q none; do
  if [ $v = none ]; then unset PCACC_SCATTER_VARIANT; else export PCACC_SCATTER_VARIANT=$v; fi
  timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r06_pipe_$v.json 2> gpurun_out/r06_pipe_$v.err
  show gpurun_out/r06_pipe_$v.json "mixed, scatter variant $v"
done; done
unset PCACC_SCATTER_VARIANT
} 2>&1 | tee gpurun_out/r06_fused_canvas_pipeline_ab.txt
bash tools/gpu_r06_prio.sh
