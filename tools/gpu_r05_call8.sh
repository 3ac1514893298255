#!/bin/bash
# round 5 call 8: probe; two-workgroups-per-CU form of the resident fp32x3 kernel: tests, isolated A/B, step A/B
bash tools/gpu_r05_probe.sh
python -m pytest tests/test_conv_split.py -q -m gpu -x 2>&1 | tail -3
python tools/bench_conv_res2.py 2>&1 | grep layer | tee gpurun_out/r05_conv_res2_ab.txt
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), "scatter frac", round(d["roofline"]["frac"], 3))
PY
}
for i in 1 2 3; do for v in 1 0; do
  PCACC_CONV_RES2=$v python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_res2_$v.json 2> gpurun_out/r05_bench_res2_$v.err
  show gpurun_out/r05_bench_res2_$v.json "mixed, PCACC_CONV_RES2=$v"
done; done
