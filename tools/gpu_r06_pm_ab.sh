#!/bin/bash
# round 6: the pillar encoder on pillar-major rows (default) against point-order rows (PCACC_PILLAR_MAJOR=0): tests, then interleaved step times, then kernel times
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_mixed.py tests/test_determinism.py tests/test_config_parity.py -q -m gpu -x 2>&1 | tail -3
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), "roofline", d["roofline"]["kernel"][:28], round(d["roofline"]["frac"], 3), round(d["roofline"].get("avg_launch_us") or 0, 1))
PY
}
for i in 1 2 3; do for v in 1 0; do
  PCACC_PILLAR_MAJOR=$v timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r06_pm_ab_$v.json 2> gpurun_out/r06_pm_ab_$v.err
  show gpurun_out/r06_pm_ab_$v.json "mixed, pillar-major rows = $v"
done; done
} 2>&1 | tee gpurun_out/r06_pillar_major_ab.txt
for v in 1 0; do
  PCACC_PILLAR_MAJOR=$v bash tools/gpu_r06_steady.sh r06_pm_steady_$v > /dev/null 2>&1
  echo "== steady profile, pillar-major rows = $v" | tee -a gpurun_out/r06_pillar_major_ab.txt
  grep -E "^steady|pfn_block|seg_max|pfn_features|rows_linear_fewk|gather_rows|csr_" gpurun_out/r06_pm_steady_$v.txt | cut -c1-200 | tee -a gpurun_out/r06_pillar_major_ab.txt
done
