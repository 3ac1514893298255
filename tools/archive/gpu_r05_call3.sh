#!/bin/bash
# round 5 call 3: fresh-box probe; wide-map weight gradient tests; scatter variants incl. streaming-store-only; library convs again; in-step scatter variants
bash tools/gpu_r05_probe.sh
python -m pytest tests/test_conv.py -q -m gpu -k "wgrad_deep" 2>&1 | tail -3
python tools/bench_scatter_variants.py 2>&1 | grep variant | tee gpurun_out/r05_scatter_variants2.txt
python tools/find_library_convs.py mixed 4 2>&1 | tail -8 | tee gpurun_out/r05_library_convs_mixed.txt
for v in 0 6 5 0 6 5; do
  PCACC_SCATTER_VARIANT=$v python bench.py --no-cpu-baseline --no-configs --no-fp32-leg > gpurun_out/r05_bench_sv$v.json 2> gpurun_out/r05_bench_sv$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_bench_sv$v.json").read().strip().splitlines()[-1])
print("variant $v", round(d["ms_per_step"], 2), round(d["ms_per_step_p50"], 2), "scatter us", round(d["roofline"]["avg_launch_us"], 2), "frac", round(d["roofline"]["frac"], 3), "cold", d["roofline"]["cold_cache"]["frac"], d["roofline"]["cold_cache"]["median_launch_us"])
PY
done
