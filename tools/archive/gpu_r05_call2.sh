#!/bin/bash
# round 5 call 2: fresh-box hang probe, pillar-scatter variants, which calls still reach the convolution library, default bench without find mode
bash tools/gpu_r05_probe.sh
python tools/bench_scatter_variants.py 2>&1 | tee gpurun_out/r05_scatter_variants.txt
python tools/find_library_convs.py mixed 4 2>&1 | tail -30 | tee gpurun_out/r05_library_convs_mixed.txt
python tools/find_library_convs.py bf16 4 2>&1 | tail -30 | tee gpurun_out/r05_library_convs_bf16.txt
python bench.py --no-cpu-baseline > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_a.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["ms_per_step_p10"], d["ms_per_step_p50"], d["ms_per_step_p90"], d["config"]["early_backward_thread"], d["bf16"]["ms_per_step"])
print(d["roofline"]["frac"], d["roofline"]["cold_cache"])
print([(c["config"], c.get("ms_per_step")) for c in d["configs"]])
PY
