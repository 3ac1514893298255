#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv.py tests/test_hip_ops.py -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
echo "=== bench_conv"; timeout 600 python tools/bench_conv.py --lib 0 2>&1 | grep '^{' | tee gpurun_out/bench_conv_nolib.jsonl | grep wgrad | cut -c1-200
echo "=== bench"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_nocpu.json | cut -c1-330
bash tools/gpu_pmc_kernels.sh > gpurun_out/pmc_kernels.log 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_kernels_summary.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'TFLOPs' in v: print(k, v['avg_us_under_pmc'], v['TFLOPs'], 'mfma', v.get('mfma_busy_frac'), 'ldsconf', v.get('lds_conflict_share'), 'hbm/alg', v.get('hbm_over_algorithmic'))
PY
