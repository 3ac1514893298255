#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_prepare_batch.py tests/test_prep.py tests/test_config_parity.py -q -m gpu -x -k "vox or collate or prepare or digest or prep or config" 2>&1 | tail -2
bash tools/gpu_pmc_kernels.sh > gpurun_out/pmc_kernels.log 2>&1; tail -2 gpurun_out/pmc_kernels.log | cut -c1-120
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/pmc_kernels_summary.json'))
for k, v in d.items():
    if isinstance(v, dict) and k.startswith('vox_'):
        print('%-24s %7.1f us  hbm/alg %s' % (k, v.get('avg_us_under_pmc', 0), v.get('hbm_over_algorithmic')))
PY
} 2>&1 | tee gpurun_out/r06_vox_xcd.txt
cp gpurun_out/pmc_kernels_summary.json gpurun_out/r06_pmc_kernels_summary.json
