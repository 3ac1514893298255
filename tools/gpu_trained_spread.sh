#!/bin/bash
# distribution of test_gpu_bf16_against_fp32_on_trained_weights' measured deltas over repeated runs (the 150 training steps are not
# bit-reproducible, so every run evaluates a slightly different model)
mkdir -p gpurun_out; rm -f gpurun_out/bf16_deltas.jsonl
for i in $(seq 1 $1); do
  PCACC_DUMP_DELTAS=1 timeout 600 python -m pytest tests/test_config_parity.py -q -p no:cacheprovider -k trained_weights 2>&1 | grep -E "passed|failed" | cut -c1-80
done
python3 - <<'PY'
import json
rows=[json.loads(l)['got'] for l in open('gpurun_out/bf16_deltas.jsonl') if '"trained_tiny"' in l]
keys=['rot_set','trans_set','mos_iou_set','epe_set','rot_median','trans_median','epe_median','flips_median','rot','trans','epe','fb_flips']
print('runs', len(rows))
for k in keys:
    v=[r[k] for r in rows]
    print('%-14s max %.5f  all %s' % (k, max(v), [round(x,5) for x in v]))
PY
