#!/bin/bash
# round 5 experiment: how much of north_star's 1e-3 does the w_hi * x_lo term of the fp32x3 forward buy?  build/libpcacc_hip_x2.so = the library with the
# ACTIVATION lo planes zeroed at staging (convolutions, row-linear layers, pillar blocks) -- i.e. products = (w_hi + w_lo) * fp16(x): accuracy of a two-MFMA
# forward without its speed.  Same parity tests, deltas dumped, against the default library.
for lib in "" build/libpcacc_hip_x2.so; do
  rm -f gpurun_out/bf16_deltas.jsonl
  echo "=== lib=${lib:-default}"
  PCACC_LIB=$lib PCACC_DUMP_DELTAS=1 timeout 1200 python -m pytest tests/test_config_parity.py -q -m gpu -k "test_gpu_config_fp32 and mixed" 2>&1 | tail -4
  python3 - <<'PY'
import json
for l in open('gpurun_out/bf16_deltas.jsonl'):
    d = json.loads(l)
    dl = {k: round(d['got'][k] - d['ref'][k], 6) for k in d['got']}
    print(d['config'], d['dtype'], dl, 'flips', round(d['fb_flips'], 5), ('loss %.6f ref %.6f' % (d['loss'], d['loss_ref'])) if 'loss' in d else '')
PY
done
