#!/bin/bash
for v in "" 1; do
  echo "== PCACC_CONV_64_CTW2=$v"
  for i in 1 2; do
  if [ -n "$v" ]; then export PCACC_CONV_64_CTW2=1; else unset PCACC_CONV_64_CTW2; fi
  timeout 600 python tools/bench_conv.py --lib 0 2>&1 | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if any(k in d['layer'] for k in ('64->64 @144','64->64 @288','32->32 @288')):
        print('  %-30s %7.1f us' % (d['layer'], d['mfma_us']))
"
  done
done
PCACC_CONV_64_CTW2=1 timeout 600 python -m pytest tests/test_conv.py -q -x -p no:cacheprovider 2>&1 | tail -1
