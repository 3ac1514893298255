#!/bin/bash
timeout 600 python -m pytest tests/test_conv.py -q -x -p no:cacheprovider 2>&1 | tail -1
for i in 1 2; do
timeout 600 python tools/bench_conv.py --lib 0 2>&1 | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if any(k in d['layer'] for k in ('32->32 @288','temporal')):
        print('  %-30s %7.1f us' % (d['layer'], d['mfma_us']))
"
done
