#!/bin/bash
# per-sequence step time against the batch size (is the four-sequence step GPU-bound?)
for b in 1 2 4 6 8; do
  ms=$(timeout 900 python bench.py --dtype mixed --batch $b --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_b.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>/dev/null)
  echo "mixed B=$b $ms"
done
