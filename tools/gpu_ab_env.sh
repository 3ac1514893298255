#!/bin/bash
# interleaved A/B of two environments on one bench command line: tools/gpu_ab_env.sh N "VAR=a" "VAR=b" "bench flags"
N=$1
for i in $(seq 1 $N); do
for arm in A B; do
  if [ $arm = A ]; then E="$2"; else E="$3"; fi
  ms=$(env $E timeout 900 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model $4 2>&1 | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$arm ($E) $ms"
done
done
