#!/bin/bash
# interleaved A/B of two bench command lines: tools/gpu_ab.sh N "flags A" "flags B"
N=$1
for i in $(seq 1 $N); do
for arm in A B; do
  if [ $arm = A ]; then F="$2"; else F="$3"; fi
  ms=$(timeout 900 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg $F 2>&1 | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$arm $ms"
done
done
