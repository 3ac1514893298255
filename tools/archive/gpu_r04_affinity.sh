#!/bin/bash
# does binding the process to the cores of one L3 domain steady the two-thread step?  (helper thread forced on / off, mixed, four sequences)
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|L3" | head -12
cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list
for i in 1 2 3 4; do for t in 1 0; do for pin in none 0-7 0-3; do
  if [ $pin = none ]; then pre=""; else pre="taskset -c $pin"; fi
  ms=$(PCACC_EARLY_THREAD=$t timeout 900 $pre python bench.py --dtype mixed --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_t.txt | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],2))" 2>/dev/null)
  echo "thread=$t pin=$pin $ms"
done; done; done
