#!/bin/bash
# round 4, verdict item 1(a), second pass: the exact command behind profiles/r03_bench_lidar.json (bf16 leg, then the fp32x3 leg in the same process), twice,
# and the per-stage GPU times of the bf16 and fp32x3 steps (forward stages / loss / backward)
mkdir -p gpurun_out
for i in 1 2; do
  timeout 900 python bench.py --points lidar --no-cpu-baseline --no-configs > gpurun_out/r04_bench_lidar_$i.json 2> gpurun_out/err_lidar_$i.txt
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bench_lidar_$i.json').read().strip().splitlines()[-1])
print('lidar run $i: bf16 %.2f ms, matched %.2f ms' % (d['ms_per_step'], d['matched_accuracy']['ms_per_step']))"
done
for d in bf16 fp32x3; do
  echo "== stage times $d"; PCACC_DTYPE=$d timeout 600 python tools/stage_gpu_times.py 2>&1 | tail -16
done
