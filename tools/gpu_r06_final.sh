#!/bin/bash
# round 6, end: bit-reproducibility over every whole-model fixture and mode; the launch sequence of a steady step; the default bench as the driver runs it
mkdir -p gpurun_out
O=gpurun_out/r06_determinism_all.txt
: > $O
for c in c2 c3 c4 c5 nus11 c3_lidar tiny_train; do for d in mixed bf16 fp32x3; do
  echo "== $c $d" >> $O
  timeout 300 python tools/r06_determinism.py --config $c --dtype $d --runs 2 2>&1 | tail -2 >> $O
done; done
echo "== staged step, 3 runs" >> $O
timeout 600 python tools/r06_determinism.py --config c3_lidar --dtype mixed --runs 3 --step 2>&1 | tail -2 >> $O
grep -c "BIT-IDENTICAL" $O; grep -c "NOT reproducible" $O; grep -B2 "NOT reproducible\|Error\|Traceback" $O | head -20
bash tools/gpu_r06_steady.sh r06_mixed_steady > /dev/null 2>&1
head -8 gpurun_out/r06_mixed_steady.txt | cut -c1-150; sed -n '/^GPU active/,/^idle gaps/p' gpurun_out/r06_mixed_steady.txt | cut -c1-250; sed -n '/^per /,/^first kernel/p' gpurun_out/r06_mixed_steady.txt | cut -c1-250
bash tools/gpu_default_bench.sh; cp gpurun_out/bench_default_full.json gpurun_out/r06_bench_default_full.json
