#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/stage_backward_times.py 2>&1 | grep -v Warning | tail -20 | tee gpurun_out/stage_backward_times.txt
for s in tubenet ego loss; do echo "=== $s"; timeout 600 python tools/profile_stage_ops.py $s 2>&1 | grep -v Warning | tail -48 | tee gpurun_out/stage_ops_$s.txt | head -40 | cut -c1-150; done
