#!/bin/bash
mkdir -p gpurun_out
for s in tubenet ego forward loss step; do echo "=== $s"; timeout 600 python tools/count_dispatched_ops.py $s 2>&1 | grep -v Warn | tail -92 > gpurun_out/dispatched_$s.txt; head -3 gpurun_out/dispatched_$s.txt | tail -2; done
for s in tubenet ego; do echo "=== $s"; timeout 600 python tools/profile_stage_ops.py $s 2>&1 | grep -v Warning | tail -48 > gpurun_out/stage_ops_$s.txt; done
