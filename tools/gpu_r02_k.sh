#!/bin/bash
mkdir -p gpurun_out
for s in tubenet ego forward loss step; do echo "=== $s"; timeout 600 python tools/count_dispatched_ops.py $s 2>&1 | grep -v Warn | tail -92 > gpurun_out/dispatched_$s.txt; head -3 gpurun_out/dispatched_$s.txt; done
