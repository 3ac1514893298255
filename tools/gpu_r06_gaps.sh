#!/bin/bash
# steady profiles with the idle-gap analysis: the default step (4 sequences) and one sequence per step
bash tools/gpu_r06_steady.sh r06_gaps_b4 > /dev/null 2>&1
sed -n '/^GPU active/,$p' gpurun_out/r06_gaps_b4.txt | cut -c1-220
PCACC_BENCH_EXTRA="--batch 1" bash tools/gpu_r06_steady.sh r06_gaps_b1 > /dev/null 2>&1
head -8 gpurun_out/r06_gaps_b1.txt; sed -n '/^GPU active/,$p' gpurun_out/r06_gaps_b1.txt | cut -c1-220
