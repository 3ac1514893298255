#!/bin/bash
# the library (fp32) mode's trajectory leg in fresh processes: with and without cudnn.deterministic -- how many distinct outcomes, do they pass
mkdir -p gpurun_out
O=gpurun_out/r06_fp32_trajectory_processes.txt
: > $O
for det in 1 0; do for i in 1 2 3 4 5 6 7 8; do
  echo "== deterministic=$det process $i" >> $O
  PCACC_TRAJ_LIBRARY_DET=$det PCACC_TRAJ_VERBOSE=1 timeout 300 python -m pytest tests/test_train_trajectory.py -q -m gpu -s -k "trajectory and fp32- and tiny" 2>&1 | grep -E "^loss rel|passed|failed" | cut -c1-200 >> $O
done; done
grep -c passed $O; grep "failed" $O | head; grep "^loss rel" $O | sort | uniq -c | sort -rn | head -20
