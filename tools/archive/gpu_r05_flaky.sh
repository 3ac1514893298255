#!/bin/bash
# how often does tests/test_config_parity.py::test_gpu_config_fp32[fp32-c3_lidar] fail, and on which assertion?  N runs of the test in fresh processes
mkdir -p gpurun_out/r05_flaky
N=${1:-8}
for i in $(seq 1 $N); do
  python -m pytest "tests/test_config_parity.py::test_gpu_config_fp32[fp32-c3_lidar]" "tests/test_config_parity.py::test_gpu_config_fp32[mixed-c3_lidar]" -q -m gpu > gpurun_out/r05_flaky/run$i.txt 2>&1
  echo "run $i: $(tail -1 gpurun_out/r05_flaky/run$i.txt)"
  grep -E "^E  " gpurun_out/r05_flaky/run$i.txt | head -12 | cut -c1-400
done
