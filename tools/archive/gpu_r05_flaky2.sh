#!/bin/bash
# the sequence in which tests/test_config_parity.py::test_gpu_config_fp32[fp32-c3_lidar] failed once (call 13), N times; failures land in gpurun_out/test_failures.txt
rm -f gpurun_out/test_failures.txt
for i in $(seq 1 ${1:-4}); do
  python -m pytest tests/test_hip_ops.py tests/test_model_parity.py tests/test_config_parity.py -q -m gpu 2>&1 | tail -1
done
[ -f gpurun_out/test_failures.txt ] && cut -c1-600 gpurun_out/test_failures.txt | head -80
