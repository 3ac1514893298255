#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_mixed.py tests/test_step.py tests/test_bench_multirank.py tests/test_train_trajectory.py "tests/test_config_parity.py::test_gpu_config_fp32" "tests/test_config_parity.py::test_gpu_config_fused_matching" tests/test_config_parity.py::test_device_key_point_sampler_gives_the_host_sampler_error_distribution -q -m gpu > gpurun_out/t_part.txt 2>&1; tail -14 gpurun_out/t_part.txt
PCACC_MODES=fp32x3,mixed timeout 900 python tools/gradnorm_dev.py c3 c5 c3_lidar 2>&1 | grep -v Warn | tail -8 | cut -c1-330
bash tools/gpu_r04_tail.sh
