#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mixed.py -q -m gpu > gpurun_out/t_mixed.txt 2>&1; tail -4 gpurun_out/t_mixed.txt
timeout 1500 python -m pytest tests/test_config_parity.py -q -m gpu -k "mixed" > gpurun_out/t_cfg_mixed.txt 2>&1; tail -6 gpurun_out/t_cfg_mixed.txt
PCACC_MODES=fp32x3,mixed timeout 900 python tools/gradnorm_dev.py c3 c5 c3_lidar 2>&1 | grep -v Warn | tail -8
PCACC_TRAJ_VERBOSE=1 timeout 1500 python -m pytest tests/test_train_trajectory.py -q -s -m gpu > gpurun_out/t_traj.txt 2>&1; tail -12 gpurun_out/t_traj.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mixed -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype mixed --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_mixed.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats_steady.py gpurun_out/prof_mixed/bench_kernel_trace.csv 4 200 > gpurun_out/r04_mixed_steady_v1.txt; head -70 gpurun_out/r04_mixed_steady_v1.txt | cut -c1-200
