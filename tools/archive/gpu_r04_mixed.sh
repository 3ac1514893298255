#!/bin/bash
# round 4: the 'mixed' compute mode (fp32x3 forward, bf16 backward inside the convolution segments): its tests, the config fixtures, and the step time beside fp32x3
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mixed.py -q -m gpu > gpurun_out/t_mixed.txt 2>&1; tail -5 gpurun_out/t_mixed.txt
timeout 1500 python -m pytest tests/test_config_parity.py -q -m gpu -k "mixed" > gpurun_out/t_cfg_mixed.txt 2>&1; tail -8 gpurun_out/t_cfg_mixed.txt
PCACC_TRAJ_VERBOSE=1 timeout 1500 python -m pytest tests/test_train_trajectory.py -q -s -m gpu -k "tiny" > gpurun_out/t_traj.txt 2>&1; tail -8 gpurun_out/t_traj.txt
if [ "$1" != "nobench" ]; then
for i in 1 2; do
for d in mixed fp32x3; do
  ms=$(timeout 900 python bench.py --dtype $d --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_$d.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$d $ms"
done
done
fi
