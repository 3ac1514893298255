#!/bin/bash
mkdir -p gpurun_out
echo "== whole-model + tube + mixed tests"
timeout 1500 python -m pytest tests/test_mixed.py tests/test_twin.py tests/test_tube.py tests/test_model_parity.py tests/test_config_parity.py tests/test_train_trajectory.py tests/test_step.py tests/test_determinism.py -q -m gpu 2>&1 | tail -6
echo "== bench x3"
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
bash tools/gpu_r06_steady.sh r06_mixed_steady_d
