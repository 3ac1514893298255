#!/bin/bash
mkdir -p gpurun_out
echo "== fused canvas kernel test + mixed-mode tests"
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_mixed.py tests/test_twin.py -q -m gpu -k "pooling_into or mixed or twin or scatter_sum_small" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_config_parity.py -q -m gpu -k "mixed and (c3 or c5 or c2) and fp32" 2>&1 | tail -4
echo "== trajectory mixed-c1_i1 verbose (this tree)"
PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1_i1" -s > gpurun_out/r06_traj_new.txt 2>&1
grep -E "loss rel|term rel|gradient samples|updates:|passed|failed" gpurun_out/r06_traj_new.txt | cut -c1-900
echo "== same, unfused canvas"
PCACC_FUSED_CANVAS=0 PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1_i1" -s 2>&1 | grep -E "loss rel|term rel|gradient samples|updates:|passed|failed" | cut -c1-900
echo "== same, unfused canvas + index_add"
PCACC_FUSED_CANVAS=0 PCACC_DETERMINISTIC=0 PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1_i1" -s 2>&1 | grep -E "loss rel|term rel|gradient samples|updates:|passed|failed" | cut -c1-900
echo "== chamfer variants"
timeout 300 build/exp_chamfer 2>&1 | tee gpurun_out/r06_chamfer_variants.txt
echo "== bench A (separate pooling + fills) / B (pooling writes the canvas)"
for i in 1 2 3; do
  a=$(PCACC_FUSED_CANVAS=0 timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'])")
  b=$(timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'])")
  echo "pair $i: A $a | B $b"
done
bash tools/gpu_r06_steady.sh r06_mixed_steady_c
grep -E "seg_max|pillar_scatter" gpurun_out/r06_mixed_steady_c.txt | cut -c1-200
