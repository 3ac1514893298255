#!/bin/bash
mkdir -p gpurun_out
echo "== trajectory mixed-c1_i1 verbose (this tree)"
PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1_i1" -s 2>&1 | grep -A14 "term rel per step" | head -40
echo "== r05 lib"
PCACC_LIB=$PWD/build/r05/libpcacc_hip.so PCACC_DETERMINISTIC=0 PCACC_FUSED_CANVAS=0 PCACC_R05_ABI=1 PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1_i1" -s 2>&1 | grep -A14 "term rel per step" | head -40
echo "== mixed2 parity"
timeout 900 python -m pytest tests/test_config_parity.py -q -m gpu -k "mixed2" 2>&1 | tail -4
echo "== bench mixed vs mixed2"
for i in 1 2; do
  a=$(timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'])")
  b=$(timeout 600 python bench.py --dtype mixed2 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'])")
  echo "pair $i: mixed $a | mixed2 $b"
done
