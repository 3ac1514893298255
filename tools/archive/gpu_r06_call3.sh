#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r06_call3.txt
: > $O
echo "== kernel tests" >> $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_loss.py tests/test_tube.py -q -m gpu -k "csr or segment or compact or scatter or tube or offset or loss" 2>&1 | tail -3 >> $O
echo "== determinism" >> $O
timeout 600 python tools/r06_determinism.py --config c3_lidar --dtype mixed --runs 2 2>&1 | tail -12 >> $O
timeout 600 python tools/r06_determinism.py --config c3_lidar --dtype mixed --runs 2 --step 2>&1 | tail -12 >> $O
echo "== bench A (r05 library, atomic sums) / B (this tree)" >> $O
for i in 1 2 3; do
  a=$(PCACC_LIB=$PWD/build/r05/libpcacc_hip.so PCACC_DETERMINISTIC=0 PCACC_R05_ABI=1 timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'])")
  b=$(timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'])")
  echo "pair $i: A $a | B $b" >> $O
done
cat $O
bash tools/gpu_r06_steady.sh r06_mixed_steady_b
echo "== full GPU suite"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
