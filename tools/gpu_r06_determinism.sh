#!/bin/bash
# round 6, call 1: the fixed-order sums.  (1) kernel tests that touch what changed; (2) is the step bit-reproducible, and if not, what differs first;
# (3) what the fixed order costs: interleaved bench pairs, round-5 library + atomic few-row sums against this tree.
mkdir -p gpurun_out
O=gpurun_out/r06_determinism.txt
: > $O
echo "== kernel tests" >> $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_loss.py tests/test_tube.py tests/test_mlp_split.py -q -m gpu -k "csr or segment or compact or scatter or tube or offset or loss or wgrad or pfn or rows" 2>&1 | tail -5 >> $O
echo "== plain step, c3_lidar mixed" >> $O
timeout 600 python tools/r06_determinism.py --config c3_lidar --dtype mixed --runs 3 2>&1 | tail -70 >> $O
echo "== staged step" >> $O
timeout 600 python tools/r06_determinism.py --config c3_lidar --dtype mixed --runs 3 --step 2>&1 | tail -40 >> $O
echo "== bf16 plain" >> $O
timeout 600 python tools/r06_determinism.py --config c3_lidar --dtype bf16 --runs 2 2>&1 | tail -30 >> $O
echo "== bench A (r05 library, atomic sums) / B (this tree)" >> $O
for i in 1 2 3; do
  a=$(PCACC_LIB=$PWD/build/r05/libpcacc_hip.so PCACC_DETERMINISTIC=0 timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 20 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'])")
  b=$(timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --steps 20 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p50'])")
  echo "pair $i: A $a | B $b" >> $O
done
cat $O
