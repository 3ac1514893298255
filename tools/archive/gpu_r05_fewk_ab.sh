#!/bin/bash
bash tools/gpu_r05_probe.sh
python -m pytest tests/test_hip_ops.py tests/test_mixed.py -q -m gpu -x -k "rows or few or linear or pfn or mixed" 2>&1 | tail -2
for i in 1 2; do
echo "== this tree"; python tools/bench_fewk_ab.py 2>&1 | grep layer
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_fewk_ab.py 2>&1 | grep layer
done
