#!/bin/bash
mkdir -p gpurun_out
bash tools/gpu_r06_steady.sh r06_mixed_steady_a
echo "== new tests"
timeout 900 python -m pytest tests/test_determinism.py tests/test_bench_multirank.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "compact_mask or csr_order" 2>&1 | tail -2
