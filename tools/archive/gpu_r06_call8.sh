#!/bin/bash
mkdir -p gpurun_out
echo "== fused canvas variants"
timeout 600 python tools/bench_fused_canvas.py 2>&1 | tee gpurun_out/r06_fused_canvas_variants.txt
echo "== fused canvas test"
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -k "pooling_into" 2>&1 | tail -2
echo "== torch tail (mixed)"
PCACC_DTYPE=mixed timeout 600 python tools/profile_torch_tail.py 170 > gpurun_out/r06_torch_tail_mixed.txt 2>&1
sed -n 1,12p gpurun_out/r06_torch_tail_mixed.txt | cut -c1-200
