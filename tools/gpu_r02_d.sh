#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/bf16_deltas.jsonl
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv.py -q -p no:cacheprovider 2>&1 | tail -8 | cut -c1-300
echo "=== bench_conv"; timeout 600 python tools/bench_conv.py 2>&1 | grep '^{' | tee gpurun_out/bench_conv.jsonl | cut -c1-260
PCACC_DUMP_DELTAS=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_conv.py 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log | cut -c1-300
echo "=== bench (no cpu baseline)"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_nocpu.json | cut -c1-1200
