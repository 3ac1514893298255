#!/bin/bash
# fp32x3 convolution kernels: parity tests, then per-layer timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_split.py -x -q 2>&1 | tail -15
timeout 600 python tools/bench_conv_split.py > gpurun_out/bench_conv_split.jsonl 2> gpurun_out/bench_conv_split.err; tail -3 gpurun_out/bench_conv_split.err; cat gpurun_out/bench_conv_split.jsonl | cut -c1-260
