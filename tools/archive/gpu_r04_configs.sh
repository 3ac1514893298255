#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/bench_configs.py 8 2>gpurun_out/err_configs.txt | grep "^{" > gpurun_out/r04_bench_configs.jsonl; cat gpurun_out/r04_bench_configs.jsonl | cut -c1-260
timeout 600 python -m pytest tests/test_mixed.py -q -m gpu 2>&1 | tail -3
