#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/bench_chunked_chain.py 2>&1 | grep -v Warn | tee gpurun_out/r06_chunked_chain.txt
