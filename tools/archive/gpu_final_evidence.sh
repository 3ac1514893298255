#!/bin/bash
# evidence files for profiles/ on the final code: bf16 / fp32 deltas against the reference's golden vectors, kernel trace of the bench
mkdir -p gpurun_out; rm -f gpurun_out/bf16_deltas.jsonl
PCACC_DUMP_DELTAS=1 timeout 1200 python -m pytest tests/test_config_parity.py -q -p no:cacheprovider 2>&1 | tail -2 | cut -c1-200
wc -l gpurun_out/bf16_deltas.jsonl
bash tools/gpu_profile_bench.sh > /dev/null 2>&1; head -9 gpurun_out/bench_summary.txt; tail -3 gpurun_out/bench_timeline.txt | cut -c1-200
