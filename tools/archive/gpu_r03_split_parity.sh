#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/exp_x3_vs_fp32_terms.py c5 2>&1 | grep -v "Warning\|detach\|amdgpu" > gpurun_out/x3_vs_fp32_terms.txt
PCACC_DUMP_DELTAS=1 timeout 1500 python -m pytest tests/test_config_parity.py -q -k "fp32x3" 2>&1 | tail -8
grep fp32x3 gpurun_out/bf16_deltas.jsonl > gpurun_out/x3_deltas.jsonl
