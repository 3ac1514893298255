#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -p no:cacheprovider -k "sinkhorn" 2>&1 | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_model_parity.py tests/test_config_parity.py -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
timeout 600 python tools/native_call_table.py 40 2>&1 | grep -v Warn > gpurun_out/native_calls.txt; grep -n "sinkhorn" gpurun_out/native_calls.txt | cut -c1-160
