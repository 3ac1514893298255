#!/bin/bash
for i in 1 2 3 4 5 6 7 8; do timeout 300 python tools/r06_diag_headconv3.py 2>&1 | grep "calls," ; done
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_determinism.py -q -m gpu -k staged 2>&1 | grep -E "passed|failed|tensors differ" | cut -c1-400; done
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_conv.py -q -m gpu -k "head_conv or head" 2>&1 | tail -2
timeout 300 python tools/bench_head_conv.py 2>&1 | tail -12 | cut -c1-200
