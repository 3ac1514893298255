#!/bin/bash
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_determinism.py -q -m gpu -k staged 2>&1 | grep -E "passed|failed|tensors differ" | cut -c1-600
done
echo "== whole file"
timeout 900 python -m pytest tests/test_determinism.py -q -m gpu 2>&1 | grep -E "passed|failed|tensors differ" | cut -c1-600
timeout 900 python -m pytest tests/test_determinism.py -q -m gpu 2>&1 | grep -E "passed|failed|tensors differ" | cut -c1-600
