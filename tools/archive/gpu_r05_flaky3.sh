#!/bin/bash
# repeat the whole-model fixture file N times in fresh processes; every failing test's report lands in gpurun_out/test_failures.txt
rm -f gpurun_out/test_failures.txt
for i in $(seq 1 ${1:-20}); do
  python -m pytest tests/test_config_parity.py -q -m gpu 2>&1 | tail -1
done
[ -f gpurun_out/test_failures.txt ] && cut -c1-700 gpurun_out/test_failures.txt | head -120
