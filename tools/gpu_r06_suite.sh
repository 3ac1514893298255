#!/bin/bash
# full GPU suite N times in a row (default 2) + smoke; every run's tail into gpurun_out/r06_gpu_suites.txt
N=${1:-2}
mkdir -p gpurun_out
: > gpurun_out/test_failures.txt
for i in $(seq 1 $N); do
  t0=$(date +%s)
  timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/suite_tail.txt
  echo "suite $i: $(tail -1 gpurun_out/suite_tail.txt)  [$(( $(date +%s) - t0 )) s]" | tee -a gpurun_out/r06_gpu_suites.txt
done
echo "test_failures.txt: $(wc -c < gpurun_out/test_failures.txt) bytes" | tee -a gpurun_out/r06_gpu_suites.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a gpurun_out/r06_gpu_suites.txt
