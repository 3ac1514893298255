#!/bin/bash
# the GPU suite and smoke exactly as the driver runs them (fresh box), tail kept
bash tools/gpu_r05_probe.sh
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r05_suite_final.txt 2>&1; tail -8 gpurun_out/r05_suite_final.txt
(time python -c "import __graft_entry__ as g; g.smoke()") 2>&1 | tail -4
[ -f gpurun_out/test_failures.txt ] && grep -c "FIRST DRAW" gpurun_out/test_failures.txt
