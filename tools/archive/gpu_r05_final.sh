#!/bin/bash
# round 5, end: probe; the GPU suite and smoke exactly as the driver runs them; 30 consecutive two-rank launches; the evidence script
bash tools/gpu_r05_probe.sh
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r05_suite_final.txt 2>&1; tail -6 gpurun_out/r05_suite_final.txt
(time python -c "import __graft_entry__ as g; g.smoke()") 2>&1 | tail -4
bash tools/gpu_r05_hang.sh ${1:-30} ";" 2>&1 | tee gpurun_out/r05_hang_consecutive_final.txt | tail -32
bash tools/gpu_r05_evidence.sh 2>&1 | tail -80
