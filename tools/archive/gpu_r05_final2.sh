#!/bin/bash
# round 5, last call: probe; the GPU suite and smoke exactly as the driver runs them; the evidence script (default bench, rocprofv3 stats, steady summaries, counters)
bash tools/gpu_r05_probe.sh
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r05_suite_final.txt 2>&1; tail -6 gpurun_out/r05_suite_final.txt
(time python -c "import __graft_entry__ as g; g.smoke()") 2>&1 | tail -4
bash tools/gpu_r05_evidence.sh 2>&1 | tail -80
