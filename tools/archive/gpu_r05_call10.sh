#!/bin/bash
# round 5 call 10: probe; full GPU suite exactly as the driver runs it (first-touch bitmask in the batched voxeliser, tiled bilinear backward, remap, streaming scatter), smoke, 10 two-rank launches
bash tools/gpu_r05_probe.sh
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r05_suite2.txt 2>&1; tail -6 gpurun_out/r05_suite2.txt
(time python -c "import __graft_entry__ as g; g.smoke()") 2>&1 | tail -4
bash tools/gpu_r05_hang.sh 10 ";" 2>&1 | tee gpurun_out/r05_hang_consecutive.txt | tail -12
