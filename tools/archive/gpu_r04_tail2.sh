#!/bin/bash
# torch-side device time of the mixed step by source line, and the standalone absmax passes
mkdir -p gpurun_out
PCACC_DTYPE=mixed timeout 600 python tools/profile_torch_tail.py 60 > gpurun_out/r04_torch_tail_mixed_v2.txt 2>&1; head -120 gpurun_out/r04_torch_tail_mixed_v2.txt | cut -c1-230
PCACC_DTYPE=mixed timeout 600 python tools/trace_absmax.py > gpurun_out/r04_absmax_mixed_v2.txt 2>&1; tail -30 gpurun_out/r04_absmax_mixed_v2.txt | cut -c1-200
