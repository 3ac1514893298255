#!/bin/bash
# Every round-5 GPU call starts with this: the driver-style two-gloo-rank launch as the FIRST GPU work of the fresh box (the condition both round-4
# hangs had in common), watchdog at 60 s per step.  One line per trial is appended to gpurun_out/r05_hang_trials.txt (merged back by gpurun).
mkdir -p gpurun_out/r05_hang
tag=$(date +%m%d_%H%M%S)
bash tools/gpu_r05_hang.sh 1 "${1:-;}" 2>&1 | sed "s/^arm1_1/fresh_$tag/" | tee -a gpurun_out/r05_hang_trials_$tag.txt
for f in gpurun_out/r05_hang/arm1_1.*; do [ -f "$f" ] && mv "$f" "gpurun_out/r05_hang/fresh_${tag}.${f##*.}"; done
