#!/bin/bash
# Regenerates the MIOpen user find-db the bf16 headline's four transposed convolutions (and the fp32 parity mode's convolutions) use:
# run on a GPU box (gpurun -- 'bash tools/regen_miopen_finddb.sh'), then unpack gpurun_out/miopen_cache.tgz at the repo root.
# The directory is git-ignored (binary kernel cache) but travels with the working tree; without it the first warm-up steps run MIOpen's
# solver search (seconds, not a different result).
set -e
mkdir -p gpurun_out
rm -rf .miopen_cache
timeout 1500 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > /dev/null
timeout 1500 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --points lidar > /dev/null
timeout 1500 python -m pytest tests/test_config_parity.py -q -m gpu -k "fp32-" > /dev/null || true     # the fp32 parity mode's library convolutions
tar czf gpurun_out/miopen_cache.tgz .miopen_cache
ls -la .miopen_cache
