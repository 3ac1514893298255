#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r06_determinism_fp32_det.txt
: > $O
for c in c3_lidar c2; do
  echo "== $c fp32, cudnn.deterministic" >> $O
  PCACC_CUDNN_DET=1 timeout 300 python tools/r06_determinism.py --config $c --dtype fp32 --runs 4 2>&1 | tail -12 >> $O
done
cat $O | cut -c1-200
