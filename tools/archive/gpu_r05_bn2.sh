#!/bin/bash
# round 5: BatchNorm passes with several rows / pieces per lane in flight (same arithmetic, same order): tests, warm A/B with digests, then the kernels INSIDE the step
python -m pytest tests/test_hip_ops.py tests/test_mixed.py -q -m gpu -x -k "bn or batch_norm or mixed" 2>&1 | tail -2
echo "== this tree"; python tools/bench_bn_ab.py 2>&1 | grep rows
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_bn_ab.py 2>&1 | grep rows
bash tools/gpu_r05_instep_ab.sh "bn_\|bilinear_gather_bwd\|csr_"
