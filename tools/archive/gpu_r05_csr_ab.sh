#!/bin/bash
# round 5: CSR build with one atomic per point (the counting pass keeps the arrival number, the fill pass needs no cursor) against the previous library
python -m pytest tests/test_hip_ops.py tests/test_prepare_batch.py -q -m gpu -x -k "csr or seg or scatter or pillar or prepare or collate or voxel or bilinear" 2>&1 | tail -2
for i in 1 2; do
echo "== this tree"; python tools/bench_csr_ab.py 2>&1 | grep case
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_csr_ab.py 2>&1 | grep case
done
