#!/bin/bash
# round 5: head convolution forward / weight gradient with the next tile's rows prefetched into registers (c_in = 32), against the previous commit's library
python -m pytest tests/test_conv.py tests/test_mixed.py -q -m gpu -x -k "head or mixed" 2>&1 | tail -2
for i in 1 2; do
echo "== this tree"; python tools/bench_headconv_ab.py 2>&1 | grep shape
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_headconv_ab.py 2>&1 | grep shape
done
