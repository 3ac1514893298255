#!/bin/bash
mkdir -p gpurun_out
echo "== trajectory mixed-c1_i1 verbose"
PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1" -s 2>&1 | grep -v "^$" | tail -40 | cut -c1-1500
echo "== same with the r05 library and the atomic sums"
PCACC_LIB=$PWD/build/r05/libpcacc_hip.so PCACC_DETERMINISTIC=0 PCACC_R05_ABI=1 PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1_i1" -s 2>&1 | grep -v "^$" | tail -14 | cut -c1-1500
echo "== rest of the suite behind the failure"
timeout 900 python -m pytest tests/test_train_trajectory.py tests/test_bench_multirank.py -q -m gpu 2>&1 | tail -6
echo "== chamfer variants"
timeout 300 build/exp_chamfer 2>&1 | tee gpurun_out/r06_chamfer_variants.txt
echo "== precision map"
PCACC_LIB=$PWD/build/x3exp/libpcacc_hip.so PCACC_BATCH_PREPARE=0 timeout 1500 python tools/r06_precision_map.py > gpurun_out/r06_precision_map.txt 2> gpurun_out/r06_precision_map.err
tail -5 gpurun_out/r06_precision_map.err | cut -c1-300
head -60 gpurun_out/r06_precision_map.txt
