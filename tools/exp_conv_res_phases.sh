#!/bin/bash
# which phase of conv3x3_split_res_kernel costs what: rebuild conv_split.hip with phase knock-outs (CSR_EXP bits) into scratch libraries
cd pcaccumulation_amd/csrc
for e in 0 1 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DCSR_EXP=$e -c conv_split.hip -o /tmp/conv_split_exp$e.o
  objs=$(ls *.o | grep -v conv_split.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libpcacc_exp$e.so $objs /tmp/conv_split_exp$e.o
done
cd ../..
for e in 0 1 4; do
  PCACC_LIB=/tmp/libpcacc_exp$e.so python tools/exp_conv_res_phases.py 2>&1 | tail -1
done
