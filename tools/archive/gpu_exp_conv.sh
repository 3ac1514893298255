#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for v in ""; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipcaccumulation_amd/csrc -Iinclude $v tools/exp_conv_deep.hip -o /tmp/exp 2>/dev/null || { echo "build failed: $v"; continue; }
  echo "== variant: [$v]"
  export PCACC_CONV_PLAN=1; for s in "20 72 72 128 128" "20 36 36 256 256" "20 18 18 512 512" "20 36 36 512 256" "20 72 72 256 128" "20 144 144 128 64" "20 18 18 256 512" "4 288 288 128 64" "4 72 72 256 128" "4 36 36 128 128" "4 18 18 256 256" "5 72 72 128 128"; do /tmp/exp $s; done
done
