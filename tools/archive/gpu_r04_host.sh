#!/bin/bash
mkdir -p gpurun_out
BATCH=1 PCACC_DTYPE=mixed timeout 600 python tools/host_profile.py > gpurun_out/r04_host_profile_b1.txt 2>&1; head -90 gpurun_out/r04_host_profile_b1.txt | cut -c1-180
