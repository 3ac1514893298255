#!/bin/bash
bash tools/gpu_r06_steady.sh r06_streams > /dev/null 2>&1
sed -n '/^per /,$p' gpurun_out/r06_streams.txt | cut -c1-200
