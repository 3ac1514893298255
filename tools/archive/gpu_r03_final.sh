#!/bin/bash
# end of round 3: full GPU suite, smoke, default bench (+ lidar points), steady-state kernel stats and call tables of both modes, 27-tap counters
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/final_tests.log; tail -2 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_default_bench.sh
timeout 900 python bench.py --points lidar --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > gpurun_out/bench_lidar.json; cut -c1-260 gpurun_out/bench_lidar.json
bash tools/gpu_r03_profiles.sh
bash tools/gpu_pmc_conv27.sh | tail -60
timeout 300 python tools/bench_conv_swz.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_conv_swz.log
