#!/bin/bash
# end-of-round soak on the final tree: the GPU suite as the driver runs it, three times; smoke after each; 60 consecutive two-rank launches
bash tools/gpu_r05_probe.sh
rm -f gpurun_out/test_failures.txt
for i in 1 2 3; do
  python -m pytest tests -x -q -m gpu 2>&1 | tail -1
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-80
done
echo "first-draw failures logged: $(grep -c 'FIRST DRAW' gpurun_out/test_failures.txt 2>/dev/null || echo 0)"
bash tools/gpu_r05_hang.sh 60 ";" 2>&1 | tee gpurun_out/r05_hang_consecutive60.txt | awk '{print $2}' | sort | uniq -c
