#!/bin/bash
# round 6: side stream in the high-priority class (PCACC_SIDE_PRIORITY=-1) against the default class, interleaved
mkdir -p gpurun_out
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2))
PY
}
{
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for i in 1 2 3; do for v in -1 0; do
  PCACC_SIDE_PRIORITY=$v timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r06_prio_$v.json 2> gpurun_out/r06_prio_$v.err
  show gpurun_out/r06_prio_$v.json "mixed, side stream priority $v"
done; done
timeout 600 python -m pytest tests/test_determinism.py -q -m gpu -x -k staged 2>&1 | tail -1
} 2>&1 | tee gpurun_out/r06_side_priority_ab.txt
PCACC_SIDE_PRIORITY=-1 bash tools/gpu_r06_steady.sh r06_prio_steady > /dev/null 2>&1
sed -n '/^GPU active/,/^idle gaps/p' gpurun_out/r06_prio_steady.txt | cut -c1-250 | tee -a gpurun_out/r06_side_priority_ab.txt
sed -n '/^per /,$p' gpurun_out/r06_prio_steady.txt | cut -c1-250 | tee -a gpurun_out/r06_side_priority_ab.txt
