#!/bin/bash
# verdict item 2: the production step WITH a process group (backend nccl = RCCL, world size 1 under the driver's launcher, PCACC_FORCE_PROCESS_GROUP=1: every gradient
# bucket through ncclAllReduce, agreement reduce, staged two-stream step) against the same step without a process group; three interleaved pairs
bash tools/gpu_r05_probe.sh
show() { python - <<PY
import json
d = json.loads([l for l in open("$1").read().splitlines() if l.startswith("{")][-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), d["distributed"]["backend"], "collectives/step", d["distributed"]["collectives_per_step"], d["config"]["step_variant"][:40])
PY
}
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_rccl_off.json 2> gpurun_out/r05_rccl_off.err
  show gpurun_out/r05_rccl_off.json "no process group   "
  PCACC_FORCE_PROCESS_GROUP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 2000)) bench.py --gpus 1 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_rccl_on.json 2> gpurun_out/r05_rccl_on.err
  show gpurun_out/r05_rccl_on.json "nccl, world size 1 "
done
