#!/bin/bash
# round 5: the instrumented extra step of bench.py (roofline_step) at world size 2.  Before the fix only rank 0 took it -- a training step whose all-reduces
# no peer answered.  Old file (bench_prev_tmp.py = the previous commit's bench.py, not committed) against the new one, gloo ranks sharing the GPU, trace on.
run() { # file
  local t0=$(date +%s.%N)
  env PCACC_DIST_BACKEND=gloo PCACC_HANG_DUMP=60 PCACC_BENCH_TRACE=1 timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 2000)) $1 --gpus 2 --steps 3 --warmup 2 --batch 2 --no-cpu-baseline --no-fp32-leg > gpurun_out/unm.out 2> gpurun_out/unm.err
  local rc=$? t1=$(date +%s.%N)
  echo "== $1 rc=$rc wall=$(python3 -c "print(round($t1-$t0,1))")s"
  grep "bench rank" gpurun_out/unm.err | tail -6
  grep -i "error\|closed\|reset\|skipped\|Timeout" gpurun_out/unm.err | head -8
  python3 - <<'PY'
import json
for l in open('gpurun_out/unm.out'):
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step', round(d['ms_per_step'], 2), 'roofline_step', {k: (v if not isinstance(v, dict) else '...') for k, v in d.get('roofline_step', {}).items() if k != 'model'})
PY
}
[ -f bench_prev_tmp.py ] && run bench_prev_tmp.py
run bench.py
run bench.py
