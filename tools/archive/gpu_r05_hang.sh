#!/bin/bash
# round 5, verdict item 1(d): the two-gloo-ranks-on-one-GPU launch as the FIRST GPU work of a fresh box, watchdog on (stacks of every thread when a step
# takes > 60 s), then N more launches per arm.  usage: tools/gpu_r05_hang.sh <launches per arm> "<arm1 env;flags>" ...
# Output: gpurun_out/r05_hang/<arm>_<i>.{out,err}, one summary line per launch on stdout.
mkdir -p gpurun_out/r05_hang
N=${1:-5}; shift
launch() { # name, env string, flags
  local name=$1 envs=$2 flags=$3 t0=$(date +%s.%N)
  env PCACC_DIST_BACKEND=gloo PCACC_HANG_DUMP=60 PCACC_BENCH_TRACE=1 $envs timeout -k 5 260 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 2000)) bench.py --gpus 2 --steps 3 --warmup 2 --batch 2 --no-cpu-baseline --no-fp32-leg $flags \
    > gpurun_out/r05_hang/$name.out 2> gpurun_out/r05_hang/$name.err
  local rc=$? t1=$(date +%s.%N)
  local ms=$(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05_hang/$name.out | head -1)
  echo "$name rc=$rc wall=$(python3 -c "print(round($t1-$t0,1))")s $ms left_behind=$(pgrep -c -f 'bench.py --gpus 2')"
  if [ $rc -ne 0 ]; then grep -n "Timeout\|File \|Thread\|bench rank" gpurun_out/r05_hang/$name.err | tail -60; fi
  [ $rc -eq 0 ] && rm -f gpurun_out/r05_hang/$name.out
}
arm=0
for spec in "$@"; do
  arm=$((arm+1))
  envs=${spec%%;*}; flags=${spec#*;}
  for i in $(seq 1 $N); do launch "arm${arm}_$i" "$envs" "$flags"; done
done
