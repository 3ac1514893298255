#!/bin/bash
# round 4, verdict item 5: two gloo ranks on one GPU with the staged two-stream step (PCACC_TWO_STREAMS_DIST=1); variants of the prefetch hand-over
mkdir -p gpurun_out
run() { echo "== $1 | $2"; env $1 bash tools/gpu_2rank.sh "$2"; }
run "PCACC_X=0" "--dtype bf16"
run "PCACC_TWO_STREAMS_DIST=1" "--dtype bf16 --pipeline"
run "PCACC_TWO_STREAMS_DIST=1 PCACC_PREFETCH_WAIT=poll" "--dtype bf16 --pipeline"
run "PCACC_TWO_STREAMS_DIST=1" "--dtype bf16 --pipeline --no-prefetch"
run "PCACC_TWO_STREAMS_DIST=1 GPU_MAX_HW_QUEUES=2" "--dtype bf16 --pipeline"
run "PCACC_TWO_STREAMS_DIST=1 GPU_MAX_HW_QUEUES=8" "--dtype bf16 --pipeline"
run "PCACC_TWO_STREAMS_DIST=1 PCACC_PREFETCH_WAIT=poll" "--dtype mixed --pipeline"
run "PCACC_X=0" "--dtype mixed"
