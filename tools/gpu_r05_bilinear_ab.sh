#!/bin/bash
# round 5: sorted bilinear backward with the first point of every tap segment fetched up front -- warm / cold A/B against the previous library, then the kernel in the step
python -m pytest tests/test_hip_ops.py tests/test_parity_ops.py -q -m gpu -x -k "bilinear or ungrid or gather" 2>&1 | tail -2
for i in 1 2; do
echo "== this tree"; python tools/bench_bilinear_ab.py 2>&1 | grep case
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_bilinear_ab.py 2>&1 | grep case
done
cd /tmp && export TMPDIR=/tmp
for lib in "" $GRAFT_REPO_ROOT/build/libpcacc_hip_prev.so; do
rm -rf /tmp/prof_bl; PCACC_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bl -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > /tmp/rocprof_bl.log 2>&1
echo "== in the step, lib=${lib:-tree}"; grep -h "bilinear_gather_bwd_sorted\|bilinear_sorted_prep" $(find /tmp/prof_bl -name "*kernel_stats.csv") | sed "s/^\"void \([a-z_]*\)[^\"]*\",/\1,/" | cut -d, -f1-5
done
