#!/bin/bash
# round 5: few-feature row kernels (forward store mapping, weight-gradient straight-line loads) -- kernel A/B and the headline step, this tree against the previous commit's library
python -m pytest tests/test_hip_ops.py tests/test_mixed.py -q -m gpu -x 2>&1 | tail -2
echo "== this tree"; python tools/bench_fewk_ab.py 2>&1 | grep "layer\|wgrad" | cut -c1-330
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_fewk_ab.py 2>&1 | grep "layer\|wgrad" | cut -c1-330
for i in 1; do
for lib in "" $PWD/build/libpcacc_hip_prev.so; do
  ms=$(PCACC_LIB=$lib timeout 900 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['ms_per_step_p50'],2))")
  echo "bench lib=${lib:-tree} ms_per_step, p50: $ms"
done
done
