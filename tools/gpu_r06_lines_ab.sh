#!/bin/bash
# round 6: the resident fp32x3 kernel with whole-line stores (this tree, CSR_LINES = 1) against the build before it (build/lines0): digests, layer times, step times
OLD=$PWD/build/${1:-lines0}/libpcacc_hip.so
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_conv_split.py tests/test_mixed.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2; do
echo "== this tree"; timeout 300 python tools/bench_conv_ab.py 2>&1 | grep layer
echo "== before"; PCACC_LIB=$OLD timeout 300 python tools/bench_conv_ab.py 2>&1 | grep layer
done
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2), "roofline frac", round(d["roofline"]["frac"], 3))
PY
}
for i in 1 2 3; do for v in new before; do
  if [ $v = before ]; then export PCACC_LIB=$OLD; else unset PCACC_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r06_lines_ab_$v.json 2> gpurun_out/r06_lines_ab_$v.err
  show gpurun_out/r06_lines_ab_$v.json "mixed, library $v"
done; done
} 2>&1 | tee gpurun_out/${2:-r06_lines_ab}.txt
