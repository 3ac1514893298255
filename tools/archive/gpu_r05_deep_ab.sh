#!/bin/bash
# round 5: conv_deep.hip with batched loads (strip forward / data gradient, strip weight gradient) against the previous commit's build; generic fp32x3 kernel likewise
bash tools/gpu_r05_probe.sh
python -m pytest tests/test_conv.py tests/test_conv_split.py tests/test_mixed.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2; do
echo "== this tree"; python tools/bench_deep_ab.py 2>&1 | grep layer; python tools/bench_conv_ab.py 2>&1 | grep -E "144|72"
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_deep_ab.py 2>&1 | grep layer; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_conv_ab.py 2>&1 | grep -E "144|72"
done
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2))
PY
}
for i in 1 2 3; do for v in new prev; do for dt in mixed bf16; do
  if [ $v = prev ]; then export PCACC_LIB=$PWD/build/libpcacc_hip_prev.so; else unset PCACC_LIB; fi
  python bench.py --dtype $dt --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_ab2_$v.json 2> gpurun_out/r05_bench_ab2_$v.err
  show gpurun_out/r05_bench_ab2_$v.json "$dt, library $v"
done; done; done
