#!/bin/bash
# round 5: fp32x3 row-linear kernels with the tile's loads issued back to back (this tree) against the previous commit's build
bash tools/gpu_r05_probe.sh
python -m pytest tests/test_mlp_split.py tests/test_mixed.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2; do
echo "== this tree"; python tools/bench_rows_ab.py 2>&1 | grep layer
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_rows_ab.py 2>&1 | grep layer
done
show() { python - <<PY
import json
d = json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", round(d["ms_per_step"], 2), "p10/p50/p90", round(d["ms_per_step_p10"], 2), round(d["ms_per_step_p50"], 2), round(d["ms_per_step_p90"], 2))
PY
}
for i in 1 2 3; do for v in new prev; do
  if [ $v = prev ]; then export PCACC_LIB=$PWD/build/libpcacc_hip_prev.so; else unset PCACC_LIB; fi
  python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r05_bench_ab3_$v.json 2> gpurun_out/r05_bench_ab3_$v.err
  show gpurun_out/r05_bench_ab3_$v.json "mixed, library $v"
done; done
