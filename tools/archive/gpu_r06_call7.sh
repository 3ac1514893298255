#!/bin/bash
mkdir -p gpurun_out
echo "== trajectory mixed-c1_i1 verbose (this tree)"
PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1_i1" -s 2>&1 | grep -A14 "term rel per step" | head -40
echo "== r05 lib"
PCACC_LIB=$PWD/build/r05/libpcacc_hip.so PCACC_DETERMINISTIC=0 PCACC_FUSED_CANVAS=0 PCACC_R05_ABI=1 PCACC_TRAJ_VERBOSE=1 timeout 600 python -m pytest "tests/test_train_trajectory.py" -q -m gpu -k "mixed and c1_i1" -s 2>&1 | grep -A14 "term rel per step" | head -40
echo "== chamfer tests"
timeout 600 python -m pytest tests -q -m gpu -k "chamfer" 2>&1 | tail -3
echo "== PMC scatter"
mkdir -p gpurun_out/pmc_scatter
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_scatter/$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_scatter/$c -o p -- python3 $R/tools/pmc_scatter.py > $R/gpurun_out/pmc_scatter/$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_scatter/$c.log
done
cd $R
F=$(find gpurun_out/pmc_scatter/FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/pmc_scatter/WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W gpurun_out/r06_pmc_scatter_summary.json | tail -12
rm -rf gpurun_out/pmc_scatter
echo "== default bench"
timeout 1500 python bench.py > gpurun_out/r06_bench_default_a.json 2> gpurun_out/r06_bench_default_a.err
tail -2 gpurun_out/r06_bench_default_a.err | cut -c1-300
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_default_a.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_p50','dtype')})
print('roofline', {k:d['roofline'].get(k) for k in ('kernel','achieved','frac','traffic','avg_launch_us','algorithmic_bytes_per_launch','cold_cache')})
print('configs', [(c.get('config'), c.get('ms_per_step'), c.get('points')) for c in d.get('configs',[])])
print('bf16', d.get('bf16',{}).get('ms_per_step'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
