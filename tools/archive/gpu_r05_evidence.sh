#!/bin/bash
# round 5 evidence on the current tree: probe; default bench as the driver runs it; the same command under rocprofv3 --kernel-trace --stats; steady-state
# kernel summaries (mixed, bf16); counter passes for the pillar-scatter kernels and the kernel set; per-config table
mkdir -p gpurun_out
# (probe: run by the caller)
bash tools/gpu_default_bench.sh; cp gpurun_out/bench_default_full.json gpurun_out/r05_bench_default_full.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -o bench -- python3 $R/bench.py --no-cpu-baseline --no-configs > $R/gpurun_out/rocprof_default.log 2>&1
cp $R/gpurun_out/prof_default/bench_kernel_stats.csv $R/gpurun_out/r05_default_kernel_stats.csv
grep -n "pillar_scatter" $R/gpurun_out/prof_default/bench_kernel_stats.csv | cut -c1-220
grep "^{" $R/gpurun_out/rocprof_default.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('under rocprofv3:', round(d['ms_per_step'],2), 'scatter avg us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
for d in mixed bf16; do
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$d -o bench -- python3 $R/bench.py --dtype $d --steps 5 --warmup 5 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $R/gpurun_out/rocprof_$d.log 2>&1
python3 $R/tools/kstats_steady.py $R/gpurun_out/prof_$d/bench_kernel_trace.csv 6 200 > $R/gpurun_out/r05_${d}_steady.txt; head -9 $R/gpurun_out/r05_${d}_steady.txt | cut -c1-190
done
cd $R
mkdir -p gpurun_out/pmc_scatter
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_scatter/$c -o p -- python3 $R/tools/pmc_scatter.py > $R/gpurun_out/pmc_scatter/$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_scatter/$c.log
done
cd $R
F=$(find gpurun_out/pmc_scatter/FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/pmc_scatter/WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W gpurun_out/r05_pmc_scatter_summary.json | grep -E "traffic_over|what" | head
bash tools/gpu_pmc_kernels.sh > gpurun_out/pmc_kernels.log 2>&1; tail -3 gpurun_out/pmc_kernels.log | cut -c1-200
cp gpurun_out/pmc_kernels_summary.json gpurun_out/r05_pmc_kernels_summary.json
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r05_pmc_kernels_summary.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'hbm_over_algorithmic' in v:
        print('%-44s hbm/alg %.2f  lds conflict share %s' % (k[:44], v['hbm_over_algorithmic'], v.get('lds_conflict_share')))
PY
