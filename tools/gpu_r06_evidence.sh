#!/bin/bash
# round 6 evidence on the current tree: default bench as the driver runs it; the same command under rocprofv3 --kernel-trace --stats; steady-state kernel
# summaries (mixed with the idle-gap / per-stream analysis, bf16); counter passes for the pillar-scatter kernels and the kernel set; torch tail
mkdir -p gpurun_out
bash tools/gpu_default_bench.sh; cp gpurun_out/bench_default_full.json gpurun_out/r06_bench_default_full.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -o bench -- python3 $R/bench.py --no-cpu-baseline --no-configs > $R/gpurun_out/rocprof_default.log 2>&1
cp $(find $R/gpurun_out/prof_default -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06_default_kernel_stats.csv
grep -n "seg_max_canvas" $R/gpurun_out/r06_default_kernel_stats.csv | cut -c1-220
grep "^{" $R/gpurun_out/rocprof_default.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('under rocprofv3:', round(d['ms_per_step'],2), 'roofline kernel avg us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
rm -rf $R/gpurun_out/prof_default
cd $R
bash tools/gpu_r06_steady.sh r06_mixed_steady > /dev/null 2>&1
head -8 gpurun_out/r06_mixed_steady.txt | cut -c1-190; sed -n '/^GPU active/,/^idle gaps/p' gpurun_out/r06_mixed_steady.txt | cut -c1-250; sed -n '/^per /,$p' gpurun_out/r06_mixed_steady.txt | cut -c1-250
PCACC_BENCH_EXTRA="--dtype bf16" bash tools/gpu_r06_steady.sh r06_bf16_steady > /dev/null 2>&1
head -8 gpurun_out/r06_bf16_steady.txt | cut -c1-190
mkdir -p gpurun_out/pmc_scatter
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_scatter/$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_scatter/$c -o p -- python3 $R/tools/pmc_scatter.py > $R/gpurun_out/pmc_scatter/$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_scatter/$c.log
done
cd $R
F=$(find gpurun_out/pmc_scatter/FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/pmc_scatter/WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W gpurun_out/r06_pmc_scatter_summary.json | grep -E "traffic_over|what" | head
bash tools/gpu_pmc_kernels.sh > gpurun_out/pmc_kernels.log 2>&1; tail -3 gpurun_out/pmc_kernels.log | cut -c1-200
cp gpurun_out/pmc_kernels_summary.json gpurun_out/r06_pmc_kernels_summary.json
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r06_pmc_kernels_summary.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'hbm_over_algorithmic' in v:
        print('%-44s %7.1f us  hbm/alg %.2f  lds conflict share %s' % (k[:44], v.get('avg_us_under_pmc', 0), v['hbm_over_algorithmic'], v.get('lds_conflict_share')))
PY
PCACC_DTYPE=mixed timeout 600 python tools/profile_torch_tail.py 170 > gpurun_out/r06_torch_tail_mixed.txt 2>&1
sed -n 4,6p gpurun_out/r06_torch_tail_mixed.txt | cut -c1-200
