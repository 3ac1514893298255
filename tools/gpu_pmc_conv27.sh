#!/bin/bash
# counters of the 27-tap 32 -> 32 layer (20 frames of 288^2): frame-major against frame-fastest tile order, swizzled rows
mkdir -p gpurun_out/pmc_conv27
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for order in fast major; do
  if [ $order = major ]; then export PCACC_CONV_FRAME_MAJOR=1; else unset PCACC_CONV_FRAME_MAJOR; fi
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
    tag=${order}_$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_conv27/$tag -o p -- python3 $R/tools/pmc_conv.py 32 32 288 20 3 > $R/gpurun_out/pmc_conv27/$tag.log 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for order in ('major', 'fast'):
    agg=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pmc_conv27/%s_*/*counter_collection.csv' % order):
        for r in csv.DictReader(open(f)):
            if 'conv3x3_resident' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(order, {k: round(sum(v)/len(v), 1) for k, v in sorted(agg.items())})
PY
