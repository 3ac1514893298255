#!/bin/bash
# counters of the 27-tap 32 -> 32 layer (20 frames of 288^2) in three builds of the same kernel: padded LDS rows (one workgroup per CU),
# swizzled rows (two per CU) with frame-major tiles, swizzled rows with frame-fastest tiles (the default).  One counter set per run.
mkdir -p gpurun_out/pmc_conv27
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in padded major fast; do
  unset PCACC_CONV_FRAME_MAJOR PCACC_CONV_SWZ_OFF
  if [ $v = padded ]; then export PCACC_CONV_SWZ_OFF=1 PCACC_CONV_FRAME_MAJOR=1; fi
  if [ $v = major ]; then export PCACC_CONV_FRAME_MAJOR=1; fi
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
    tag=${v}_$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_conv27/$tag -o p -- python3 $R/tools/pmc_conv.py 32 32 288 20 3 > $R/gpurun_out/pmc_conv27/$tag.log 2>&1
  done
done
cd $R
python3 - <<'PY' | tee gpurun_out/pmc_conv27_summary.txt
import csv, glob, collections
print('27-tap 32 -> 32 convolution, 20 frames of 288 x 288 (tools/pmc_conv.py 32 32 288 20 3), rocprofv3 --pmc, one counter set per run, mean of 5 launches;')
print('FETCH_SIZE / WRITE_SIZE in KiB (uncorrected); *_sum over the device; us = kernel duration under counter collection')
for v, what in (('padded', 'padded 80-byte LDS rows, 96 KB: one 8-wave workgroup per CU; frame-major tiles'),
                ('major', 'swizzled 64-byte rows, 77 KB: two workgroups per CU; frame-major tiles'),
                ('fast', 'swizzled rows, two workgroups per CU; frame-fastest tiles (default)')):
    agg = collections.defaultdict(list); dur = []
    for f in glob.glob('gpurun_out/pmc_conv27/%s_*/*counter_collection.csv' % v):
        for r in csv.DictReader(open(f)):
            if 'conv3x3_resident' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob('gpurun_out/pmc_conv27/%s_*/*kernel_trace.csv' % v):
        for r in csv.DictReader(open(f)):
            if 'conv3x3_resident' in r['Kernel_Name']:
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('\n%s  [%.1f us]' % (what, sum(dur) / max(len(dur), 1)))
    for k, x in sorted(agg.items()):
        print('  %-32s %14.1f' % (k, sum(x) / len(x)))
PY
