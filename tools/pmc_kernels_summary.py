#!/usr/bin/env python
"""profiles/rNN_pmc_kernels_summary.json from the counter files of tools/gpu_pmc_kernels.sh (one rocprofv3 --pmc run per counter set).
Per kernel (mean over its launches): duration, MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs),
achieved TFLOP/s from the algorithmic FLOPs, LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, HBM bytes
(FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950 correction + WRITE_SIZE, KiB units).
Usage: pmc_kernels_summary.py <dir with one sub-directory per counter set> out.json"""
import collections
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
KERNELS = {  # name fragment -> (label, algorithmic FLOPs per launch, algorithmic HBM bytes per launch)
    'conv3x3_resident_kernel': ('conv 64->64 @288^2 x20 (resident weights)', 2.0 * 20 * 288 * 288 * 64 * 64 * 9, 20 * 288 * 288 * 128 * 2),
    'conv3x3_strip_kernel': ('conv 256->256 @36^2 x20 (deep, strips)', 2.0 * 20 * 36 * 36 * 256 * 256 * 9, 20 * 36 * 36 * 512 * 2 + 9 * 256 * 256 * 2),
    'conv3x3_wgrad_kernel': ('conv wgrad 32x32 @288^2 x20', 2.0 * 20 * 288 * 288 * 32 * 32 * 9, 20 * 288 * 288 * 64 * 2),
    'conv3x3_wgrad_strip_kernel': ('conv wgrad 256x256 @36^2 x20 (deep)', 2.0 * 20 * 36 * 36 * 256 * 256 * 9, 20 * 36 * 36 * 512 * 2),
    'rows_wgrad_bf16_kernel': ('rows wgrad 32x32, 3.2 M rows', 2.0 * 3.2e6 * 32 * 33, 3.2e6 * 64 * 2),
    'rows_linear_bf16_kernel': ('rows linear 64->32, 3.2 M rows', 2.0 * 3.2e6 * 64 * 32, 3.2e6 * 96 * 2),
    # fp32x3 (round 3): algorithmic FLOPs of the operation (the kernels issue 3 MFMAs per product: MFMA busy / 3 is the algorithmic share)
    'conv3x3_split_res_kernel': ('fp32x3 conv 32->32 @288^2 x20 (resident, persistent)', 2.0 * 20 * 288 * 288 * 32 * 32 * 9, 20 * 288 * 288 * 64 * 4),
    'conv3x3_split_kernel': ('fp32x3 conv 256->256 @36^2 x20 (streaming)', 2.0 * 20 * 36 * 36 * 256 * 256 * 9, 20 * 36 * 36 * 512 * 4 + 9 * 256 * 256 * 4),
    'conv3x3_wgrad_split_kernel<1, 1': ('fp32x3 conv wgrad 32x32 @288^2 x20', 2.0 * 20 * 288 * 288 * 32 * 32 * 9, 20 * 288 * 288 * 64 * 4),
    'conv3x3_wgrad_split_kernel<2, 2': ('fp32x3 conv wgrad 256x256 @36^2 x20', 2.0 * 20 * 36 * 36 * 256 * 256 * 9, 20 * 36 * 36 * 512 * 4),
    'rows_linear_split_kernel': ('fp32x3 rows linear 32->32, 3.2 M rows', 2.0 * 3.2e6 * 32 * 32, 3.2e6 * 64 * 4),
    'rows_wgrad_split_kernel': ('fp32x3 rows wgrad 32x32, 3.2 M rows', 2.0 * 3.2e6 * 32 * 33, 3.2e6 * 64 * 4),
    'absmax256_kernel': ('absmax of 3.2 M x 32 f32', 3.2e6 * 32, 3.2e6 * 32 * 4),
    'pfn_block_split_fwd_kernel': ('fp32x3 pillar-encoder block forward, 3.2 M two-piece rows', 2.0 * 3.2e6 * (2 * 32 * 64 + 32 * 32), 3.2e6 * (128 + 128 + 128 + 128 + 12 + 4)),
    'pfn_block_split_dgrad_kernel': ('fp32x3 pillar-encoder block data gradient, 3.2 M rows', 2.0 * 3.2e6 * (2 * 32 * 64 + 32 * 32), 3.2e6 * (128 + 12 + 256 + 128)),
    'rows_linear_split_fm_kernel': ('fp32x3 rows linear 128->128, 430 k rows (weights in registers)', 2.0 * 429567 * 128 * 128, 429567 * 256 * 4),
    'head_conv_fwd_kernel': ('fg/bg head conv 32->2 forward @288^2 x20 (bf16 in)', 2.0 * 20 * 288 * 288 * 32 * 2 * 9, 20 * 288 * 288 * (32 * 2 + 2 * 4)),
    'head_conv_dgrad_kernel': ('fg/bg head conv data gradient (bf16 out)', 2.0 * 20 * 288 * 288 * 32 * 2 * 9, 20 * 288 * 288 * (32 * 2 + 2 * 4)),
    # round 4
    'pool_skip_relu_bwd_y32_kernel': ('pool tail backward, fp32 winners + bf16 gradients, 32 ch @288^2 x20', 0.0, 20 * 288 * 288 * 32 * (4 + 2 + 2) + 20 * 144 * 144 * 32 * 2),
    'vox_batch_keys': ('batched collate + first-touch keys, 4 x 800 k points', 0.0, 3.2e6 * (56 + 24 + 16 + 24 + 4)),
    'vox_batch_assign': ('batched voxeliser: ranks + float64 coordinates rows, 3.2 M points', 0.0, 3.2e6 * 8 + 1.17e6 * (40 + 4)),
    'vox_batch_p2v': ('point -> collated pillar id, 3.2 M points', 0.0, 3.2e6 * 12),
    'vox_count': ('first-touch flags per chunk, 3.2 M points', 0.0, 3.2e6 * 8),
    # round 4, second half (verdict item 7 kernels; late round-4 kernels)
    'bilinear_gather_bwd_sorted_kernel': ('bilinear gather backward (sorted), 320 k points x 64 ch into 4 x 288^2 (bf16 out)', 0.0, 320000 * (64 * 4 + 16) + 4 * 288 * 288 * (64 * 2 + 4)),
    'bilinear_sorted_prep_kernel': ('its preparation pass: gradient rows into CSR order + tap weights', 0.0, 320000 * (64 * 8 + 12 + 4 + 16)),
    'ego_sinkhorn_rows_ro_kernel': ('Sinkhorn forward row half-step, 16 x 1024^2 (matrix read once, one vector written)', 0.0, 16 * 1024 * 1024 * 4),
    'ego_sinkhorn_cols_ro_kernel': ('Sinkhorn forward column half-step, 16 x 1024^2', 0.0, 16 * 1024 * 1024 * 4),
    'ego_sinkhorn_finish_ro_kernel': ('Sinkhorn forward: the normalised matrix written', 0.0, 16 * 1024 * 1024 * 8),
    'sk_cols_kernel': ('Sinkhorn backward column pass, 16 x 1024^2', 0.0, 16 * 1024 * 1024 * 8),
    'sk_rows_kernel': ('Sinkhorn backward row pass, 16 x 1024^2', 0.0, 16 * 1024 * 1024 * 8),
    'sk_final_kernel': ('Sinkhorn backward: gradient matrix written', 0.0, 16 * 1024 * 1024 * 12),
    'seg_rows_kernel': ('fg/bg loss: cross entropy + error keys, 600 k rows', 0.0, 600000 * (8 + 8 + 8)),
    'seg_lovasz_kernel': ('fg/bg loss: Lovasz sums over the sorted errors, 600 k x 2 classes', 0.0, 600000 * 2 * (8 + 4)),
    'upconv_bf16_kernel<4, 0>': ('bf16 transposed conv forward 64->32 @144^2 x20', 2.0 * 20 * 144 * 144 * 64 * 128, 20 * 144 * 144 * (64 + 128) * 2),
    'upconv_bf16_kernel<2, 1>': ('bf16 transposed conv data gradient 64->32 (dy = channel slice)', 2.0 * 20 * 144 * 144 * 64 * 128, 20 * 144 * 144 * (64 + 128) * 2),
    'upconv_bf16_wgrad_kernel': ('bf16 transposed conv weight gradient 64->32', 2.0 * 20 * 144 * 144 * 64 * 128, 20 * 144 * 144 * (64 + 128) * 2 + 1024 * 8192 * 4),
    'upconv_bf16_wgrad_reduce_kernel': ('its partial-slot reduce (1024 slots x 8192)', 0.0, 1024 * 8192 * 4),
    'rows_linear_fewk_kernel<9, 8>': ('position layer 9->64, 3.2 M rows: fp32 + bf16 outputs + maxima', 2.0 * 3.2e6 * 9 * 64, 3.2e6 * (36 + 256 + 128)),
    'conv3x3_split_res_kernel<32, 1, 2, 5, true': ('fp32x3 conv on two inputs cat(32, 32)->32 @288^2 x20 (+ bf16 second output)', 2.0 * 20 * 288 * 288 * 64 * 32 * 9, 20 * 288 * 288 * (64 * 4 + 32 * 4 + 32 * 2)),
    'head_conv_wgrad_kernel': ('fg/bg head conv weight gradient (bf16 in)', 2.0 * 20 * 288 * 288 * 32 * 2 * 9, 20 * 288 * 288 * (32 * 2 + 2 * 4)),
    # round 6
    'seg_max_canvas_kernel': ('pooling + pillar scatter in one pass: 3.2 M fp32 rows -> fp32 canvas + bf16 shadow + winners (20 x 288^2 cells, 1.17 M pillars)', 0.0,
                              4 * 32 * 3_200_000 + 4 * 3_200_000 + 4 * 1_169_434 + 4 * 1_658_880 + 6 * 32 * 1_658_880 + 4 * 32 * 1_169_433),
    'seg_max_kernel<8>': ('per-pillar max of 3.2 M fp32 rows x 32 ch (1.17 M pillars): values + bf16 copy + winners', 0.0,
                          4 * 32 * 3_200_000 + 4 * 3_200_000 + 4 * 1_169_434 + (4 + 2 + 4) * 32 * 1_169_433),
    'scatter_sum_small_kernel': ('few-row sums in 64-bit fixed point: 320 k rows x 16 ch into 400 rows', 0.0, 320_000 * (64 + 4)),
    'csr_fill_small': ('stable counting-sort fill, 320 k points into 400 segments', 0.0, 320_000 * 8),
    'csr_sort_long': ('workgroup sort of the > 64-point segments (3.2 M points / 1.17 M pillars: a handful; no algorithmic byte count -- the ratio column is meaningless for it)', 0.0, 0.0),
}


# [r6] which input distribution each row was measured on: the dense kernels' time does not depend on the values; the irregular ones do
def distribution(frag):
    if 'seg_max_canvas_kernel' in frag or frag == 'seg_max_kernel<8>':
        return 'pillar sizes: 3.2 M points in 1.17 M pillars, every pillar >= 1 point, the rest uniform over the pillars (the bench\'s uniform synthetic points); ' \
               'the LiDAR-shaped row (1 / r weights, a third of the cells occupied) is tools/bench_fused_canvas.py, profiles/r06_fused_canvas_variants.txt'
    if frag.startswith('vox_'):
        return 'the step\'s own points: 4 sequences of synthetic.make_sequence (uniform over the crop box)'
    if 'bilinear' in frag:
        return 'points uniform over the map (the clustered foreground case -- 20 boxes per sequence -- is tools/bench_bilinear_ab.py, profiles/r05_bilinear_ab.txt)'
    if frag.startswith(('seg_rows', 'seg_lovasz')):
        return 'normal logits, uniform random labels'
    if frag.startswith(('ego_', 'sk_')):
        return 'normal log-affinities'
    if frag.startswith(('scatter_sum_small', 'csr_')):
        return 'slot / segment numbers uniform random (320 k points over 400 segments; csr_sort_long: the > 64-point segments of 3.2 M uniform points over 1.17 M pillars)'
    if 'pfn_block' in frag:
        return 'normal rows, point -> pillar map uniform random (gather side); the step\'s map is sorted by cell within a frame'
    return 'dense tensor of normal values: no dependence on the data distribution (every element is touched once per tap)'


def main():
    root, out = sys.argv[1], sys.argv[2]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, '*', '*counter_collection.csv')) + glob.glob(os.path.join(root, '*', '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            for frag in sorted(KERNELS, key=len, reverse=True):
                if frag in r['Kernel_Name']:
                    vals[frag][r['Counter_Name']].append(float(r['Counter_Value']))
                    durs[frag].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
                    break
    res = {'source': 'rocprofv3 --pmc (one counter set per run, --kernel-trace) on tools/pmc_kernels.py; durations are those of the counter runs '
                     '(clocks sit lower under counter collection than in a plain run)'}
    for frag, (label, flops, hbm) in KERNELS.items():
        if frag not in vals:
            continue
        c = {k: sum(v) / len(v) for k, v in vals[frag].items()}
        us = sum(durs[frag]) / len(durs[frag])
        row = {'what': label, 'distribution': distribution(frag), 'avg_us_under_pmc': round(us, 1), 'TFLOPs': round(flops / us / 1e6, 1), 'algorithmic_GBps': round(hbm / us / 1e3, 1),
               'counters': {k: round(v, 1) for k, v in sorted(c.items())}}
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
            row["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
            row['lds_conflict_share'] = round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4)
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            row['hbm_bytes'] = round((2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024)
            if hbm > 0:
                row['hbm_over_algorithmic'] = round(row['hbm_bytes'] / hbm, 3)
        res[frag] = row
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
