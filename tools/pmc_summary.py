#!/usr/bin/env python
"""profiles/rNN_pmc_scatter_summary.json from the two rocprofv3 counter files of tools/pmc_scatter.py.
Usage: pmc_summary.py FETCH_SIZE_counter_collection.csv WRITE_SIZE_counter_collection.csv out.json
HBM bytes per launch = FETCH_SIZE x fetch_correction + WRITE_SIZE x write_correction (counter unit: KiB), both corrections taken
from the 128 MiB calibration copy in the same run, as MI355X_MICROARCH.md's HBM section prescribes."""
import csv
import json
import sys

csv.field_size_limit(1 << 30)
N_CELLS, M, C = 20 * 288 * 288, 1_169_433, 32
KERNELS = {                                  # kernel-name fragment -> (label, algorithmic bytes: SURVEY 8d  C*s*cells + C*s_in*M + 4*M)
    'pillar_scatter_vec4<0>': ('f32 rows -> f32 canvas', N_CELLS * C * 4 + M * C * 4 + 4 * M),
    'pillar_scatter_f32_nt': ('f32 rows -> f32 canvas, streaming loads / stores (round 5 default for large canvases)', N_CELLS * C * 4 + M * C * 4 + 4 * M),
    'pillar_scatter_vec4<1>': ('f32 rows -> bf16 canvas', N_CELLS * C * 2 + M * C * 4 + 4 * M),
    'pillar_scatter_rows16': ('bf16 rows -> bf16 canvas (bf16 compute mode)', N_CELLS * C * 2 + M * C * 2 + 4 * M),
    # [r6] pooling + scatter in one pass (bench.fused_alg_bytes): rows, order, offsets, cell table in; fp32 canvas, bf16 shadow, winners out
    'seg_max_canvas': ('3.2 M fp32 point rows -> fp32 canvas + bf16 shadow + winners (mixed mode: the encoder\'s last pooling writes the canvas)',
                       4 * C * 3_200_000 + 4 * 3_200_000 + 4 * (M + 1) + 4 * N_CELLS + 4 * C * N_CELLS + 2 * C * N_CELLS + 4 * C * M),
}


def collect(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name']
        key = None
        for frag in KERNELS:
            if frag in name:
                key = frag
        if key is None and 'copyBuffer' in name and float(r['Counter_Value']) > 32 * 1024:      # dst.copy_(src): the 128 MiB device copy
            key = 'copy'
        if key:
            out.setdefault(key, []).append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
    copy_kib = 128 * 1024.0
    fc, wc = copy_kib / fetch['copy'], copy_kib / write['copy']
    res = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/pmc_scatter.py: 4-sequence step size, %d cells, '
                     'M=%d, C=%d, pillar ids in cell order; Infinity Cache evicted between launches with a 512 MiB fill' % (N_CELLS, M, C),
           'units': 'counter values are KiB',
           'calibration_copy_128MiB': {'FETCH_SIZE': fetch['copy'], 'WRITE_SIZE': write['copy'],
                                       'fetch_correction': fc, 'write_correction': wc}}
    for frag, (label, alg) in KERNELS.items():
        if frag in fetch and frag in write:
            hbm = (fetch[frag] * fc + write[frag] * wc) * 1024.0
            res[frag] = {'what': label, 'FETCH_SIZE': fetch[frag], 'WRITE_SIZE': write[frag], 'hbm_bytes': hbm,
                         'algorithmic_bytes': alg, 'traffic_over_algorithmic': hbm / alg}
    json.dump(res, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
