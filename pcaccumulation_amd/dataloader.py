"""Batch collation: the layout contract between the host data step and MotionNet.forward.

Mirrors libs/dataloader.py:7-40 (collate_fn): samples are concatenated; `coordinates` and
`time_indice` get a leading batch-index column (which makes them float64, because the column is built
with np.ones); `point_to_voxel_map` of sample b is offset by the number of pillars in samples < b.
"""
from collections import defaultdict

import numpy as np
import torch


def collate_fn(batch):
    merged = defaultdict(list)
    for example in batch:
        for k, v in example.items():
            merged[k].append(v)
    results = dict()
    for key, elems in merged.items():
        if key in ('coordinates', 'time_indice'):
            rows = [np.concatenate((np.ones((e.shape[0], 1)) * i, e), axis=1) for i, e in enumerate(elems)]
            results[key] = torch.tensor(np.concatenate(rows, axis=0))
        elif key in ('ego_motion_gt', 'shape'):
            results[key] = torch.tensor(np.stack(elems, axis=0))
        elif key == 'inst_motion_gt':
            results[key] = [torch.tensor(e) for e in elems]
        elif key == 'data_path':
            results[key] = elems
        else:
            results[key] = torch.tensor(np.concatenate(elems, axis=0))

    num_points, num_voxels = results['num_points'], results['num_voxels']
    start, n_pillars = 0, 0
    for b in range(num_voxels.size(0)):
        results['point_to_voxel_map'][start:start + num_points[b]] += n_pillars
        start += int(num_points[b])
        n_pillars += int(num_voxels[b])
    return results
