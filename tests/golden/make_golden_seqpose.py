"""Golden vectors for pose_estimation.seq_pose = 'chain' / 'full' (models/egomotion.py:195-307): the reference MotionNet on the tiny
validation scene of model_tiny_val.npz (same seeds, weights and head-bias offsets), eval mode.
Run: python tests/golden/make_golden_seqpose.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness as rh  # noqa: E402

rh.install()
from make_golden_model import _run, _common  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402


def main():
    g = np.load(os.path.join(HERE, 'model_tiny_val.npz'))
    tweaks = {str(k): v for k, v in zip(g['tweak_keys'], g['tweak_vals'])}
    out_arrays = {}
    for mode in ('chain', 'full'):
        cfg = default_config('waymo', 'val', n_sweeps=3, xy_range=8)
        cfg['pose_estimation']['seq_pose'] = mode
        model, inp, out, stats, _ = _run(cfg, [int(s) for s in g['seeds']], 3, int(g['pts_per_frame']), 'val', int(g['fwd_seed']),
                                         train=False, tweaks=tweaks)
        d, epe = _common(out, stats, inp, 3)
        out_arrays.update({mode + '_ego_motion_est': out['ego_motion_est'].numpy(), mode + '_ego_motion_gt': out['ego_motion_gt'].numpy(),
                           mode + '_ego_rot_error': d['ego_rot_error'], mode + '_ego_trans_error': d['ego_trans_error'],
                           mode + '_ego_l1_loss': d['ego_l1_loss'], mode + '_ego_l2_loss': d['ego_l2_loss'], mode + '_perm_loss': d['perm_loss'],
                           mode + '_n_perm': len(out['perm_matrix']), mode + '_epe_mean': d['epe_mean'], mode + '_mos_iou': d['mos_iou'],
                           mode + '_perm_rowsum': np.stack([p.sum(2)[0].numpy() for p in out['perm_matrix']])})
        print(mode, d['ego_rot_error'], d['ego_trans_error'], d['ego_l1_loss'], len(out['perm_matrix']))
    np.savez_compressed(os.path.join(HERE, 'seqpose.npz'), **out_arrays)


if __name__ == '__main__':
    main()
