"""Golden vector for the test-mode motion-segmentation metric: the reference MotionNet in misc.mode = 'test' on the tiny scene of
model_tiny_test.npz, then FuseLoss.get_mos_loss as SegTrainer.test calls it (libs/tester.py:87): supervised on the points that are
foreground in the ground truth OR in the estimate (libs/loss.py:145-147), although the test-mode forward itself decodes the estimated
foreground only.  Run: python tests/golden/make_golden_test_loss.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness as rh  # noqa: E402

rh.install()
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.synthetic import make_sequence, attach_voxels, fill_state_dict_  # noqa: E402


def main():
    from models.motionnet import MotionNet
    import libs.loss as L
    g = np.load(os.path.join(HERE, 'model_tiny_test.npz'))
    cfg = default_config('waymo', 'val', n_sweeps=3, xy_range=8)
    cfg['misc'] = dict(cfg['misc'], mode='test')
    vox = rh.voxeliser(cfg)
    inp = rh.collate([attach_voxels(make_sequence(int(g['seeds'][0]), 3, int(g['pts_per_frame']), cfg), vox)])
    model = MotionNet(cfg)
    fill_state_dict_(model)
    with torch.no_grad():
        sd = model.state_dict()
        for k, v in zip(g['tweak_keys'], g['tweak_vals']):
            sd[str(k)] += torch.from_numpy(v)
        sd['motionhead.offset_head.seg_head.3.weight'] *= float(g['offset_scale'])
        sd['motionhead.offset_head.seg_head.3.bias'] *= float(g['offset_scale'])
        sd['motionhead.mos_seg.seg_head.3.bias'] += torch.tensor([0.0, float(g['mos_shift'])])
    model.eval()
    torch.manual_seed(int(g['fwd_seed']))
    with torch.no_grad():
        out = model(inp)
    L.ClusterEvaluation = lambda c: None
    loss_fn = L.FuseLoss(dict(cfg['loss'], save_dir='/tmp', min_p_cluster=15))
    with torch.no_grad():
        mos = loss_fn.get_mos_loss(out, inp)
    fb_gt, fb_est = inp['fb_labels'][:, 0] == 1, out['fb_est_per_points'][:, 0] == 1
    print('supervised points: union %d, estimate only %d' % (int((fb_gt | fb_est).sum()), int(fb_est.sum())), mos['metric'])
    np.savez_compressed(os.path.join(HERE, 'model_tiny_test_loss.npz'), n_union=int((fb_gt | fb_est).sum()), n_est=int(fb_est.sum()),
                        bce_loss=float(mos['bce_loss']), lovasz_loss=float(mos['lovasz_loss']),
                        **{'mos_' + k: v for k, v in mos['metric'].items()})


if __name__ == '__main__':
    main()
