"""Scene-flow evaluation of the test loop: host mirror of the per-scene error computation and result dump of
libs/tester.py:58-107 and of the metric definitions in toolbox/sf_eval_utils.py:51-86.

Everything stays on the device until the dump: per scene one fused pass builds the ground-truth accumulation, the end-point
error and the relative error of the points of frames t > 0; `FlowErrorDump` keeps them (as the reference does, in the narrow
dtypes of its .npz: int8 frame index, bool labels, float16 errors) and writes `flow_error.npz` with the reference's keys."""
import os

import numpy as np
import torch

from .tpointnet import ego_motion_compensation, reconstruct_sequence

_EPS = 1e-20                                   # libs/tester.py:16


def flow_errors(predictions, input_dict, n_frames):
    """libs/tester.py:58-83 for one scene (batch size 1).  Returns a dict of device tensors restricted to the points with t > 0:
    epe_per_point, relative_error (f32), time_indice (i64), fb_label, sd_label (bool)."""
    x = input_dict['input_points'].float()                # the tester's matmul with ego_motion_gt.float() needs f32 points
    t = input_dict['time_indice'][:, 1].long()
    ego = input_dict['ego_motion_gt'].float()[0]
    inst_motion = input_dict['inst_motion_gt'][0].to(x.device).float()
    comp = ego_motion_compensation(x, t, ego)
    rec_gt = reconstruct_sequence(comp, t, input_dict['inst_labels'][:, 0], inst_motion, n_frames)
    est_flow = predictions['rec_est'] - x
    gt_flow = rec_gt - x
    epe = torch.norm(est_flow - gt_flow, p=2, dim=1)
    rel = epe / (torch.norm(gt_flow, p=2, dim=1) + _EPS)
    sel = t > 0
    return {'epe_per_point': epe[sel], 'relative_error': rel[sel], 'time_indice': t[sel],
            'fb_label': input_dict['fb_labels'][:, 0][sel] != 0, 'sd_label': input_dict['sd_labels'][:, 0][sel] != 0}


def compute_sf_metrics(epe_per_point, relative_error):
    """toolbox/sf_eval_utils.py:71-86 (compute_sf_metrics_torch): EPE3D (mean, median), Acc3DS / Acc3DR, Outlier, ROutlier, each
    mean with the sample count.  One device->host transfer for all six numbers."""
    e, r = epe_per_point, relative_error
    vals = torch.stack([e.mean(), torch.median(e), ((e < 0.05) | (r < 0.05)).float().mean(), ((e < 0.1) | (r < 0.1)).float().mean(),
                        ((e > 0.3) | (r > 0.1)).float().mean(), ((e > 0.3) & (r > 0.3)).float().mean()]).tolist()
    size = e.size(0)
    return {'EPE3D': [vals[0], size], 'EPE3D_med': vals[1], 'Acc3DS': [vals[2], size], 'Acc3DR': [vals[3], size],
            'Outlier': [vals[4], size], 'ROutlier': [vals[5], size]}


class FlowErrorDump(object):
    """The accumulation lists of SegTrainer.test (libs/tester.py:45-49, 79-83) and its flow_error.npz (:95-107)."""

    KEYS = ('fb_label', 'sd_label', 'epe_per_point', 'relative_error', 'time_indice')

    def __init__(self):
        self.parts = {k: [] for k in self.KEYS}

    def add(self, errors):
        self.parts['time_indice'].append(errors['time_indice'].to(torch.int8))
        self.parts['fb_label'].append(errors['fb_label'])
        self.parts['sd_label'].append(errors['sd_label'])
        self.parts['relative_error'].append(errors['relative_error'].to(torch.float16))
        self.parts['epe_per_point'].append(errors['epe_per_point'].to(torch.float16))

    def arrays(self):
        out = {}
        for k, chunks in self.parts.items():
            out[k] = torch.cat(chunks).cpu().numpy() if chunks else np.zeros((0,))
        return out

    def save(self, save_dir):
        path = os.path.join(save_dir, 'flow_error')
        np.savez_compressed(path, **self.arrays())
        return path + '.npz'
