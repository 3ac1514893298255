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


# ---------------------------------------------------------------------------------------------------------------------
# Aggregation over a results folder: host mirror of toolbox/evaluation.py:20-98 (collect_results) with its helpers
# toolbox/sf_eval_utils.py:88-101 (collect_scene_stats), toolbox/metrics.py:5-41 (init / update_stats_meter) and
# toolbox/timer.py:23-42 (AverageMeter).  Reductions over two vectors per scene: torch ops on `device`, no kernel of its own.
# The three files it writes are the reference's (dynamic_dict.pth, scene_stats.pkl, static_stats.pkl); the pickled meters
# unpickle as toolbox.timer.AverageMeter inside the reference tree and as this class when only this package is importable.
# ---------------------------------------------------------------------------------------------------------------------
SAMPLE_FREQ = {'waymo': 4, 'nuscene': 1}                  # toolbox/evaluation.py:10-13


class AverageMeter(object):
    """toolbox/timer.py:23-42."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.avg, self.sum, self.sq_sum, self.count = 0, 0, 0.0, 0.0, 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        self.sq_sum += val ** 2 * n
        self.var = self.sq_sum / self.count - self.avg ** 2


def init_stats_meter(stats):
    """toolbox/metrics.py:29-41 (the hot path's metrics hold no arrays: dicts, [value, count] pairs and scalars)."""
    return {k: (init_stats_meter(v) if isinstance(v, dict) else AverageMeter()) for k, v in stats.items()}


def update_stats_meter(stats_meter, stats):
    """toolbox/metrics.py:5-27."""
    for key, value in stats.items():
        if key not in stats_meter:
            stats_meter[key] = init_stats_meter(value) if isinstance(value, dict) else AverageMeter()
    for key, value in stats.items():
        if isinstance(value, dict):
            update_stats_meter(stats_meter[key], value)
        elif isinstance(value, list):
            stats_meter[key].update(value[0], value[1])
        else:
            stats_meter[key].update(value)


def collect_scene_stats(epe_per_point, relative_error, sd_label, fb_label):
    """toolbox/sf_eval_utils.py:88-101 (its 'Static' entry is computed on the FOREGROUND mask, as written there)."""
    metrics = {'moving_ratio': sd_label.float().mean().item(), 'FG_ratio': fb_label.float().mean().item()}
    gt_mag = epe_per_point / (relative_error + 1e-20)
    if sd_label.sum():
        dyn = sd_label == 1
        metrics['Dynamic'] = compute_sf_metrics(epe_per_point[dyn], relative_error[dyn])
        metrics['Dynamic_motion_mag'] = gt_mag[dyn].mean().item()
    fg = fb_label == 1
    metrics['Static'] = compute_sf_metrics(epe_per_point[fg], relative_error[fg])
    return metrics


def scene_metrics(data, device):
    """One scene of collect_results (toolbox/evaluation.py:27-88) from the arrays of its flow_error.npz.
    -> (c_metrics, scene_stats, dynamic relative errors, dynamic end-point errors) -- the last two still un-sampled f16 tensors."""
    t = {k: torch.as_tensor(np.asarray(v)).to(device) for k, v in data.items()}
    fb_label, sd_label = t['fb_label'], t['sd_label']
    epe, rel = t['epe_per_point'].float(), t['relative_error'].float()
    if 'length' in t:                                       # run-length coded frame index (toolbox/evaluation.py:37-46)
        time_indice = torch.repeat_interleave(t['time_indice'].long(), t['length'].long())
        time_indice = torch.nn.functional.pad(time_indice, (0, fb_label.numel() - time_indice.numel()))
    else:
        time_indice = t['time_indice']
    dyn = sd_label == 1
    c = {'scene_overall': compute_sf_metrics(epe, rel)}
    sel = sd_label == 0
    c['static_overall'] = compute_sf_metrics(epe[sel], rel[sel])
    sel = torch.logical_and(sd_label == 0, fb_label == 0)
    c['static_BG'] = compute_sf_metrics(epe[sel], rel[sel])
    sel = torch.logical_and(sd_label == 0, fb_label == 1)
    if sel.sum():
        c['static_FG'] = compute_sf_metrics(epe[sel], rel[sel])
    n_frames = int(time_indice.max().item()) + 1
    for t_idx in range(1, n_frames):
        sel = torch.logical_and(sd_label == 0, time_indice == t_idx)
        c['%d-th frame' % t_idx] = compute_sf_metrics(epe[sel], rel[sel])
    return c, collect_scene_stats(epe, rel, sd_label, fb_label), rel[dyn].half(), epe[dyn].half()


def collect_results(target_folder, save_dir, dataset, device=None):
    """toolbox/evaluation.py:20-98: every <target_folder>/<scene>/flow_error.npz -> static_stats.pkl (running meters of the static
    splits: scene / static overall / static BG / static FG / per frame), scene_stats.pkl (per-scene dictionaries) and
    dynamic_dict.pth (every SAMPLE_FREQ-th error of the moving points, float16 values in float32 tensors).  Scenes are visited in
    sorted order (the reference takes glob's order; the meters do not depend on it, the sampled lists are concatenated in it)."""
    import pickle
    from glob import glob
    device = torch.device(device) if device is not None else torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    stats_meter, scene_stats, rel_list, epe_list = None, {}, [], []
    for path in sorted(glob(os.path.join(target_folder, '*', 'flow_error.npz'))):
        with np.load(path) as data:
            c_metrics, sstats, rel_dyn, epe_dyn = scene_metrics({k: data[k] for k in data.files}, device)
        if rel_dyn.numel():
            rel_list.append(rel_dyn[::SAMPLE_FREQ[dataset]].cpu())
            epe_list.append(epe_dyn[::SAMPLE_FREQ[dataset]].cpu())
        if stats_meter is None:
            stats_meter = init_stats_meter(c_metrics)
        update_stats_meter(stats_meter, c_metrics)
        scene_stats[path.split(os.sep)[-2]] = sstats
    cat = lambda parts: torch.cat(parts).float() if parts else torch.zeros(0)
    os.makedirs(save_dir, exist_ok=True)
    torch.save({'relative_error': cat(rel_list), 'epe_per_point': cat(epe_list)}, os.path.join(save_dir, 'dynamic_dict.pth'))
    for name, obj in (('scene_stats.pkl', scene_stats), ('static_stats.pkl', _as_reference_meters(stats_meter))):
        with open(os.path.join(save_dir, name), 'wb') as f:
            pickle.dump(obj, f)
    return stats_meter, scene_stats


def _as_reference_meters(meters):
    """Inside the reference tree (toolbox.timer importable) the meters are pickled as the reference's own AverageMeter, so that
    toolbox/evaluation.py's __main__ block (load_pkl + .avg) reads static_stats.pkl unchanged; elsewhere as this module's class."""
    try:
        from toolbox.timer import AverageMeter as RefMeter
    except ImportError:
        return meters

    def conv(m):
        if isinstance(m, dict):
            return {k: conv(v) for k, v in m.items()}
        r = RefMeter()
        r.__dict__.update(m.__dict__)
        return r
    return conv(meters) if meters is not None else None
