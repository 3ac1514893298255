"""The per-sample data step in front of the model, on the device: host mirror of BaseDataset.prep_input
(libs/dataset.py:147-204) -- augmentation, crop, ground removal, voxelisation -- for a raw sample that has been moved to HBM.

The reference runs this in numpy on DataLoader workers.  Here the point pass is one HIP kernel (include/pcacc.h D1), the kept
rows are compacted on the device and voxelised by the A1 kernel; the dict that comes back has the keys, shapes and dtypes
`collate_fn` expects (libs/dataloader.py:7-40), as device tensors.  Random numbers: `rng='reference'` draws them from numpy's
global generator in the reference's order (same seed -> same sample, used by the parity tests); `rng='device'` draws the
per-point noise with torch on the GPU (no 19 MB host array per 800 k points)."""
import numpy as np
import torch

from . import native
from .voxel_generator import Voxelization


def sample_random_tsfm(rot_aug, shift_range):
    """libs/dataset.py:106-116: rotation about z by U(0, pi * rot_aug), then an xy shift of U(-r, r) each."""
    angle = np.random.uniform(0, np.pi * rot_aug)
    from scipy.spatial.transform import Rotation                       # the reference's own call: same matrix bit for bit
    tsfm = np.eye(4)
    tsfm[:3, :3] = Rotation.from_euler('xyz', [0, 0, angle]).as_matrix()
    tsfm[0, 3] = np.random.uniform(-shift_range, shift_range)
    tsfm[1, 3] = np.random.uniform(-shift_range, shift_range)
    return tsfm


def conjugate_motions(aug_tsfm, ego_motion, inst_motion):
    """libs/dataset.py:118-139: poses of the augmented scene, T' T T'^-1 (numpy float64, tiny)."""
    inv = np.linalg.inv(aug_tsfm)
    ego = aug_tsfm[None] @ ego_motion @ inv[None]
    shape = inst_motion.shape
    inst = (aug_tsfm[None] @ inst_motion.reshape(-1, 4, 4) @ inv[None]).reshape(shape)
    return ego, inst


class PrepInput(object):
    def __init__(self, config, augmentation=True, rng='reference'):
        aug, vg, data = config['data_aug'], config['voxel_generator'], config['data']
        self.augmentation = augmentation
        self.augment_noise, self.augment_shift_range = aug['augment_noise'], aug['augment_shift_range']
        self.augment_scale_min, self.augment_scale_max, self.rot_aug = aug['augment_scale_min'], aug['augment_scale_max'], aug['rot_aug']
        self.crop_xy, self.crop_z_min, self.crop_z_max = vg['crop_range']
        self.remove_ground = data['remove_ground']
        self.ground_height = data['ground_height'] + data['ground_slack']
        self.voxeliser = Voxelization(vg)
        if rng not in ('reference', 'device'):
            raise ValueError("rng must be 'reference' or 'device'")
        self.rng = rng

    def point_pass(self, raw_points, ego_motion_gt, inst_motion_gt):
        """Augmentation + crop / ground flags.  raw_points [m,3] f64 on the device; the motions are numpy (host, tiny).
        Returns (points [m,3] f64, keep [m] u8, ego_motion_gt, inst_motion_gt)."""
        dev = raw_points.device
        tsfm12 = noise = None
        scale = 1.0
        if self.augmentation:
            tsfm = sample_random_tsfm(self.rot_aug, self.augment_shift_range)
            m = raw_points.shape[0]
            if self.rng == 'reference':
                noise = torch.from_numpy(np.random.rand(m, 3)).to(dev)
            else:
                noise = torch.rand((m, 3), dtype=torch.float64, device=dev)
            scale = float(np.random.uniform(self.augment_scale_min, self.augment_scale_max))
            tsfm12 = native.upload_small(np.concatenate([tsfm[:3, :3].reshape(-1), tsfm[:3, 3]]), torch.float64, dev)
            ego_motion_gt, inst_motion_gt = conjugate_motions(tsfm, np.asarray(ego_motion_gt), np.asarray(inst_motion_gt))
        pts, keep = native.prep_points(raw_points.contiguous(), tsfm12, noise, float(self.augment_noise), scale, float(self.crop_xy),
                                       float(self.crop_z_min), float(self.crop_z_max), self.remove_ground, float(self.ground_height))
        return pts, keep, ego_motion_gt, inst_motion_gt

    def __call__(self, raw_points, sd_labels, fb_labels, inst_labels, time_indice, ego_motion_gt, inst_motion_gt):
        """Arguments as BaseDataset.prep_input; the per-point arrays are device tensors ([m,3] f64, [m] labels / frame index)."""
        assert raw_points.shape[0] == sd_labels.shape[0] == fb_labels.shape[0] == inst_labels.shape[0] == time_indice.shape[0]
        pts, keep, ego, inst = self.point_pass(raw_points, ego_motion_gt, inst_motion_gt)
        idx = torch.nonzero(keep)[:, 0]                                            # the host sync of the sample (sizes)
        pts, t = pts[idx], time_indice[idx]
        points4 = torch.cat((pts, t[:, None].to(pts.dtype)), dim=1).float()
        coords, p2v, m = self.voxeliser.voxelize_device(points4)
        dev = pts.device
        return {
            'input_points': pts, 'num_points': torch.tensor([pts.shape[0]], dtype=torch.int64),
            'time_indice': t[:, None], 'sd_labels': sd_labels[idx][:, None], 'inst_labels': inst_labels[idx][:, None],
            'fb_labels': fb_labels[idx][:, None], 'ego_motion_gt': torch.as_tensor(np.asarray(ego)).to(dev),
            'inst_motion_gt': torch.as_tensor(np.asarray(inst)), 'coordinates': coords, 'num_voxels': torch.tensor([m], dtype=torch.int64),
            'shape': torch.from_numpy(np.hstack((self.voxeliser.grid_size, np.array([self.voxeliser.n_sweeps])))),
            'point_to_voxel_map': p2v[:, None],
        }
