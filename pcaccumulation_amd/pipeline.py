"""Device-side data step in front of MotionNet: prep_input's voxelisation (libs/dataset.py:183-199) and
collate_fn (libs/dataloader.py:7-40) for samples that already live in HBM.

Produces exactly the input_dict layout the reference's collate_fn produces (same keys, shapes and dtypes,
see SURVEY.md 8b) so the model boundary does not change; only the voxeliser runs in the HIP kernel instead
of numba on a DataLoader worker.
"""
import torch

from . import native
from .voxel_generator import Voxelization


def sample_to_device(sample, device):
    """numpy sample dict (pcaccumulation_amd.synthetic.make_sequence) -> torch tensors on `device`."""
    # everything, including the per-instance motion table (the reference's collate keeps that one as a host list and every
    # consumer moves it with .to(device), models/alignnet.py:22, libs/loss.py:222: a blocking copy per sample per step)
    return {k: torch.from_numpy(v).to(device) for k, v in sample.items()}


class DeviceBatcher(object):
    def __init__(self, cfg):
        self.voxeliser = Voxelization(cfg['voxel_generator'])

    def __call__(self, samples):
        vox = self.voxeliser
        dev = samples[0]['input_points'].device
        coords, p2vs, tis, n_vox = [], [], [], []
        offset = 0
        launched = []
        for s in samples:                                                # launch every voxelisation, then ONE host sync
            pts4 = torch.cat((s['input_points'].float(), s['time_indice'].float()), dim=1)
            launched.append(vox.voxelize_launch(pts4))
        counts = torch.cat([l[2] for l in launched]).cpu().tolist()
        for b, s in enumerate(samples):
            t = s['time_indice']
            m = int(counts[b])
            c, p2v = launched[b][0][:m], launched[b][1]
            bcol = torch.full((m, 1), float(b), dtype=torch.float64, device=dev)
            coords.append(torch.cat((bcol, c.double()), dim=1))
            tis.append(torch.cat((torch.full((t.shape[0], 1), float(b), dtype=torch.float64, device=dev), t.double()), dim=1))
            p2vs.append((p2v + offset)[:, None])
            n_vox.append(m)
            offset += m
        cat = lambda key: torch.cat([s[key] for s in samples], dim=0)
        grid = torch.tensor(list(vox.grid_size) + [vox.n_sweeps], dtype=torch.int64)
        return {
            'input_points': cat('input_points'), 'num_points': cat('num_points'), 'time_indice': torch.cat(tis, 0),
            'sd_labels': cat('sd_labels'), 'inst_labels': cat('inst_labels'), 'fb_labels': cat('fb_labels'),
            'ego_motion_gt': torch.stack([s['ego_motion_gt'] for s in samples], 0),
            'inst_motion_gt': [s['inst_motion_gt'] for s in samples],
            'coordinates': torch.cat(coords, 0), 'num_voxels': native.upload_small(n_vox, torch.int64, dev),
            'shape': native.upload_small(grid[None].repeat(len(samples), 1), torch.int64, dev), 'point_to_voxel_map': torch.cat(p2vs, 0),
        }
