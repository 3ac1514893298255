"""Golden vectors for the host data step: the reference's BaseDataset.prep_input (libs/dataset.py:147-204) run on a synthetic
raw sample with numpy's global generator seeded.  The dataset object is created without its __init__ (which needs the
dataset on disk); only the attributes prep_input reads are set.  Run: python tests/golden/make_golden_prep.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
from helpers import raw_sample  # noqa: E402


def gen_prep(save):
    ref_harness.install()
    from libs.dataset import BaseDataset
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
    vg, da, dd = cfg['voxel_generator'], cfg['data_aug'], cfg['data']
    out = {}
    for tag, aug, seed in (('aug', True, 11), ('plain', False, 12)):
        ds = object.__new__(BaseDataset)
        ds.augmentation = aug
        ds.augment_noise, ds.augment_shift_range = da['augment_noise'], da['augment_shift_range']
        ds.augment_scale_min, ds.augment_scale_max, ds.rot_aug = da['augment_scale_min'], da['augment_scale_max'], da['rot_aug']
        ds.voxeliser = ref_harness.voxeliser(cfg)
        ds.n_frames = 3
        ds.crop_xy, ds.crop_z_min, ds.crop_z_max = vg['crop_range']
        ds.remove_ground = dd['remove_ground']
        ds.ground_height = dd['ground_height'] + dd['ground_slack']
        raw = raw_sample(seed, 3, 1500, cfg)
        np.random.seed(1000 + seed)
        d = ds.prep_input(raw['raw_points'].copy(), raw['sd_labels'], raw['fb_labels'], raw['inst_labels'], raw['time_indice'],
                          raw['ego_motion_gt'].copy(), raw['inst_motion_gt'].copy())
        for k in ('input_points', 'num_points', 'time_indice', 'sd_labels', 'inst_labels', 'fb_labels', 'ego_motion_gt', 'inst_motion_gt',
                  'coordinates', 'point_to_voxel_map', 'num_voxels'):
            out['%s_%s' % (tag, k)] = np.asarray(d[k])
        out['%s_seed' % tag] = 1000 + seed
        out['%s_sample_seed' % tag] = seed
        print(tag, 'kept', int(d['num_points'][0]), 'of', raw['raw_points'].shape[0], 'voxels', int(np.asarray(d['num_voxels']).reshape(-1)[0]))
    save('prep_input', **out)


if __name__ == '__main__':
    def save(name, **arrays):
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **arrays)
    gen_prep(save)
