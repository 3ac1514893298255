"""Config dictionary for the hot path.

The reference builds its cfg dict by merging configs/default.yaml with a dataset
yaml (toolbox/config.py:119-138) and then copying the voxel-generator keys into
cfg['pillar_encoder'] (main.py:10-14).  Only the keys MotionNet reads are kept
here ("Config / flags" row); values are the reference's
(configs/default.yaml:51-117, configs/waymo/waymo.yaml, configs/nuscene/nuscene.yaml).
"""
import copy

_COMMON = {
    'misc': {'mode': 'val', 'use_gpu': True, 'seed': 42},
    'data': {'max_speed': 20, 'speed_threshold': 0.5, 'ground_slack': 0.3, 'remove_ground': True},
    'data_aug': {'augment_noise': 0.01, 'augment_shift_range': 0.25, 'augment_scale_min': 0.995, 'augment_scale_max': 1.005,
                 'rot_aug': 0.5},                                       # configs/default.yaml:44-49
    'cluster': {'cluster_metric': 'euclidean', 'min_p_cluster': 15, 'min_samples_dbscan': 5,
                'eps_dbscan': 0.4, 'voxel_size': 0.15},
    'pillar_encoder': {'depth': 3, 'num_input_features': 9, 'num_filters': 32},
    'unet': {'start_filts': 32, 'in_channels': 32, 'depth': 5, 'merge_mode': 'concat'},
    'pose_estimation': {'n_kpts': 1024, 'add_slack': True, 'sinkhorn_iter': 3, 'feats_dim': 64,
                        'icp_threshold': 0.15, 'icp_max_iter': 50, 'seq_pose': 'skip'},
    'stpn': {'feat_dim': 32},
    'tpointnet': {'n_iterations': 1, 'min_points': 10, 'icp_threshold': 0.25},
    'loss': {'w_pose_l1_loss': 1.0, 'w_perm_loss': 0.005, 'w_mos_bce_loss': 1.0,
             'w_mos_lovasz_loss': 1.0, 'w_fb_bce_loss': 1.0, 'w_fb_lovasz_loss': 1.0,
             'w_offset_norm_loss': 0.5, 'w_offset_dir_loss': 0.5, 'w_obj_l1_loss': 1.0,
             'w_obj_pose_loss': 1.0, 'w_obj_loss': 0.3, 'w_obj_rot_loss': 50,
             'w_obj_trans_loss': 1.0, 'obj_gamma': 0.7},
    'model': {'ego_icp': False, 'tpointnet_icp': False},
    'train': {'iter_size': 2, 'batch_size': 4, 'grad_clip': 1.0},
    'Adam': {'learning_rate': 0.0005, 'weight_decay': 0.0},
}

_DATASETS = {
    'waymo': {
        'voxel_generator': {'range': [-36, -36, -2, 36, 36, 6], 'voxel_size': [0.25, 0.25, 8],
                            'n_sweeps': 5, 'crop_range': [32, -2, 6]},
        'data': {'dataset': 'waymo', 'n_frames': 5, 'freq': 10.0, 'ground_height': 0.04, 'max_speed': 30},
        'pose_estimation': {'icp_threshold': 0.1},
        'tpointnet': {'n_iterations': 2, 'min_points': 50, 'icp_threshold': 0.15},
    },
    'nuscene': {
        'voxel_generator': {'range': [-36, -36, -5, 36, 36, 3], 'voxel_size': [0.25, 0.25, 8],
                            'n_sweeps': 11, 'crop_range': [32, -5, 3]},
        'data': {'dataset': 'nuscene', 'n_frames': 11, 'freq': 20.0, 'ground_height': -1.84,
                 'max_speed': 10},
        'pose_estimation': {'icp_threshold': 0.2},
        'tpointnet': {'n_iterations': 2, 'min_points': 50, 'icp_threshold': 0.25},
    },
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict):
            _merge(dst.setdefault(k, {}), v)
        else:
            dst[k] = v


def update_config(config):
    """main.py:10-14: the pillar encoder reads the voxel-generator geometry."""
    config['pillar_encoder']['voxel_size'] = config['voxel_generator']['voxel_size']
    config['pillar_encoder']['pc_range'] = config['voxel_generator']['range']
    config['pillar_encoder']['n_sweeps'] = config['voxel_generator']['n_sweeps']
    return config


def default_config(dataset='waymo', mode='val', n_sweeps=None, xy_range=None):
    """Merged cfg dict as main.py would hand it to MotionNet(cfg).

    n_sweeps / xy_range override the sequence length and the half-extent of the BEV
    square (cell size stays 0.25 m) so that BASELINE.json's T=5/T=10 workloads and
    the small parity grids use the same code path.
    """
    cfg = copy.deepcopy(_COMMON)
    _merge(cfg, copy.deepcopy(_DATASETS[dataset]))
    cfg['misc']['mode'] = mode
    if n_sweeps is not None:
        cfg['voxel_generator']['n_sweeps'] = int(n_sweeps)
        cfg['data']['n_frames'] = int(n_sweeps)
    if xy_range is not None:
        r = cfg['voxel_generator']['range']
        r[0] = r[1] = -xy_range
        r[3] = r[4] = xy_range
        cfg['voxel_generator']['crop_range'][0] = xy_range - 4 if xy_range > 8 else xy_range - 1
    return update_config(cfg)
