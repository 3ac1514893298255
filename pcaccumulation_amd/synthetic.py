"""Synthetic LiDAR sequences in the reference's sample format (no dataset is available offline).

One sample has the keys BaseDataset.prep_input emits before voxelisation
(libs/dataset.py:186-196): input_points [N,3] f64, num_points [1] i64,
time_indice [N,1] i64, sd_labels / inst_labels / fb_labels [N,1] i64,
ego_motion_gt [T,4,4] f64, inst_motion_gt [K,T,4,4] f64.  Points are generated inside
the crop (|x|,|y| < crop, z above the ground slack, libs/dataset.py:170-183) so every
point lands in a pillar, which is the invariant the reference enforces by redrawing
samples (libs/dataset.py:218-219).  numpy's legacy RandomState is used so the same seed
gives the same bytes on every box.
"""
import numpy as np


def _se2(yaw, tx, ty):
    m = np.eye(4)
    c, s = np.cos(yaw), np.sin(yaw)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    m[0, 3], m[1, 3] = tx, ty
    return m


def make_sequence(seed, n_frames, pts_per_frame, cfg, mode='uniform', n_inst=20, fg_ratio=0.1):
    """One synthetic sequence.

    mode 'uniform': x,y ~ U(-c, c) with c just inside the crop, z ~ U(z_min+0.4, z_max-0.1).
    mode 'lidar'  : range density ~ 1/r on [2, 45] m, 64 beam elevations, then the same crop
                    (about 3 points per pillar as in models/motionnet.py:142).
    Foreground: fg_ratio of the points sit in n_inst boxes (4 x 2 x 1.5 m); every second
    instance moves 0-1 m per frame and inst_motion_gt is consistent with that motion.
    Ego motion: per-frame SE(2) step (yaw ~ U(-1,1) deg, forward 0.5-1.5 m); ego_motion_gt[t]
    maps frame t into the anchor frame 0 (anchor = identity).
    """
    rng = np.random.RandomState(seed)
    vg = cfg['voxel_generator']
    x_half = float(vg['range'][3])
    crop = x_half - 4.1 if x_half > 8 else x_half - 0.1
    z_lo, z_hi = float(vg['range'][2]) + 0.4, float(vg['range'][5]) - 0.1
    z_lo = max(z_lo, float(cfg['data']['ground_height']) + float(cfg['data']['ground_slack']) + 0.05)
    T = int(n_frames)

    # ego trajectory
    ego = [np.eye(4)]
    for _ in range(1, T):
        step = _se2(np.deg2rad(rng.uniform(-1, 1)), rng.uniform(0.5, 1.5), rng.uniform(-0.05, 0.05))
        ego.append(ego[-1] @ step)
    ego_motion_gt = np.stack(ego).astype(np.float64)

    # instances: label 0 is the static background, 1..n_inst are boxes
    K = n_inst + 1
    centres = np.stack([rng.uniform(-crop + 4, crop - 4, K), rng.uniform(-crop + 4, crop - 4, K),
                        rng.uniform(z_lo + 0.8, min(z_lo + 2.0, z_hi - 0.8), K)], axis=1)
    yaw = rng.uniform(0, 2 * np.pi, K)
    speed = np.where(np.arange(K) % 2 == 1, rng.uniform(0.2, 1.0, K), 0.0)   # odd labels move
    speed[0] = 0.0
    inst_motion_gt = np.tile(np.eye(4)[None, None], (K, T, 1, 1))
    for k in range(1, K):
        d = speed[k] * np.array([np.cos(yaw[k]), np.sin(yaw[k])])
        for t in range(T):
            inst_motion_gt[k, t, 0, 3] = -d[0] * t
            inst_motion_gt[k, t, 1, 3] = -d[1] * t

    pts, tix, inst = [], [], []
    n_fg = int(pts_per_frame * fg_ratio) if n_inst > 0 else 0
    n_bg = pts_per_frame - n_fg
    for t in range(T):
        if mode == 'uniform':
            bg = np.stack([rng.uniform(-crop, crop, n_bg), rng.uniform(-crop, crop, n_bg),
                           rng.uniform(z_lo, z_hi, n_bg)], axis=1)
        elif mode in ('lidar', 'lidar_scan'):
            r = 2.0 * (45.0 / 2.0) ** rng.uniform(0, 1, n_bg)          # density ~ 1/r
            az = rng.uniform(0, 2 * np.pi, n_bg)
            beam = rng.randint(0, 64, n_bg)
            elev = np.deg2rad(np.linspace(-17.6, 2.4, 64))[beam]
            if mode == 'lidar_scan':                                   # the order a spinning sensor delivers: beam by beam, by azimuth
                order = np.lexsort((az, beam))
                r, az, elev = r[order], az[order], elev[order]
            bg = np.stack([r * np.cos(az), r * np.sin(az), 1.8 + r * np.tan(elev)], axis=1)
            bg[:, 0] = np.clip(bg[:, 0], -crop, crop)
            bg[:, 1] = np.clip(bg[:, 1], -crop, crop)
            bg[:, 2] = np.clip(bg[:, 2], z_lo, z_hi)
        else:
            raise ValueError(mode)
        pts.append(bg)
        inst.append(np.zeros(n_bg, np.int64))
        if n_fg:
            lab = rng.randint(1, K, n_fg)
            local = np.stack([rng.uniform(-2, 2, n_fg), rng.uniform(-1, 1, n_fg),
                              rng.uniform(-0.75, 0.75, n_fg)], axis=1)
            c, s = np.cos(yaw[lab]), np.sin(yaw[lab])
            world = np.stack([c * local[:, 0] - s * local[:, 1], s * local[:, 0] + c * local[:, 1],
                              local[:, 2]], axis=1) + centres[lab]
            world[:, 0] += speed[lab] * np.cos(yaw[lab]) * t
            world[:, 1] += speed[lab] * np.sin(yaw[lab]) * t
            # express in the sensor frame of sweep t (inverse ego motion)
            inv = np.linalg.inv(ego_motion_gt[t])
            world = world @ inv[:3, :3].T + inv[:3, 3]
            world[:, 0] = np.clip(world[:, 0], -crop, crop)
            world[:, 1] = np.clip(world[:, 1], -crop, crop)
            world[:, 2] = np.clip(world[:, 2], z_lo, z_hi)
            pts.append(world)
            inst.append(lab.astype(np.int64))
        tix.append(np.full(pts_per_frame, t, np.int64))
    points = np.concatenate(pts).astype(np.float32).astype(np.float64)   # exactly representable in f32
    inst = np.concatenate(inst)
    time_indice = np.concatenate(tix)
    fb = (inst > 0).astype(np.int64)
    sd = ((inst % 2 == 1) & (inst > 0)).astype(np.int64)
    return {
        'input_points': points,
        'num_points': np.array([points.shape[0]], dtype=np.int64),
        'time_indice': time_indice[:, None],
        'sd_labels': sd[:, None],
        'inst_labels': inst[:, None],
        'fb_labels': fb[:, None],
        'ego_motion_gt': ego_motion_gt,
        'inst_motion_gt': inst_motion_gt.astype(np.float64),
    }


def attach_voxels(sample, voxeliser):
    """prep_input step 4 (libs/dataset.py:183-199): points [N,4] f32 = (x,y,z,t) -> pillar dict."""
    pts4 = np.concatenate((sample['input_points'], sample['time_indice']), axis=1).astype(np.float32)
    out = dict(sample)
    out.update(voxeliser(pts4))
    return out


def fill_state_dict_(module, gain=1.0):
    """Deterministic, RNG-free weights: every tensor is a closed-form hash of (key, flat index).

    Used on both sides of every parity test (do not rely on nn.init call order).
    Weights get a He-uniform amplitude sqrt(6/fan_in), biases are small, BatchNorm statistics stay near (0, 1).
    """
    import zlib
    import torch
    sd = module.state_dict()
    for key, t in sd.items():
        if key.endswith('num_batches_tracked'):
            t.zero_()
            continue
        n = t.numel()
        idx = np.arange(n, dtype=np.uint64)
        h = (idx + np.uint64(zlib.crc32(key.encode()))) * np.uint64(0x9E3779B1) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(15)
        h = h * np.uint64(0x85EBCA77) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(13)
        h = h * np.uint64(0xC2B2AE3D) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(16)
        u = (h.astype(np.float64) / 4294967296.0) * 2.0 - 1.0                 # U(-1, 1)
        leaf = key.rsplit('.', 1)[-1]
        if leaf == 'running_var':
            v = 1.0 + 0.1 * u
        elif leaf == 'running_mean':
            v = 0.02 * u
        elif leaf == 'weight' and t.dim() == 1:                                # BatchNorm scale
            v = 1.0 + 0.1 * u
        elif leaf == 'weight':
            fan_in = t[0].numel() if t.dim() > 1 else 1
            if 'upconv' in key:                                                # ConvTranspose2d (Cin,Cout,2,2)
                fan_in = t.shape[0]
            v = gain * np.sqrt(6.0 / max(fan_in, 1)) * u
        elif leaf == 'bias':
            v = 0.05 * u
        else:                                                                  # alpha / beta scalars etc.
            v = t.detach().cpu().numpy().astype(np.float64).reshape(-1)
        t.copy_(torch.from_numpy(np.asarray(v, np.float64).reshape(t.shape)).to(t.dtype))
    module.load_state_dict(sd)
    return module
