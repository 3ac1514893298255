"""Device-side data step (include/pcacc.h D1) against the reference's BaseDataset.prep_input.

Golden vectors: tests/golden/prep_input.npz (tests/golden/make_golden_prep.py: the reference's own prep_input, numpy generator
seeded).  CPU leg: the oracle restatement bit-exact, and the host mirror through the oracle-backed test double.
GPU leg: PrepInput on cuda:0 -- selection, labels, voxel coordinates and the point-to-voxel map exact; augmented points to
1e-12 (the 3x3 product is an FMA chain on the device, a BLAS call in numpy)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import raw_sample
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.dataset import PrepInput

KEYS = ('input_points', 'num_points', 'time_indice', 'sd_labels', 'inst_labels', 'fb_labels', 'ego_motion_gt', 'inst_motion_gt')


def _cfg():
    return default_config('waymo', 'train', n_sweeps=3, xy_range=8)


def _params(cfg):
    vg, dd = cfg['voxel_generator'], cfg['data']
    return dict(cfg['data_aug'], crop_xy=vg['crop_range'][0], crop_z_min=vg['crop_range'][1], crop_z_max=vg['crop_range'][2],
                remove_ground=dd['remove_ground'], ground_height=dd['ground_height'] + dd['ground_slack'], n_frames=3)


def test_oracle_prep_points_golden(golden):
    g, cfg = golden('prep_input'), _cfg()
    for tag, aug in (('aug', True), ('plain', False)):
        raw = raw_sample(int(g[tag + '_sample_seed']), 3, 1500, cfg)
        np.random.seed(int(g[tag + '_seed']))
        d = oracle.prep_points(raw['raw_points'], raw['sd_labels'], raw['fb_labels'], raw['inst_labels'], raw['time_indice'],
                               raw['ego_motion_gt'], raw['inst_motion_gt'], _params(cfg), aug)
        for k in KEYS:
            assert np.array_equal(d[k], g['%s_%s' % (tag, k)]), (tag, k)
        assert 0 < d['input_points'].shape[0] < raw['raw_points'].shape[0]


def _run(dev, g, tag, aug, rng='reference'):
    cfg = _cfg()
    raw = raw_sample(int(g[tag + '_sample_seed']), 3, 1500, cfg)
    np.random.seed(int(g[tag + '_seed']))
    t = lambda a: torch.from_numpy(a).to(dev)
    return PrepInput(cfg, augmentation=aug, rng=rng)(t(raw['raw_points']), t(raw['sd_labels']), t(raw['fb_labels']), t(raw['inst_labels']),
                                                     t(raw['time_indice']), raw['ego_motion_gt'], raw['inst_motion_gt'])


def _check(g, tag, d, atol):
    c = lambda v: v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    assert c(d['input_points']).shape == g[tag + '_input_points'].shape
    np.testing.assert_allclose(c(d['input_points']), g[tag + '_input_points'], rtol=0, atol=atol)
    for k in ('num_points', 'time_indice', 'sd_labels', 'inst_labels', 'fb_labels'):
        assert np.array_equal(c(d[k]), g['%s_%s' % (tag, k)]), (tag, k)
    np.testing.assert_allclose(c(d['ego_motion_gt']), g[tag + '_ego_motion_gt'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(c(d['inst_motion_gt']), g[tag + '_inst_motion_gt'], rtol=0, atol=1e-12)
    assert np.array_equal(c(d['coordinates']), g[tag + '_coordinates'])
    assert np.array_equal(c(d['point_to_voxel_map']), g[tag + '_point_to_voxel_map'])
    assert int(c(d['num_voxels'])[0]) == int(np.asarray(g[tag + '_num_voxels']).reshape(-1)[0])


def test_host_mirror_cpu(golden, monkeypatch):
    from oracle import cpu_backend
    cpu_backend.install(monkeypatch)
    g = golden('prep_input')
    for tag, aug in (('aug', True), ('plain', False)):
        _check(g, tag, _run(torch.device('cpu'), g, tag, aug), 1e-12)


@pytest.mark.gpu
def test_prep_input_gpu(golden):
    g = golden('prep_input')
    dev = torch.device('cuda:0')
    for tag, aug in (('aug', True), ('plain', False)):
        d = _run(dev, g, tag, aug)
        assert d['input_points'].is_cuda and d['input_points'].dtype == torch.float64 and d['coordinates'].is_cuda
        _check(g, tag, d, 0.0 if not aug else 1e-12)


@pytest.mark.gpu
def test_prep_input_device_rng_statistics():
    """rng='device': same selection rules and transform, noise drawn on the GPU: bounded by augment_noise / 2 per coordinate."""
    cfg = _cfg()
    cfg['data_aug']['augment_scale_min'] = cfg['data_aug']['augment_scale_max'] = 1.0     # the two modes draw the scale at different
    raw = raw_sample(5, 3, 1500, cfg)                                                     # positions of numpy's stream
    dev = torch.device('cuda:0')
    prep = PrepInput(cfg, augmentation=True, rng='device')
    np.random.seed(3)
    pts, keep, ego, inst = prep.point_pass(torch.from_numpy(raw['raw_points']).to(dev), raw['ego_motion_gt'], raw['inst_motion_gt'])
    np.random.seed(3)
    prep_ref = PrepInput(cfg, augmentation=True, rng='reference')
    pts_ref, keep_ref, _, _ = prep_ref.point_pass(torch.from_numpy(raw['raw_points']).to(dev), raw['ego_motion_gt'], raw['inst_motion_gt'])
    assert (pts - pts_ref).abs().max().item() <= cfg['data_aug']['augment_noise'] * 1.01       # both within +-noise/2 of the clean transform
    assert (keep != keep_ref).float().mean().item() < 0.01
