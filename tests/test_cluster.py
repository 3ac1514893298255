"""Test-mode instance clustering (include/pcacc.h C1).

Golden vectors: tests/golden/cluster.npz -- the reference's models/cluster.py:Cluster run on synthetic scenes with the real
scikit-learn DBSCAN (tests/golden/make_golden_cluster.py); model_tiny_test.npz -- the reference MotionNet in misc.mode='test'.
CPU leg: the oracle's restated DBSCAN / sparse_quantize against the fixture and against scikit-learn itself.
GPU leg: pcacc_cluster through the C ABI, bit-exact against the fixture and against scikit-learn on random scenes."""
import numpy as np
import pytest
import torch

import oracle
from pcaccumulation_amd.cluster import Cluster
from pcaccumulation_amd.config import default_config

CFG = {'cluster': {'min_p_cluster': 15, 'voxel_size': 0.15, 'min_samples_dbscan': 5, 'cluster_metric': 'euclidean', 'eps_dbscan': 0.4}}


def _sklearn():
    from sklearn.cluster import DBSCAN
    return DBSCAN(min_samples=5, metric='euclidean', eps=0.4)


def _scene(seed, n_blobs, n_noise, n_static, spread):
    rng = np.random.RandomState(seed)
    pts, mos = [], []
    for _ in range(n_blobs):
        c = rng.uniform(-spread, spread, 2)
        n = int(rng.randint(5, 600))
        xy = c + rng.randn(n, 2) * rng.uniform(0.1, 1.2, 2)
        pts.append(np.concatenate([xy, rng.uniform(-1.5, 0.5, (n, 1))], 1))
        mos.append((rng.rand(n) < 0.9).astype(np.int64))
    for n, m in ((n_noise, 1), (n_static, 0)):
        pts.append(np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(-2, 1, (n, 1))], 1))
        mos.append(np.full(n, m, np.int64))
    pts, mos = np.concatenate(pts).astype(np.float32), np.concatenate(mos)
    p = rng.permutation(len(pts))
    return pts[p], mos[p], (rng.randn(len(pts), 2) * 0.03).astype(np.float32)


def _run_cluster(dev, pts, mos, off, ti, use_offset):
    res = {}
    Cluster(CFG)(torch.from_numpy(pts).to(dev), torch.from_numpy(mos).to(dev), torch.from_numpy(off).to(dev),
                 torch.from_numpy(ti).to(dev), res, use_offset=use_offset)
    return res['inst_labels_est'].cpu().numpy()


# ------------------------------------------------------------------------------------------------ CPU leg
def test_oracle_cluster_golden(golden):
    g = golden('cluster')
    for key, use_offset in (('labels_offset', True), ('labels_plain', False)):
        got = oracle.cluster_forward(g['points'], g['mos'], g['offset'], g['time_indice'], float(g['eps']), int(g['min_samples']),
                                     int(g['min_p_cluster']), use_offset)
        assert np.array_equal(got, g[key])
    assert g['labels_offset'].max() > 10
    b = g['time_indice'][:, 0]
    assert g['labels_offset'][b == 2].max() == 0 and (g['mos'][b == 2] == 1).sum() > 0       # the <= min_p_cluster gate


def test_oracle_dbscan_is_sklearn():
    """The restated DBSCAN against scikit-learn itself (installed here): labels identical, border points included."""
    for seed in range(4):
        pts, _, _ = _scene(seed, 12, 80, 0, 12.0)
        x = pts.copy()
        x[:, 2] = 0
        assert np.array_equal(oracle.dbscan(x, 0.4, 5), _sklearn().fit_predict(x))


def test_host_mirror_cpu(golden, monkeypatch):
    from oracle import cpu_backend
    cpu_backend.install(monkeypatch)
    g = golden('cluster')
    got = _run_cluster(torch.device('cpu'), g['points'], g['mos'], g['offset'], g['time_indice'], True)
    assert np.array_equal(got, g['labels_offset'])


def test_product_refuses_cpu_tensors():
    from pcaccumulation_amd import native
    with pytest.raises((native.NativeError, OSError)):
        _run_cluster(torch.device('cpu'), np.zeros((4, 3), np.float32), np.ones(4, np.int64), np.zeros((4, 2), np.float32),
                     np.zeros((4, 2), np.int64), True)


# ------------------------------------------------------------------------------------------------ GPU leg
@pytest.mark.gpu
def test_cluster_golden_gpu(golden):
    g = golden('cluster')
    dev = torch.device('cuda:0')
    for key, use_offset in (('labels_offset', True), ('labels_plain', False)):
        got = _run_cluster(dev, g['points'], g['mos'], g['offset'], g['time_indice'], use_offset)
        assert np.array_equal(got, g[key]), (key, int((got != g[key]).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize('seed,n_blobs,spread', [(0, 30, 30.0), (1, 120, 40.0), (2, 6, 3.0), (3, 0, 10.0)])
def test_cluster_random_vs_sklearn_gpu(seed, n_blobs, spread):
    """Random scenes, 3 samples per batch (the last one possibly empty of moving points), checked against the oracle
    driven by scikit-learn's DBSCAN (fast enough for ~50k points)."""
    scenes = [_scene(10 * seed + i, n_blobs, 200, 2000, spread) for i in range(2)] + [_scene(99, 0, 3 if seed else 0, 500, spread)]
    pts = np.concatenate([s[0] for s in scenes])
    mos = np.concatenate([s[1] for s in scenes])
    off = np.concatenate([s[2] for s in scenes])
    ti = np.stack([np.concatenate([np.full(len(s[0]), i) for i, s in enumerate(scenes)]), np.zeros(len(pts), np.int64)], 1).astype(np.int64)
    dev = torch.device('cuda:0')
    for use_offset in (True, False):
        want = oracle.cluster_forward(pts, mos, off, ti, 0.4, 5, 15, use_offset, estimator=_sklearn())
        got = _run_cluster(dev, pts, mos, off, ti, use_offset)
        assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.gpu
def test_cluster_degenerate_gpu():
    dev = torch.device('cuda:0')
    # no moving point at all; all points identical; exactly min_p_cluster moving points (gate is a strict >)
    pts = np.random.RandomState(0).uniform(-5, 5, (300, 3)).astype(np.float32)
    ti = np.zeros((300, 2), np.int64)
    off = np.zeros((300, 2), np.float32)
    assert _run_cluster(dev, pts, np.zeros(300, np.int64), off, ti, True).max() == 0
    same = np.tile(pts[:1], (300, 1))
    want = oracle.cluster_forward(same, np.ones(300, np.int64), off, ti, 0.4, 5, 15, True, estimator=_sklearn())
    assert np.array_equal(_run_cluster(dev, same, np.ones(300, np.int64), off, ti, True), want)
    tight = (pts[:1] + np.random.RandomState(1).randn(300, 3) * 0.2).astype(np.float32)
    for k in (15, 16):
        mos = np.zeros(300, np.int64)
        mos[:k] = 1
        want = oracle.cluster_forward(tight, mos, off, ti, 0.4, 5, 15, False, estimator=_sklearn())
        assert np.array_equal(_run_cluster(dev, tight, mos, off, ti, False), want)


def _tiny_test_model(dev, golden):
    from helpers import make_batch
    from pcaccumulation_amd.motionnet import MotionNet
    from pcaccumulation_amd.synthetic import fill_state_dict_
    g = golden('model_tiny_test')
    cfg = default_config('waymo', 'test', n_sweeps=3, xy_range=8)
    inp = make_batch(cfg, [int(s) for s in g['seeds']], int(g['n_frames']), int(g['pts_per_frame']))
    model = MotionNet(cfg)
    fill_state_dict_(model)
    with torch.no_grad():
        sd = model.state_dict()
        for k, v in zip(g['tweak_keys'], g['tweak_vals']):
            sd[str(k)] += torch.from_numpy(v)
        sd['motionhead.offset_head.seg_head.3.weight'] *= float(g['offset_scale'])
        sd['motionhead.offset_head.seg_head.3.bias'] *= float(g['offset_scale'])
        sd['motionhead.mos_seg.seg_head.3.bias'] += torch.tensor([0.0, float(g['mos_shift'])])
    model = model.to(dev).eval()
    if dev.type == 'cuda':
        model.channels_last_()
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    torch.manual_seed(int(g['fwd_seed']))
    with torch.no_grad():
        out = model(inp)
    _assert_test_mode_mos_metric(cfg, out, inp, golden)
    return g, out


def _assert_test_mode_mos_metric(cfg, out, inp, golden):
    """SegTrainer.test calls FuseLoss.get_mos_loss on test-mode predictions (libs/tester.py:87); it supervises the points that are
    foreground in the ground truth OR in the estimate (libs/loss.py:145-147) although the test-mode forward decodes the estimated
    foreground only -- the index list the forward hands over must not be mistaken for that set."""
    from pcaccumulation_amd.loss import FuseLoss
    gl = golden('model_tiny_test_loss')
    assert int(gl['n_union']) > int(gl['n_est'])                             # the two sets differ on this scene
    assert '_mos_idx' not in out and '_gtfg_idx' not in out
    with torch.no_grad():
        mos = FuseLoss(cfg['loss']).get_mos_loss(out, inp)
    metric = mos['metric']
    if torch.is_tensor(metric):
        metric = {k: metric[i].cpu().numpy() for i, k in enumerate(('intersection', 'union', 'pred_positives', 'gt_positives'))}
    for k in ('intersection', 'union', 'pred_positives', 'gt_positives'):
        np.testing.assert_allclose(np.asarray(metric[k], np.float64), gl['mos_' + k], atol=2.5e-3)     # counts / 1e3: a couple of points
    assert abs(float(mos['bce_loss']) - float(gl['bce_loss'])) < 2e-2 * max(1.0, float(gl['bce_loss']))


def _assert_tiny_test(g, out):
    lab = out['inst_labels_est'].cpu().numpy()
    # clustering sits behind fp32 conv outputs (offsets, moving/static argmax): a point that moves across a 5 cm voxel or
    # the eps radius may relabel a cluster, so the whole-model check is on agreement, the bit-exact one is test_cluster_golden
    assert lab.shape == g['inst_labels_est'].shape
    agree = (lab == g['inst_labels_est']).mean()
    assert agree > 0.99, agree
    assert abs(int(lab.max()) - int(g['inst_labels_est'].max())) <= 1
    np.testing.assert_allclose(out['ego_motion_est'].cpu().numpy(), g['ego_motion_est'], atol=1e-3)
    if agree == 1.0:
        np.testing.assert_allclose(out['inst_pose_est'].cpu().numpy(), g['inst_pose_est'], rtol=1e-2, atol=1e-2)
        np.testing.assert_allclose(out['rec_est'].cpu().numpy(), g['rec_est'], rtol=1e-2, atol=1e-2)
        assert abs(out['inst_l2_error'] - float(g['inst_l2_error'])) < 1e-3 * max(1.0, float(g['inst_l2_error']))


def test_model_test_mode_cpu(golden, monkeypatch):
    from oracle import cpu_backend
    cpu_backend.install(monkeypatch)
    _assert_tiny_test(*_tiny_test_model(torch.device('cpu'), golden))


@pytest.mark.gpu
def test_model_test_mode_gpu(golden):
    _assert_tiny_test(*_tiny_test_model(torch.device('cuda:0'), golden))
