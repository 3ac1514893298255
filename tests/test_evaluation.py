"""Scene-flow evaluation: per-scene errors, metric definitions and the flow_error.npz dump against the
reference (tests/golden/eval_tiny.npz: toolbox/sf_eval_utils.py:71-86 and the tester's error computation on the tiny scene)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import make_batch
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.evaluation import FlowErrorDump, compute_sf_metrics, flow_errors

METRICS = ('EPE3D', 'EPE3D_med', 'Acc3DS', 'Acc3DR', 'Outlier', 'ROutlier')


def _scene(golden, dev):
    g, gm = golden('eval_tiny'), golden('model_tiny_val')
    cfg = default_config('waymo', 'val', n_sweeps=3, xy_range=8)
    inp = make_batch(cfg, [int(gm['seeds'][0])], 3, int(gm['pts_per_frame']))
    n0 = int(g['n_points'])
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    return g, {'rec_est': torch.from_numpy(gm['rec_est'][:n0]).to(dev)}, inp


def test_oracle_flow_errors_and_metrics(golden):
    g, pred, inp = _scene(golden, torch.device('cpu'))
    epe, rel = oracle.flow_errors(pred['rec_est'].numpy(), inp['input_points'].numpy(), inp['time_indice'][:, 1].numpy(),
                                  inp['ego_motion_gt'][0].numpy(), inp['inst_labels'][:, 0].numpy(), inp['inst_motion_gt'][0].numpy(), 3)
    np.testing.assert_allclose(epe, g['epe_per_point'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rel, g['relative_error'], rtol=2e-4, atol=1e-6)
    m = oracle.compute_sf_metrics(g['epe_per_point'], g['relative_error'])
    for k in METRICS:
        assert abs(m[k] - float(g[k])) < 1e-5 * max(1.0, abs(float(g[k]))), k
    ms = oracle.compute_sf_metrics(g['synth_epe'], g['synth_rel'])           # errors around the 0.05 / 0.1 / 0.3 thresholds
    mt = compute_sf_metrics(torch.from_numpy(g['synth_epe']), torch.from_numpy(g['synth_rel']))
    for k in METRICS:
        assert abs(ms[k] - float(g['synth_' + k])) < 1e-6, k
        got = mt[k][0] if isinstance(mt[k], list) else mt[k]
        assert abs(got - float(g['synth_' + k])) < 1e-6, k
    assert 0.05 < float(g['synth_Acc3DS']) < 0.95 and 0.0 < float(g['synth_ROutlier']) < 0.95 and 0.05 < float(g['synth_Outlier']) < 0.95


def _check(golden, dev, use_native=False, monkeypatch=None):
    if monkeypatch is not None:
        from oracle import cpu_backend
        cpu_backend.install(monkeypatch)
    g, pred, inp = _scene(golden, dev)
    err = flow_errors(pred, inp, 3)
    np.testing.assert_allclose(err['epe_per_point'].cpu().numpy(), g['epe_per_point'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(err['relative_error'].cpu().numpy(), g['relative_error'], rtol=2e-4, atol=1e-6)
    assert np.array_equal(err['time_indice'].cpu().numpy(), g['time_indice'])
    m = compute_sf_metrics(err['epe_per_point'], err['relative_error'])
    for k in METRICS:
        got = m[k][0] if isinstance(m[k], list) else m[k]
        assert abs(got - float(g[k])) < 1e-5, k
    assert m['EPE3D'][1] == int(g['size'])
    return err


def test_flow_errors_cpu(golden, monkeypatch, tmp_path):
    err = _check(golden, torch.device('cpu'), monkeypatch=monkeypatch)
    dump = FlowErrorDump()
    dump.add(err)
    dump.add(err)
    data = np.load(dump.save(str(tmp_path)))
    assert set(data.files) == {'fb_label', 'sd_label', 'epe_per_point', 'relative_error', 'time_indice'}       # libs/tester.py:98-104
    assert data['epe_per_point'].dtype == np.float16 and data['time_indice'].dtype == np.int8 and data['fb_label'].dtype == np.bool_
    assert data['epe_per_point'].shape[0] == 2 * err['epe_per_point'].shape[0]


@pytest.mark.gpu
def test_flow_errors_gpu(golden):
    _check(golden, torch.device('cuda:0'))


def _flat_meters(meters, prefix=''):
    out = {}
    for k, v in meters.items():
        if isinstance(v, dict):
            out.update(_flat_meters(v, prefix + k + '/'))
        else:
            out[prefix + k] = [float(v.avg), float(v.sum), int(v.count)]
    return out


def _collect(golden, tmp_path, dev):
    """collect_results (toolbox/evaluation.py:20-98) against the reference's own run on the same three synthetic scenes
    (tests/golden/make_golden_collect.py): running meters, per-scene dictionaries, sampled dynamic errors, the three files."""
    import json
    import pickle
    from helpers import flow_error_scenes
    from pcaccumulation_amd.evaluation import collect_results
    g = golden('collect')
    src, dst = str(tmp_path / 'results'), str(tmp_path / 'metrics')
    flow_error_scenes(src)
    meters, scenes = collect_results(src, dst, 'waymo', device=dev)
    want = json.loads(str(g['static']))
    got = _flat_meters(meters)
    assert set(got) == set(want)
    for k, (avg, total, count) in want.items():
        assert got[k][2] == count, k
        assert abs(got[k][0] - avg) <= 1e-6 * max(1.0, abs(avg)) and abs(got[k][1] - total) <= 1e-6 * max(1.0, abs(total)), k
    ws = json.loads(str(g['scene']))
    assert set(scenes) == set(ws)

    def same(a, b, path):
        if isinstance(b, dict):
            assert set(a) == set(b), path
            for k in b:
                same(a[k], b[k], path + '/' + k)
        elif isinstance(b, list):
            same(a[0], b[0], path)
            assert a[1] == b[1], path
        elif b != b:                                           # an empty selection: the mean of nothing is NaN in the reference too
            assert a != a, path
        else:
            assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), path
    same(scenes, ws, '')
    dyn = torch.load(str(tmp_path / 'metrics' / 'dynamic_dict.pth'))
    assert dyn['relative_error'].dtype == torch.float32
    assert np.array_equal(np.sort(dyn['relative_error'].numpy()), g['dyn_rel_sorted'])
    assert np.array_equal(np.sort(dyn['epe_per_point'].numpy()), g['dyn_epe_sorted'])
    static = pickle.load(open(str(tmp_path / 'metrics' / 'static_stats.pkl'), 'rb'))
    assert abs(static['static_BG']['EPE3D'].avg - want['static_BG/EPE3D'][0]) < 1e-6            # what toolbox/evaluation.py:113 prints
    assert 'static_FG' in static and 'Dynamic' not in pickle.load(open(str(tmp_path / 'metrics' / 'scene_stats.pkl'), 'rb'))['scene_002']


def test_collect_results_cpu(golden, tmp_path):
    _collect(golden, tmp_path, 'cpu')


@pytest.mark.gpu
def test_collect_results_gpu(golden, tmp_path):
    _collect(golden, tmp_path, 'cuda:0')
