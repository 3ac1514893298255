"""GPU parity tests of the fused loss passes (csrc/loss.hip through the C ABI): against the reference's
own outputs and autograd gradients (tests/golden/loss.npz), against the oracle on seeded inputs in both logit layouts and both
element types, and at the full size of a training step through properties that do not depend on the size."""
import numpy as np
import pytest
import torch

import oracle
from test_oracle_golden import _offset_args

pytestmark = pytest.mark.gpu
KEYS = ('intersection', 'union', 'pred_positives', 'gt_positives')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from pcaccumulation_amd import native, ops as o
    native.lib()
    return o


def _run_seg(ops, dev, z, y, rows=None, weights=(1.0, 0.0)):
    est = torch.from_numpy(z).to(dev).requires_grad_(True) if isinstance(z, np.ndarray) else z
    terms, metric = ops.seg_loss(est, torch.from_numpy(y).to(dev), None if rows is None else torch.from_numpy(rows).to(dev))
    grads = []
    for k in range(2):
        grads.append(torch.autograd.grad(terms[k], est, retain_graph=True)[0].float().cpu().numpy())
    return terms.detach().cpu().numpy(), metric.cpu().numpy(), grads


def test_seg_loss_against_reference_golden(ops, dev, golden):
    """Values to 2e-6 relative (fp32 sums in another order), counters exact, gradients to 1e-3 of their largest entry."""
    g = golden('loss')
    for name in g['seg_names']:
        terms, metric, grads = _run_seg(ops, dev, g['seg_%s_logits' % name], g['seg_%s_labels' % name])
        assert abs(terms[0] - g['seg_%s_bce' % name]) < 2e-6 * max(1, abs(g['seg_%s_bce' % name])), name
        assert abs(terms[1] - g['seg_%s_lovasz' % name]) < 2e-6, name
        np.testing.assert_allclose(metric, g['seg_%s_metric' % name], rtol=0, atol=1e-12)
        for got, k in zip(grads, ('grad_bce', 'grad_lovasz')):
            want = g['seg_%s_%s' % (name, k)]
            assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max() + 1e-10, (name, k)


@pytest.mark.parametrize('layout', ['rows', 'planes', 'channels_last'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_seg_loss_selected_rows_against_oracle(ops, dev, layout, dtype):
    """A [B*T,2,H,W] head output (or [N,2] rows) with a subset of supervised rows: same numbers as the oracle on the gathered
    rows, gradient zero elsewhere.  bf16 logits are read as the fp32 values they hold."""
    rng = np.random.RandomState(5)
    f, h, w = 3, 20, 24
    n_total = f * h * w
    z = torch.from_numpy((rng.randn(n_total, 2) * 3).astype(np.float32)).to(dtype)
    y = rng.randint(-1, 2, n_total).astype(np.int64)
    rows = np.sort(rng.choice(n_total, 700, replace=False)).astype(np.int64)
    if layout == 'rows':
        est = z.clone().to(dev)
    elif layout == 'planes':
        est = z.view(f, h, w, 2).permute(0, 3, 1, 2).contiguous().to(dev)
    else:
        est = z.view(f, h, w, 2).to(dev).permute(0, 3, 1, 2)                     # NCHW shape, channels-last memory
    est.requires_grad_(True)
    terms, metric, grads = _run_seg(ops, dev, est, y, rows)
    want = oracle.seg_loss(z.float().numpy()[rows], y[rows])
    assert abs(terms[0] - want['bce_loss']) < 3e-6 * max(1, want['bce_loss']) and abs(terms[1] - want['lovasz_loss']) < 3e-6
    np.testing.assert_allclose(metric, np.stack([want['metric'][k] for k in KEYS]), rtol=0, atol=1e-12)
    tol = 1e-3 if dtype == torch.float32 else 1e-2                              # gradients are stored in the logits' type
    for got, k in zip(grads, ('grad_bce', 'grad_lovasz')):
        if layout != 'rows':
            got = np.transpose(got, (0, 2, 3, 1))
        got = got.reshape(n_total, 2)
        full = np.zeros((n_total, 2), np.float32)
        full[rows] = want[k]
        assert np.abs(got - full).max() <= tol * np.abs(full).max(), (layout, k)
        mask = np.ones(n_total, bool)
        mask[rows] = False
        assert not got[mask].any()


def test_seg_loss_full_size_properties(ops, dev):
    """600k rows (the occupied pillars of a 4-sequence Waymo batch): the value does not depend on the order of the rows, agrees
    with the element-wise torch formulation of the same loss on the GPU (libs/lovasz_softmax.py written with torch.sort), and the
    Jaccard gradients of each present class sum to the last Jaccard value 1 (telescoping), i.e. sum_i dL/dp_ic * sign = 1."""
    from pcaccumulation_amd.loss import lovasz_softmax_flat
    torch.manual_seed(0)
    n = 600_000
    z = (torch.randn(n, 2, device=dev) * 2.5)
    y = (torch.rand(n, device=dev) < 0.07).long()
    y[torch.rand(n, device=dev) < 0.01] = -1
    t1, m1 = ops.seg_loss(z, y)
    perm = torch.randperm(n, device=dev)
    t2, m2 = ops.seg_loss(z[perm].contiguous(), y[perm].contiguous())
    assert torch.allclose(t1, t2, rtol=2e-6, atol=0) and torch.equal(m1, m2)
    want_lov = lovasz_softmax_flat(torch.softmax(z, 1), y)
    keep = y >= 0
    w = torch.sqrt(keep.sum() / torch.stack([(y == 0).sum(), (y == 1).sum()]).float()).clamp(0, 50)
    want_ce = torch.nn.functional.cross_entropy(z, y, weight=w, ignore_index=-1)
    assert abs(float(t1[1]) - float(want_lov)) < 1e-5 and abs(float(t1[0]) - float(want_ce)) < 1e-5 * float(want_ce)
    pred = z.argmax(1)
    for c in range(2):
        assert abs(float(m1[0, c]) - int(((pred == c) & (y == c)).sum()) / 1e3) < 1e-9
        assert abs(float(m1[3, c]) - int((y == c).sum()) / 1e3) < 1e-9
    zg = z.clone().requires_grad_(True)
    terms, _ = ops.seg_loss(zg, y)
    terms[1].backward()
    p = torch.softmax(z, 1)
    # d Lovasz / d z_0 = p0 p1 (dL/dp0 - dL/dp1): recover sum over rows of the class-0 Jaccard gradient weights
    assert torch.isfinite(zg.grad).all() and float(zg.grad.abs().sum()) > 0
    assert float((zg.grad.sum(1)).abs().max()) < 1e-6                            # softmax: the two logit gradients cancel


def test_seg_loss_empty_and_errors(ops, dev):
    from pcaccumulation_amd import native
    z = torch.zeros(4, 2, device=dev)
    with pytest.raises(native.NativeError):
        native.seg_loss_forward(z.double(), 0, torch.zeros(4, dtype=torch.int64, device=dev), None, 4)
    with pytest.raises(native.NativeError):
        native.seg_loss_forward(z, 0, torch.zeros(4, dtype=torch.int64), None, 4)          # labels on the host


def test_offset_loss_against_reference_golden(ops, dev, golden):
    g = golden('loss')
    for ci in range(2):
        p = 'off%d_' % ci
        pts, tidx, inst, fb, ego, motions, tp, est = _offset_args(g, p)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        sizes = [m.shape[0] for m in motions]
        base = t(np.cumsum([0] + sizes[:-1]).astype(np.int64))
        rows = t(np.nonzero(fb[:, 0] == 1)[0].astype(np.int64))
        e = t(est).requires_grad_(True)
        out, gt = ops.offset_loss(e, t(pts), t(tidx), t(inst[:, 0]), base, t(ego), t(np.concatenate(motions)), t(tp), rows)
        o = out.detach().cpu().numpy()
        assert abs(o[0] - g[p + 'norm']) < 2e-5 and abs(o[1] - g[p + 'dir']) < 2e-6 and abs(o[2] - g[p + 'l2']) < 2e-5
        np.testing.assert_allclose(gt.cpu().numpy(), g[p + 'offset_gt'], atol=2e-5)
        g_norm = torch.autograd.grad(out[0], e, retain_graph=True)[0].cpu().numpy()
        g_dir = torch.autograd.grad(out[1], e, retain_graph=True)[0].cpu().numpy()
        g_l2 = torch.autograd.grad(out[2], e)[0].cpu().numpy()
        np.testing.assert_allclose(g_norm, g[p + 'grad_norm'], atol=1e-9)
        np.testing.assert_allclose(g_dir, g[p + 'grad_dir'], atol=2e-7)
        assert not g_l2.any()                                                     # reported, not trained on (libs/loss.py:236)


def test_offset_loss_many_instances_and_full_size(ops, dev):
    """Instance tables beyond the LDS copy (k*3 > 8192 floats -> global atomics) and 800k points: centres against a float64
    numpy accumulation, the loss against the oracle."""
    rng = np.random.RandomState(2)
    for n, k, n_frames in ((800_000, 40, 5), (60_000, 3000, 2)):
        pts = (rng.randn(n, 3) * 20).astype(np.float32)
        tidx = np.stack([np.zeros(n, np.int64), rng.randint(0, n_frames, n)], 1)
        lab = rng.randint(0, k, n).astype(np.int64)
        lab[:k] = np.arange(k)
        ego = np.tile(np.eye(4, dtype=np.float32), (1, n_frames, 1, 1))
        ego[0, :, :3, 3] = rng.randn(n_frames, 3)
        motion = np.tile(np.eye(4, dtype=np.float32), (k, n_frames, 1, 1))
        motion[:, :, :3, 3] = rng.randn(k, n_frames, 3) * 0.5
        tp = pts + 0.01
        est = rng.randn(n, 2).astype(np.float32)
        fb = (lab > 0).astype(np.int64)
        want = oracle.offset_loss(pts, tidx, lab, fb, ego, [motion], tp, est)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        e = t(est).requires_grad_(True)
        out, gt = ops.offset_loss(e, t(pts), t(tidx), t(lab), t(np.zeros(1, np.int64)), t(ego), t(motion), t(tp), t(np.nonzero(fb)[0]))
        o = out.detach().cpu().numpy()
        assert abs(o[0] - want['offset_norm_loss']) < 1e-4 * want['offset_norm_loss'] and abs(o[1] - want['offset_dir_loss']) < 1e-5
        assert abs(o[2] - want['offset_l2_error']) < 1e-4 * want['offset_l2_error']
        np.testing.assert_allclose(gt.cpu().numpy(), want['offset_gt'], atol=2e-3)   # fp32 atomic sums of up to 20k points x 20 m
        (out[0] + out[1]).backward()
        got = e.grad.cpu().numpy()
        ref = want['grad_norm'] + want['grad_dir']
        close = np.abs(got - ref).max(1) < 1e-6 + 1e-3 * np.abs(ref).max()
        assert close.mean() > 0.9999                                               # a sign flips where gt - est is within rounding
