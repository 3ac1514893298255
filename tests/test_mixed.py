"""The 'mixed' compute mode (ops.set_mixed): forward values of the fp32x3 mode, bf16 gradient graph inside the two convolution segments.

1. Forward: the SAME numbers as the fp32x3 mode, bit for bit, on the same weights / scene / seed -- also with every bf16 shadow filled with NaN
   (ops.set_poison): no forward value may come from a shadow.
2. Backward: per-parameter gradients against the fp32x3 mode's -- norms within 1 %, directions (cosine) within 1e-3 for every parameter whose
   gradient is not rounding noise.
3. The registry: shadows, views of shadows, concatenations (host logic, CPU).
"""
import numpy as np
import pytest
import torch

from helpers import make_batch
from pcaccumulation_amd import ops
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.loss import FuseLoss
from pcaccumulation_amd.motionnet import MotionNet
from pcaccumulation_amd.synthetic import fill_state_dict_


def test_twin_registry_views_and_cat():
    ops.set_split(True)
    ops.set_mixed(True)
    x = torch.randn(2, 4, 6, 8)
    x16 = ops.shadow(x)
    assert ops.twin(x16) is x
    v = x16.permute(0, 3, 1, 2)[:, 2:6]                                    # a channel slice of the NCHW view
    assert torch.equal(ops.twin(v), x.permute(0, 3, 1, 2)[:, 2:6])
    assert torch.equal(ops.twin(x16.detach().view(8, 6, 8)), x.view(8, 6, 8))
    a = torch.randn(2, 3, 4, 4).contiguous(memory_format=torch.channels_last)
    b = torch.randn(2, 5, 4, 4).contiguous(memory_format=torch.channels_last)
    c = ops.cat_maps((ops.shadow(a), ops.shadow(b)), 1)
    assert c.dtype == torch.bfloat16 and torch.equal(ops.twin(c), torch.cat((a, b), 1))
    with pytest.raises(RuntimeError):
        ops.twin(torch.zeros(4, 4, dtype=torch.bfloat16))                   # a bf16 tensor nobody registered: refused, not up-cast
    ops.twins_clear()
    with pytest.raises(RuntimeError):
        ops.twin(x16)


def _model(cfg, dev):
    model = MotionNet(cfg)
    fill_state_dict_(model)
    with torch.no_grad():
        model.semseg_head.seg_head[3].bias += torch.tensor([2.0, 0.0])     # some pillars foreground, some background
    return model.to(dev).train().channels_last_()


def _step(mode, inp, poison=False, backward=True):
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
    cfg['misc']['compute_dtype'] = mode
    model = _model(cfg, dev)
    ops.set_poison(poison)
    torch.manual_seed(7)
    out = model(inp)
    stats = FuseLoss(cfg['loss'])(out, inp)
    if backward:
        stats['loss'].backward()
    ops.set_poison(False)
    keep = {k: out[k].detach().float().clone() for k in ('fb_seg_est', 'mos_est', 'offset_est', 'rec_est', 'ego_motion_est', 'transformed_points')}
    return keep, float(stats['loss'].detach()), ({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None} if backward else None)


@pytest.mark.gpu
def test_mixed_forward_is_the_fp32x3_forward_and_never_reads_a_shadow():
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
    inp = make_batch(cfg, [51, 52], 3, 6000)
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    ref, ref_loss, _ = _step('fp32x3', inp, backward=False)
    again, _, _ = _step('fp32x3', inp, backward=False)                     # the fp32x3 forward against itself (the TubeNet's LDS-atomic row sums)
    for poison in (False, True):
        got, loss, _ = _step('mixed', inp, poison=poison, backward=False)
        bad = []
        for k in ref:
            if not bool(torch.isfinite(got[k]).all()):
                bad.append((k, 'non-finite'))
                continue
            # same kernels on the same fp32 inputs; the scales come from maxima collected with atomics (order-independent)
            d, scale = float((got[k] - ref[k]).abs().max()), float(ref[k].abs().max())
            if d > max(1e-6 * scale, 4 * float((again[k] - ref[k]).abs().max())):
                bad.append((k, d, scale))
        assert not bad, (poison, bad)
        assert abs(loss - ref_loss) <= 1e-6 * abs(ref_loss), (poison, loss, ref_loss)


@pytest.mark.gpu
def test_mixed_gradients_against_fp32x3():
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
    inp = make_batch(cfg, [51, 52], 3, 6000)
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    _, _, ref = _step('fp32x3', inp)
    _, _, again = _step('fp32x3', inp)
    _, _, got = _step('mixed', inp)
    assert got.keys() == ref.keys()
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref.values())))
    bad = []
    for k, g in ref.items():
        n = float(g.norm())
        if n < 1e-5 * total:                                               # rounding noise on both sides (a conv bias in front of a BatchNorm)
            continue
        own = float((again[k] - g).norm()) / n                              # the fp32x3 step against itself (atomics in the row sums)
        dn = abs(float(got[k].norm()) - n) / n
        cos = float((got[k].double() * g.double()).sum() / (got[k].double().norm() * g.double().norm()))
        if dn > max(1e-2, 4 * own) or 1 - cos > max(1e-3, 4 * own):
            bad.append((k, dn, 1 - cos, own))
    assert not bad, bad[:10]


@pytest.mark.gpu
@pytest.mark.parametrize('n,h,w,c,pitch', [(3, 16, 24, 32, 32), (2, 17, 9, 64, 128), (1, 288, 288, 32, 64)])
def test_pool_backward_with_fp32_winners(n, h, w, c, pitch):
    """pcacc_pool_skip_relu_backward_strided_y32 (y f32, gradients bf16) == the f32 kernel on the up-cast gradients, rounded once -- and NOT what a
    bf16 copy of y gives (values closer than 2^-8 tie there)."""
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(n + h + c)
    y = torch.relu(torch.randn(n, h, w, c, generator=g) + 0.5).to(dev)
    y[0, :2, :2, :4] = torch.tensor([1.0, 1.0 + 2.0 ** -12, 1.0 + 2.0 ** -11, 1.0 - 2.0 ** -12], device=dev).view(2, 2, 1)   # one bf16 value, four fp32 values
    gp = torch.randn(n, h // 2, w // 2, c, generator=g).to(dev).to(torch.bfloat16)
    wide = torch.randn(n, h, w, pitch, generator=g).to(dev).to(torch.bfloat16)
    gs = wide[..., pitch - c:]                                              # a channel slice of a wider map, read in place
    got = native.pool_skip_relu_backward(y, gp, gs)
    ref = native.pool_skip_relu_backward(y, gp.float(), gs.float().contiguous()).to(torch.bfloat16)
    assert got.dtype == torch.bfloat16 and torch.equal(got, ref)
    for a, b in ((gp, None), (None, gs)):
        assert torch.equal(native.pool_skip_relu_backward(y, a, b),
                           native.pool_skip_relu_backward(y, a.float() if a is not None else None, b.float().contiguous() if b is not None else None).to(torch.bfloat16))
    low = native.pool_skip_relu_backward(y.to(torch.bfloat16), gp, gs)      # winners from the rounded copy
    assert not torch.equal(low[0, :2, :2, :4], got[0, :2, :2, :4])


@pytest.mark.gpu
@pytest.mark.parametrize('n,t,h,w,ci,co,kt', [(2, 1, 16, 32, 32, 32, 1), (4, 2, 33, 70, 32, 64, 3), (2, 1, 18, 18, 128, 64, 1), (1, 1, 100, 300, 64, 64, 1)])
def test_split_kernels_second_output_is_the_rounded_first(n, t, h, w, ci, co, kt):
    """pcacc_conv3x3_split_dual / pcacc_upconv2x2_split_dual: the bf16 shadow written by the epilogue == the fp32 result rounded to nearest even,
    and the fp32 result == the plain entry's."""
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(n + h + ci + co)
    x = torch.randn(n, h, w, ci, generator=g).to(dev)
    wt = (torch.randn(*((co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)), generator=g) / (4 * ci ** 0.5)).to(dev)
    bias = torch.randn(co, generator=g).to(dev)
    wf, _ = native.conv3x3_split_prepare_weights(wt)
    am = native.absmax256(x)
    for force in ('0', '2'):                                                # streaming and resident kernels
        import os
        os.environ['PCACC_CONV_RES'] = force
        native.reload_switches()
        try:
            y, ya = native.conv3x3_split(x, wf, bias, t, True, amax=am, want_amax=True)
            y2, ya2, y16 = native.conv3x3_split(x, wf, bias, t, True, amax=am, want_amax=True, want_bf16=True)
        finally:
            del os.environ['PCACC_CONV_RES']
            native.reload_switches()
        assert torch.equal(y, y2) and torch.equal(ya, ya2) and torch.equal(y16, y.to(torch.bfloat16))
    if kt == 1 and native.upconv2x2_split_supported(h, w, ci, co):
        wu = (torch.randn(ci, co, 2, 2, generator=g) / (2 * ci ** 0.5)).to(dev)
        uf, _ = native.upconv2x2_split_prepare_weights(wu)
        u, ua = native.upconv2x2_split(x, am, uf, bias, 0)
        u2, ua2, u16 = native.upconv2x2_split(x, am, uf, bias, 0, want_bf16=True)
        assert torch.equal(u, u2) and torch.equal(u16, u.to(torch.bfloat16))
        # written straight into the first halves of two concatenation buffers (pixel pitch 2 c_up): same values, the other halves untouched
        b32 = torch.full((n, 2 * h, 2 * w, 2 * co), 7.0, device=dev)
        b16 = torch.full((n, 2 * h, 2 * w, 2 * co), 7.0, device=dev, dtype=torch.bfloat16)
        native.upconv2x2_split(x, am, uf, bias, 0, want_bf16=True, into=(b32, b16))
        assert torch.equal(b32[..., :co], u) and torch.equal(b16[..., :co], u16)
        assert bool((b32[..., co:] == 7.0).all()) and bool((b16[..., co:] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize('pooled', [False, True])
def test_pfn_block_second_outputs_are_the_rounded_first(pooled):
    """pcacc_pfn_block_split_forward_dual: out16 / hr16 == bf16 of the plain entry's out / relu(h); out itself unchanged."""
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(5)
    rows, m = 70000, 9000
    xa = torch.randn(rows, 32 if pooled else 64, generator=g).to(dev)
    pl = torch.randn(m, 32, generator=g).to(dev) if pooled else None
    p2v = torch.randint(0, m, (rows,), generator=g).to(dev).to(torch.int32) if pooled else None
    w0, ws, w1 = (torch.randn(32, 64, generator=g) / 8).to(dev), (torch.randn(32, 64, generator=g) / 8).to(dev), (torch.randn(32, 32, generator=g) / 6).to(dev)
    b0, b1 = torch.randn(32, generator=g).to(dev), torch.randn(32, generator=g).to(dev)
    am, pm = native.absmax256(xa), (native.absmax256(pl) if pooled else None)
    out, hr, _, _, oa, _ = native.pfn_block_split_forward(xa, am, pl, pm, p2v, w0, b0, ws, w1, b1)
    out2, oa2, out16, hr16 = native.pfn_block_split_forward_dual(xa, am, pl, pm, p2v, w0, b0, ws, w1, b1)
    assert torch.equal(out, out2) and torch.equal(oa, oa2)
    assert torch.equal(out16, out.to(torch.bfloat16)) and torch.equal(hr16, hr.to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize('k,n', [(32, 32), (64, 128), (128, 128), (128, 64)])
def test_rows_linear_second_output_is_the_rounded_first(k, n):
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(k + n)
    x = torch.randn(50000, k, generator=g).to(dev)
    w, b = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev), torch.randn(n, generator=g).to(dev)
    res = torch.randn(50000, n, generator=g).to(dev)
    am = native.absmax256(x)
    y, ya = native.rows_linear_split(x, am, w, b, res, True, True, want_amax=True)
    y2, ya2, y16 = native.rows_linear_split(x, am, w, b, res, True, True, want_bf16=True)
    assert torch.equal(y, y2) and torch.equal(ya, ya2) and torch.equal(y16, y.to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize('k,n', [(9, 64), (3, 32), (4, 128), (2, 8)])
@pytest.mark.parametrize('relu', [False, True])
def test_few_input_rows_second_output_and_maxima(k, n, relu):
    """pcacc_rows_linear_few_dual: the fp32 result is pcacc_rows_linear's bit for bit, the bf16 output its rounding, the 256 partial maxima bound
    it exactly (the maximum of the array == the maximum of |y|)."""
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(7 * k + n)
    rows = 70001
    x = torch.randn(rows, k, generator=g).to(dev)
    w, b = torch.randn(n, k, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
    res = torch.randn(rows, n, generator=g).to(dev) if relu else None
    ref = native.rows_linear(x, w, b, res, relu, relu, out_dtype=torch.float32)
    y, ya, y16 = native.rows_linear_few_dual(x, w, b, res, relu, relu)
    assert torch.equal(y, ref) and torch.equal(y16, ref.to(torch.bfloat16))
    assert ya.shape == (256,) and float(ya.max()) == float(ref.abs().max())
    x[5, 0] = float('nan')
    _, ya, _ = native.rows_linear_few_dual(x, w, b, None, False, False)
    assert float(ya.max()) == float('inf')                      # a NaN anywhere poisons the scale the way pcacc_absmax256 does


@pytest.mark.gpu
def test_concatenated_twin_keeps_its_maxima():
    """ops.cat_maps in the mixed mode: the consumer's view of the concatenated twin (rebuilt from the storage) finds the merged maxima -- no pass over it."""
    dev = torch.device('cuda:0')
    ops.set_split(True)
    ops.set_mixed(True)
    try:
        tw = [torch.randn(2, 8, 8, 32, device=dev) for _ in range(2)]
        sh = []
        for t in tw:
            ops.set_amax_tag(t, native_absmax(t))
            sh.append(ops.shadow(t).permute(0, 3, 1, 2))
        y = ops.cat_maps(sh, 1)
        rows = y.permute(0, 2, 3, 1)
        t32 = ops.twin(rows)
        tag = ops.amax_tag(t32)
        assert tag is not None and float(tag.max()) == float(torch.cat(tw, -1).abs().max())
    finally:
        ops.set_mixed(False)
        ops.set_split(False)
        ops.twins_clear()


def native_absmax(t):
    from pcaccumulation_amd import native
    return native.absmax256(t)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape,ca,cb', [((3, 17, 9), 32, 32), ((2, 36, 36), 256, 256), ((5000,), 64, 64), ((4, 8, 8), 8, 24)])
def test_own_concatenation_kernel(dtype, shape, ca, cb):
    """pcacc_cat2_rows == torch.cat along the channels, forward and (through ops.cat_maps) backward."""
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    a = torch.randn(*shape, ca, device=dev).to(dtype)
    b = torch.randn(*shape, cb, device=dev).to(dtype)
    assert torch.equal(native.cat2_rows(a, b), torch.cat((a, b), -1))
    if len(shape) == 3:
        an, bn = a.permute(0, 3, 1, 2).requires_grad_(True), b.permute(0, 3, 1, 2).requires_grad_(True)
        y = ops.cat_maps((an, bn), 1)
        assert torch.equal(y, torch.cat((an, bn), 1)) and y.permute(0, 2, 3, 1).is_contiguous()
        g = torch.randn_like(y)
        ga, gb = torch.autograd.grad(y, (an, bn), g)
        assert torch.equal(ga, g[:, :ca]) and torch.equal(gb, g[:, ca:])


@pytest.mark.gpu
@pytest.mark.parametrize('sel', [(5, 700), (8, 1024)])          # 8 192 cells: the weight gradient as a batched product over 16 slices of the cells
def test_sparse_conv_rows_is_the_dense_convolution_at_the_selected_cells(sel):
    """ops.SparseConvRows.at(cells) == rows `cells` of conv(h), values and gradients (h, weight, bias), incl. cells on the border and repeated cells."""
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(3)
    n, C, O, H, W = 3, 64, 64, 20, 24
    h = torch.randn(n, C, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    conv = torch.nn.Conv2d(C, O, 3, padding=1).to(dev)
    cells = torch.randint(0, n * H * W, sel, generator=g).to(dev)
    cells[0, :6] = torch.tensor([0, W - 1, (H - 1) * W, H * W - 1, 5, 5], device=dev)          # corners of image 0, one cell twice
    assert ops.sparse_conv_available(h, conv)
    got = ops.SparseConvRows(h, conv).at(cells)
    gy = torch.randn(*sel, O, generator=g).to(dev)
    gh, gw, gb = torch.autograd.grad(got, (h, conv.weight, conv.bias), gy)
    hd = h.detach().double().requires_grad_(True)
    wd, bd = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    dense = torch.nn.functional.conv2d(hd, wd, bd, padding=1).permute(0, 2, 3, 1).reshape(-1, O)[cells]
    rh, rw, rb = torch.autograd.grad(dense, (hd, wd, bd), gy.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    assert rel(got, dense.detach()) < 1e-5 and rel(gh, rh) < 1e-5 and rel(gw, rw) < 1e-5 and rel(gb, rb) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fp32x3', 'mixed'])
def test_sparse_ego_feature_head_matches_the_dense_head(mode, monkeypatch):
    """With the device key-point sampler the ego feature head's last convolution is evaluated at the key-point cells only: same poses, loss and
    gradients as the dense head (PCACC_SPARSE_EGO=0) on the same seed."""
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
    cfg['misc']['compute_dtype'] = mode
    cfg['pose_estimation']['kpt_sampler'] = 'device'
    inp = make_batch(cfg, [61, 62], 3, 6000)
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    res = {}
    for flag in ('0', '1'):
        monkeypatch.setenv('PCACC_SPARSE_EGO', flag)
        model = _model(cfg, dev)
        torch.manual_seed(11)
        out = model(inp)
        stats = FuseLoss(cfg['loss'])(out, inp)
        stats['loss'].backward()
        res[flag] = (out['ego_motion_est'].detach().clone(), float(stats['loss'].detach()),
                     {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    (pose0, loss0, g0), (pose1, loss1, g1) = res['0'], res['1']
    assert float((pose0 - pose1).abs().max()) < 1e-4 and abs(loss0 - loss1) < 1e-4 * abs(loss0)
    assert g0.keys() == g1.keys()
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g0.values())))
    bad = [(k, float((g1[k] - g0[k]).norm()) / float(g0[k].norm())) for k in g0
           if float(g0[k].norm()) > 1e-5 * total and float((g1[k] - g0[k]).norm()) > 5e-2 * float(g0[k].norm())]      # (the STPN's temporal-conv biases move by 2 % for poses that differ by 1e-5: the known sensitive group)
    assert not bad, bad[:8]


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['mixed', 'bf16', 'fp32x3'])
def test_pillar_major_rows_change_no_forward_bit(mode, monkeypatch):
    """[r6] The pillar encoder on rows stored pillar by pillar (PillarIndex.pillar_major: the reference's own [M, max_points, C] order,
    libs/voxel_generator.py:41-58) against the same encoder on rows in point order: a per-pillar maximum does not depend on where its rows lie, a stable
    order keeps every tie -- every result of the forward is bit-identical (LiDAR-distributed points: crowded pillars); the gradients add the same
    terms in another order."""
    dev = torch.device('cuda:0')
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
    inp = make_batch(cfg, [61, 62], 3, 9000, mode='lidar')
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    runs = {}
    for major in (False, True):
        cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=16)
        cfg['misc']['compute_dtype'] = mode
        cfg['misc']['pillar_major_rows'] = major
        model = _model(cfg, dev)
        assert model.pillar_major_rows == major
        torch.manual_seed(7)
        out = model(inp)
        stats = FuseLoss(cfg['loss'])(out, inp)
        stats['loss'].backward()
        keep = {k: out[k].detach().clone() for k in ('fb_seg_est', 'mos_est', 'offset_est', 'rec_est', 'ego_motion_est', 'transformed_points')}
        runs[major] = (keep, stats['loss'].detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    for k, v in runs[False][0].items():
        assert torch.equal(v, runs[True][0][k]), k
    assert torch.equal(runs[False][1], runs[True][1])
    ref, got = runs[False][2], runs[True][2]
    assert ref.keys() == got.keys()
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref.values())))
    for k, g in ref.items():
        if not k.startswith('pillar_encoder.'):
            assert torch.equal(g, got[k]), k                                # behind the canvas nothing knows about the rows' order
            continue
        n = float(g.norm())
        if n < 1e-5 * total:
            continue
        assert float((got[k] - g).norm()) <= 2e-2 * n, (k, float((got[k] - g).norm()) / n)      # bf16 rows in the backward: sums in another order
