"""MFMA 3x3 / 3x3x3 convolution (include/pcacc.h A6/A9) against the library convolution on the same bf16 inputs.
Tolerance: both sides accumulate bf16 products in fp32 and round the result to bf16 once, so they differ by summation
order only: one bf16 ulp (2^-8 relative) of the largest term."""
import pytest
import torch
import torch.nn.functional as F

from pcaccumulation_amd import native, ops

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _ref_conv(x_rows, w, b, relu):
    y = F.conv2d(x_rows.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), b, padding=1)
    y = torch.relu(y) if relu else y
    return y.permute(0, 2, 3, 1)


def _close(a, b, scale):
    err = (a.float() - b.float()).abs().max().item()
    assert err <= scale * 2 ** -7, (err, scale)


@pytest.mark.parametrize('n,h,w,ci,co,relu', [
    (2, 16, 32, 32, 32, True), (1, 19, 45, 64, 32, False), (3, 8, 33, 32, 64, True), (1, 24, 64, 96, 32, True),
    (1, 9, 18, 128, 128, True), (1, 18, 18, 256, 128, False), (1, 7, 5, 128, 256, True), (2, 40, 40, 64, 64, True),
    # the deep layers of the two U-Nets (csrc/conv_deep.hip): strips of consecutive pixels, ragged last strips, 64-channel groups
    (3, 18, 18, 512, 512, True), (2, 36, 36, 256, 256, True), (1, 72, 72, 128, 128, False), (2, 36, 36, 512, 256, True),
    (1, 18, 18, 256, 512, False), (2, 20, 144, 128, 64, True), (1, 5, 288, 128, 64, True), (1, 19, 37, 128, 192, False),
    (1, 72, 72, 256, 128, True), (1, 3, 400, 128, 128, True)])
def test_conv3x3_forward(n, h, w, ci, co, relu):
    g = torch.Generator(device='cpu').manual_seed(n * 1000 + ci + co)
    x = torch.randn(n, h, w, ci, generator=g).to(DEV).to(torch.bfloat16)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    y = ops.conv3x3_rows(x, wt, b, 1, relu)
    ref = _ref_conv(x, wt, b, relu)
    assert y.dtype == torch.bfloat16 and y.shape == ref.shape
    _close(y, ref, ref.abs().max().item())


def test_conv3x3_identity_weights_asymmetric():
    """Centre-tap permutation weights: the output must be the input with channels permuted (catches operand swaps)."""
    ci = co = 64
    x = torch.randn(1, 12, 40, ci).to(DEV).to(torch.bfloat16)
    perm = torch.randperm(ci)
    wt = torch.zeros(co, ci, 3, 3)
    wt[torch.arange(co), perm, 1, 1] = 1.0
    y = ops.conv3x3_rows(x, wt.to(DEV), None, 1, False)
    assert torch.equal(y, x[..., perm.to(DEV)])
    wt = torch.zeros(co, ci, 3, 3)
    wt[torch.arange(co), perm, 0, 2] = 1.0           # tap (dy=-1, dx=+1): shifted copy with zero border
    y = ops.conv3x3_rows(x, wt.to(DEV), None, 1, False)
    ref = torch.zeros_like(x)
    ref[:, 1:, :-1] = x[:, :-1, 1:][..., perm.to(DEV)]
    assert torch.equal(y, ref)


@pytest.mark.parametrize('b,t,h,w,ci,co', [(2, 5, 16, 40, 32, 32), (1, 3, 9, 33, 32, 64)])
def test_conv3x3x3_forward(b, t, h, w, ci, co):
    g = torch.Generator(device='cpu').manual_seed(7)
    x = torch.randn(b * t, h, w, ci, generator=g).to(DEV).to(torch.bfloat16)
    wt = (torch.randn(co, ci, 3, 3, 3, generator=g) / (5 * ci ** 0.5)).to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    y = ops.conv3x3_rows(x, wt, bias, t, True)
    x5 = x.float().view(b, t, h, w, ci).permute(0, 4, 1, 2, 3)
    ref = torch.relu(F.conv3d(x5, wt.to(torch.bfloat16).float(), bias, padding=1)).permute(0, 2, 3, 4, 1).reshape(b * t, h, w, co)
    _close(y, ref, ref.abs().max().item())


@pytest.mark.parametrize('b,t,h,w,co', [(2, 5, 16, 40, 32), (1, 3, 9, 33, 64), (1, 5, 41, 70, 32), (3, 1, 8, 32, 32), (2, 2, 17, 64, 64)])
def test_conv3x3x3_kernels_agree(b, t, h, w, co, monkeypatch):
    """The 27-tap 32-channel layers run on unpadded XOR-swizzled LDS rows (two workgroups per CU) with their tiles in frame-fastest order;
    PCACC_CONV_FRAME_MAJOR selects the frame-by-frame order, PCACC_CONV_SWZ_OFF the padded-row kernel of round 2.  Same products in the
    same order per output: bit-identical, with and without an epilogue mask."""
    g = torch.Generator(device='cpu').manual_seed(11 + co)
    x = torch.randn(b * t, h, w, 32, generator=g).to(DEV).to(torch.bfloat16)
    wt = (torch.randn(co, 32, 3, 3, 3, generator=g) / (5 * 32 ** 0.5)).to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    mask = torch.randn(b * t, h, w, co, generator=g).to(DEV).to(torch.bfloat16)
    wp = native.conv3x3_prepare_weights(wt)
    run = lambda: (native.conv3x3(x, wp, bias, t, True), native.conv3x3(x, wp, None, t, False, out_mask=mask))
    got = run()
    for env in ({'PCACC_CONV_FRAME_MAJOR': '1'}, {'PCACC_CONV_SWZ_OFF': '1'}, {'PCACC_CONV_SWZ_OFF': '1', 'PCACC_CONV_FRAME_MAJOR': '1'}):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            native.reload_switches()                           # the launchers read their switches once per process
            ref = run()
        native.reload_switches()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), env


@pytest.mark.parametrize('kt', [1, 3])
def test_conv3x3_backward(kt):
    b, t, h, w, ci, co = 2, 3, 12, 36, 32, 64
    g = torch.Generator(device='cpu').manual_seed(11)
    x = torch.randn(b * t, h, w, ci, generator=g).to(DEV).to(torch.bfloat16).requires_grad_(True)
    shape = (co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)
    wt = (torch.randn(*shape, generator=g) / (4 * ci ** 0.5)).to(DEV).requires_grad_(True)
    bias = torch.randn(co, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(b * t, h, w, co, generator=g).to(DEV).to(torch.bfloat16)
    y = ops.conv3x3_rows(x, wt, bias, t if kt == 3 else 1, True)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    wr = wt.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    if kt == 3:
        x5 = xr.view(b, t, h, w, ci).permute(0, 4, 1, 2, 3)
        yr = torch.relu(F.conv3d(x5, wr, br, padding=1)).permute(0, 2, 3, 4, 1).reshape(b * t, h, w, co)
    else:
        yr = torch.relu(F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1)).permute(0, 2, 3, 1)
    # use the kernel's own ReLU mask: outputs within a bf16 ulp of zero may differ between the two
    (yr * (y.detach() > 0)).backward(gy.float())
    _close(x.grad, xr.grad, xr.grad.abs().max().item())
    assert (wt.grad - wr.grad).abs().max().item() <= 2e-2 * wr.grad.abs().max().item()
    assert (bias.grad - br.grad).abs().max().item() <= 2e-2 * br.grad.abs().max().item()


def test_conv3x3_module_dispatch():
    conv = torch.nn.Conv2d(32, 64, 3, padding=1).to(DEV)
    x = torch.randn(2, 32, 16, 32, device=DEV).to(memory_format=torch.channels_last)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = ops.conv3x3(x, conv, relu=True)
        ref = torch.relu(conv(x))
    assert y.dtype == torch.bfloat16 and y.shape == ref.shape
    _close(y, ref, ref.abs().max().item())
    y32 = ops.conv3x3(x, conv, relu=True)              # fp32, no autocast: library path, fp32 result
    assert y32.dtype == torch.float32
    with pytest.raises(native.NativeError):
        native.conv3x3(torch.zeros(1, 8, 8, 32, dtype=torch.bfloat16), torch.zeros(9, 32, 32, dtype=torch.bfloat16), None, 1, False)


@pytest.mark.parametrize('ci,co,kt,h,w', [(32, 32, 1, 19, 45), (64, 32, 1, 16, 40), (32, 64, 1, 33, 17), (64, 64, 1, 19, 45), (32, 32, 3, 16, 40),
                                          (128, 64, 1, 20, 70)])
def test_conv3x3_out_mask(ci, co, kt, h, w):
    """pcacc_conv3x3_outmask_bf16 (the data gradient of conv -> ReLU -> conv masked for the first ReLU in the epilogue) == the plain kernel
    followed by aten::threshold_backward, bit for bit -- zeros, negatives, NaN and inf in the mask included."""
    b, t = 2, 3
    g = torch.Generator(device='cpu').manual_seed(ci + 3 * co + kt)
    gy = torch.randn(b * t, h, w, ci, generator=g).to(DEV).to(torch.bfloat16)
    wt = torch.randn(*((co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)), generator=g).to(DEV) / (3 * (ci * kt) ** 0.5)
    wp = native.conv3x3_prepare_weights(wt)
    x = torch.relu(torch.randn(b * t, h, w, co, generator=g)).to(DEV).to(torch.bfloat16)      # about half zeros, as a ReLU output is
    x[0, 0, 0, :4] = torch.tensor([float('nan'), float('inf'), -1.0, -0.0], dtype=torch.bfloat16)
    frames = t if kt == 3 else 1
    if not native.conv3x3_outmask_supported(h, w, ci, co, kt):          # a layer of the strip kernels (they mask on the input side): refused, not ignored
        assert ci >= 128
        with pytest.raises(native.NativeError):
            native.conv3x3(gy, wp, None, frames, False, out_mask=x)
        return
    plain = native.conv3x3(gy, wp, None, frames, False)
    got = native.conv3x3(gy, wp, None, frames, False, out_mask=x)
    want = torch.ops.aten.threshold_backward(plain, x, 0)
    assert torch.equal(got, want)
    assert float(got[0, 0, 0, 2]) == 0.0 and float(got[0, 0, 0, 3]) == 0.0 and torch.equal(got[0, 0, 0, :2], plain[0, 0, 0, :2])      # NaN <= 0 is false: kept


@pytest.mark.parametrize('ci,co,kt', [(32, 32, 1), (64, 32, 1), (32, 64, 1), (64, 64, 1), (32, 32, 3)])
def test_conv3x3_wgrad_kernel(ci, co, kt):
    """pcacc_conv3x3_wgrad_bf16 against the library's weight gradient on the same bf16 tensors (fp32 sums, different order)."""
    b, t, h, w = 2, 3, 19, 45
    g = torch.Generator(device='cpu').manual_seed(ci + 2 * co + kt)
    x = torch.randn(b * t, h, w, ci, generator=g).to(DEV).to(torch.bfloat16)
    gy = torch.randn(b * t, h, w, co, generator=g).to(DEV).to(torch.bfloat16)
    if kt == 1:
        dw, db = native.conv3x3_wgrad(gy, x)
        dw = dw.view(co, 3, 3, ci).permute(0, 3, 1, 2)
        wr = torch.zeros(co, ci, 3, 3, device=DEV, requires_grad=True)
        F.conv2d(x.float().permute(0, 3, 1, 2), wr, None, padding=1).backward(gy.float().permute(0, 3, 1, 2))
    else:
        parts = [native.conv3x3_wgrad(gy, x, t, dt) for dt in (-1, 0, 1)]
        dw = torch.stack([p[0].view(co, 3, 3, ci) for p in parts], 1).permute(0, 4, 1, 2, 3)
        db = parts[1][1]
        wr = torch.zeros(co, ci, 3, 3, 3, device=DEV, requires_grad=True)
        x5 = x.float().view(b, t, h, w, ci).permute(0, 4, 1, 2, 3)
        F.conv3d(x5, wr, None, padding=1).backward(gy.float().view(b, t, h, w, co).permute(0, 4, 1, 2, 3))
    assert (dw - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item()
    ref_b = gy.float().sum(dim=(0, 1, 2))
    assert (db - ref_b).abs().max().item() <= 1e-3 * max(1.0, ref_b.abs().max().item())


@pytest.mark.parametrize('n,h,w,ci,co', [(2, 18, 18, 256, 128), (1, 36, 36, 128, 256), (2, 9, 72, 128, 128)])
def test_conv3x3_backward_deep(n, h, w, ci, co):
    """Deep layers: forward and data gradient through the strip kernel (csrc/conv_deep.hip; the data gradient is the same kernel on
    mirrored / transposed weights, here with c_in and c_out exchanged), weight gradient as the layer currently takes it."""
    g = torch.Generator(device='cpu').manual_seed(n + ci + co)
    x = torch.randn(n, h, w, ci, generator=g).to(DEV).to(torch.bfloat16).requires_grad_(True)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (4 * ci ** 0.5)).to(DEV).requires_grad_(True)
    bias = torch.randn(co, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(n, h, w, co, generator=g).to(DEV).to(torch.bfloat16)
    assert native.conv3x3_deep_supported(h, w, ci, co) and native.conv3x3_deep_supported(h, w, co, ci)
    y = ops.conv3x3_rows(x, wt, bias, 1, True)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    wr = wt.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    yr = torch.relu(F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1)).permute(0, 2, 3, 1)
    _close(y, yr, yr.abs().max().item())
    (yr * (y.detach() > 0)).backward(gy.float())
    _close(x.grad, xr.grad, xr.grad.abs().max().item())
    assert (wt.grad - wr.grad).abs().max().item() <= 2e-2 * wr.grad.abs().max().item()
    assert (bias.grad - br.grad).abs().max().item() <= 2e-2 * br.grad.abs().max().item()


@pytest.mark.parametrize('n,h,w,ci,co', [(3, 18, 18, 256, 128), (2, 36, 36, 128, 256), (1, 72, 72, 128, 128), (5, 18, 18, 512, 512),
                                         (2, 20, 144, 128, 64), (1, 19, 37, 64, 192), (2, 7, 5, 128, 128),
                                         (2, 24, 288, 128, 64), (1, 5, 301, 64, 128), (1, 3, 600, 128, 64)])
def test_conv3x3_wgrad_deep_kernel(n, h, w, ci, co):
    """pcacc_conv3x3_wgrad_deep_bf16 against the library's weight gradient on the same bf16 tensors (fp32 sums, different order).  The last three
    shapes are wider than one strip's pixel budget: rows cut into 2 / 2 / 3 column segments with their own halos (the ego feature head's 128 -> 64
    layer on 288-wide maps, the last 3x3 weight gradient that went to the convolution library until round 5)."""
    g = torch.Generator(device='cpu').manual_seed(ci + 2 * co + n)
    x = torch.randn(n, h, w, ci, generator=g).to(DEV).to(torch.bfloat16)
    gy = torch.randn(n, h, w, co, generator=g).to(DEV).to(torch.bfloat16)
    assert native.conv3x3_wgrad_deep_supported(h, w, ci, co)
    dw, db = native.conv3x3_wgrad_deep(gy, x)
    dw = dw.view(co, 3, 3, ci).permute(0, 3, 1, 2)
    wr = torch.zeros(co, ci, 3, 3, device=DEV, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), wr, None, padding=1).backward(gy.float().permute(0, 3, 1, 2))
    assert (dw - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item()
    ref_b = gy.float().sum(dim=(0, 1, 2))
    assert (db - ref_b).abs().max().item() <= 1e-3 * max(1.0, ref_b.abs().max().item())


@pytest.mark.parametrize('shape', [(64, 32, 3, 3), (128, 256, 3, 3), (32, 32, 3, 3, 3)])
def test_conv3x3_prepare_weights_pair_and_cache(shape):
    """One launch for both prepared forms, any dense storage order; ops.prepared_conv_weights re-prepares when the weight is written."""
    w = torch.randn(*shape, device=DEV)
    want = native.conv3x3_prepare_weights(w), native.conv3x3_prepare_weights(w, transpose=True)
    for wt in (w, w.contiguous(memory_format=torch.channels_last if w.dim() == 4 else torch.channels_last_3d)):
        fwd, bwd = native.conv3x3_prepare_weights_pair(wt)
        assert torch.equal(fwd, want[0]) and torch.equal(bwd, want[1])
    p = torch.nn.Parameter(w.clone())
    a = ops.prepared_conv_weights(p)
    assert ops.prepared_conv_weights(p)[0] is a[0]                       # cached
    with torch.no_grad():
        p.mul_(2.0)                                                      # what an optimizer step does: a new version
    b = ops.prepared_conv_weights(p)
    assert b[0] is not a[0] and torch.equal(b[0], native.conv3x3_prepare_weights(p.detach().contiguous()))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('n,h,w,c', [(2, 16, 24, 32), (1, 9, 7, 64), (3, 2, 2, 8), (1, 37, 40, 128)])
def test_pool_skip_kernels_bit_exact(n, h, w, c, dtype):
    """csrc/pool.hip against the three library passes it replaces (max_pool2d backward, add, threshold_backward) on the same bf16 / f32
    maps: bit-identical, including ties inside a window (first maximum in scan order), zeros from the ReLU and odd sizes; the f32 pass
    reports the largest magnitude it wrote."""
    g = torch.Generator(device='cpu').manual_seed(h * 100 + w)
    y = torch.relu(torch.randn(n, h, w, c, generator=g)).to(DEV).to(dtype)
    y[:, : h // 2 * 2 : 2, : w // 2 * 2 : 2][..., ::3] = y[:, 1::2, 1::2][:, : h // 2, : w // 2][..., ::3]      # ties between window corners
    pooled = native.maxpool2x2(y)
    ref_pooled, idx = F.max_pool2d(y.permute(0, 3, 1, 2), 2, return_indices=True)
    assert torch.equal(pooled, ref_pooled.permute(0, 2, 3, 1))
    gp = torch.randn(n, h // 2, w // 2, c, generator=g).to(DEV).to(dtype)
    gs = torch.randn(n, h, w, c, generator=g).to(DEV).to(dtype)
    unpooled = torch.ops.aten.max_pool2d_with_indices_backward(gp.permute(0, 3, 1, 2), y.permute(0, 3, 1, 2), [2, 2], [2, 2], [0, 0], [1, 1],
                                                               False, idx).permute(0, 2, 3, 1)
    for a, b, want in ((gp, gs, torch.ops.aten.threshold_backward(unpooled + gs, y, 0)),
                       (gp, None, torch.ops.aten.threshold_backward(unpooled, y, 0)),
                       (None, gs, torch.ops.aten.threshold_backward(gs, y, 0))):
        got = native.pool_skip_relu_backward(y, a, b)
        assert torch.equal(got, want.contiguous())
        if dtype == torch.float32:
            got2, amax = native.pool_skip_relu_backward(y, a, b, want_amax=True)
            assert torch.equal(got2, got) and float(amax.max()) == float(got.abs().max())
        if b is not None:                                      # the skip gradient as a channel slice of a wider map (the decoder's concatenation), read in place
            wide = torch.cat([torch.randn_like(b), b], dim=3)
            assert not wide[..., c:].is_contiguous() and torch.equal(native.pool_skip_relu_backward(y, a, wide[..., c:]), got)


@pytest.mark.parametrize('mode', ['bf16', 'fp32x3'])
def test_down_conv_fused_tail_matches_separate_ops(mode):
    """unet.DownConv with the fused conv + ReLU + pool tail against the same stage written with separate ops: same outputs, same
    gradients of the input and of both convolutions (bf16 rows, and fp32 rows in the fp32x3 mode)."""
    from pcaccumulation_amd.unet import DownConv
    torch.manual_seed(3)
    dt = torch.bfloat16 if mode == 'bf16' else torch.float32
    ops.set_split(mode == 'fp32x3')
    try:
        stage = DownConv(32, 64).to(DEV)
        x = torch.randn(2, 32, 24, 40, device=DEV).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        gp = torch.randn(2, 64, 12, 20, device=DEV).to(dt)
        gs = torch.randn(2, 64, 24, 40, device=DEV).to(dt)
        assert ops.conv3x3_native(x, stage.conv2) == ('bf16' if mode == 'bf16' else 'split')
        calls = []
        orig = native.pool_skip_relu_backward
        native.pool_skip_relu_backward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            pooled, skip = stage(x)
            ((pooled * gp).sum() + (skip * gs).sum()).backward()
        finally:
            native.pool_skip_relu_backward = orig
        assert calls, 'the stage runs the fused tail'
        got = [pooled.detach(), skip.detach(), x.grad.clone()] + [p.grad.clone() for p in stage.parameters()]
        x.grad = None
        stage.zero_grad()
        y = ops.conv3x3(ops.conv3x3(x, stage.conv1, relu=True), stage.conv2, relu=True)
        p2 = F.max_pool2d(y, 2)
        ((p2 * gp).sum() + (y * gs).sum()).backward()
        want = [p2.detach(), y.detach(), x.grad.clone()] + [p.grad.clone() for p in stage.parameters()]
    finally:
        ops.set_split(False)
    tol = 1e-3 if mode == 'bf16' else 1e-5
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert torch.allclose(a.float(), b.float(), rtol=tol, atol=tol * float(b.float().abs().max())), float((a.float() - b.float()).abs().max())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,h,w,ci,co', [(2, 37, 53, 32, 2), (1, 16, 16, 64, 2), (3, 9, 70, 32, 4), (1, 1, 1, 32, 1), (1, 288, 288, 32, 2)])
def test_head_conv3x3(n, h, w, ci, co, dtype):
    """Conv2d(c_in, c_out <= 4, 3, padding 1) -- the fg / bg head's last layer (models/unet.py:259-277) -- on the streamed fp32 kernels of
    csrc/head_conv.hip: logits, input gradient, weight and bias gradients against the same layer in float64 on the same (f32 / bf16) input."""
    g = torch.Generator(device='cpu').manual_seed(h + w + ci + co)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(n, ci, h, w, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(n, co, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    assert ops.head_conv3x3_available(x, conv)
    y = ops.conv3x3(x, conv)
    assert y.dtype == torch.float32 and y.shape == (n, co, h, w)
    y.backward(gy)
    c64 = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV).double()
    c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    xr = x.detach().double().requires_grad_(True)
    yr = c64(xr)
    yr.backward(gy.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(y, yr.detach()) <= 2e-6
    assert rel(x.grad, xr.grad) <= (2e-6 if dtype == torch.float32 else 8e-3)            # bf16 gradient: rounded once on store
    assert rel(conv.weight.grad, c64.weight.grad) <= 2e-5 and rel(conv.bias.grad, c64.bias.grad) <= 2e-5      # fp32 sums over n*h*w pixels, atomics
