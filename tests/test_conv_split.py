"""fp32x3 convolution (include/pcacc.h "A6/A9 at fp32 accuracy", csrc/conv_split.hip): fp32 rows in / out, products formed on the 16-bit
matrix cores from fp16 hi / lo halves of power-of-two scaled operands.  Reference: the same convolution in float64.  Tolerance: hi + lo
keep 22 significant bits per factor (fp32: 24), the accumulation is fp32 on both sides -- asserted at 3e-6 of the largest output
(bf16 operands: 4e-3; the bf16 hi / lo split this file started with: 4e-6 per layer) and within 10x of what the fp32 library
convolution itself shows against float64 (so the bound is not vacuous).  Dynamic range: tensors of magnitude 1e-8 .. 1e+8, a single
outlier 1e4 x the rest, non-finite values."""
import pytest
import torch
import torch.nn.functional as F

from pcaccumulation_amd import native, ops

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
TOL = 3e-6


@pytest.fixture(autouse=True)
def _split_mode():
    ops.set_split(True)
    yield
    ops.set_split(False)


def _ref64(x_rows, w, b, relu, frames=1):
    x = x_rows.double()
    if w.dim() == 5:
        n, h, ww, ci = x.shape
        x5 = x.view(n // frames, frames, h, ww, ci).permute(0, 4, 1, 2, 3)
        y = F.conv3d(x5, w.double(), b.double() if b is not None else None, padding=1).permute(0, 2, 3, 4, 1).reshape(n, h, ww, -1)
    else:
        y = F.conv2d(x.permute(0, 3, 1, 2), w.double(), b.double() if b is not None else None, padding=1).permute(0, 2, 3, 1)
    return torch.relu(y) if relu else y


def _rel(a, ref):
    return ((a.double() - ref).abs().max() / ref.abs().max()).item()


SHAPES = [
    # (n, h, w, c_in, c_out, relu): every (NW, NGW) shape of the kernel, both slice widths, bands and whole rows, ragged tiles
    (2, 16, 32, 32, 32, True), (1, 19, 45, 64, 32, False), (3, 8, 33, 32, 64, True), (1, 24, 64, 96, 32, True),
    (1, 9, 18, 128, 128, True), (1, 18, 18, 256, 128, False), (1, 7, 5, 128, 256, True), (2, 40, 40, 64, 64, True),
    (3, 18, 18, 512, 512, True), (2, 36, 36, 256, 256, True), (1, 72, 72, 128, 128, False), (2, 36, 36, 512, 256, True),
    (1, 18, 18, 256, 512, False), (2, 20, 144, 128, 64, True), (1, 5, 288, 128, 64, True), (1, 19, 37, 128, 192, False),
    (1, 72, 72, 256, 128, True), (1, 3, 400, 128, 128, True), (1, 288, 288, 32, 32, True), (1, 144, 144, 64, 64, True),
    (1, 1, 1, 32, 32, False), (1, 2, 700, 64, 96, True)]


@pytest.mark.parametrize('n,h,w,ci,co,relu', SHAPES)
def test_conv3x3_split_forward(n, h, w, ci, co, relu):
    g = torch.Generator(device='cpu').manual_seed(n * 1000 + ci + co)
    x = torch.randn(n, h, w, ci, generator=g).to(DEV)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    assert native.conv3x3_split_supported(h, w, ci, co)
    y = ops.conv3x3_rows(x, wt, b, 1, relu)
    ref = _ref64(x, wt, b, relu)
    assert y.dtype == torch.float32 and y.shape == ref.shape
    err = _rel(y, ref)
    lib = _rel(torch.relu(F.conv2d(x.permute(0, 3, 1, 2), wt, b, padding=1)).permute(0, 2, 3, 1) if relu
               else F.conv2d(x.permute(0, 3, 1, 2), wt, b, padding=1).permute(0, 2, 3, 1), ref)
    assert err <= TOL, (err, lib)
    assert err <= max(10 * lib, 2e-6), (err, lib)                # fp32-like: within 10x of the fp32 library's own deviation from float64


def test_conv3x3_split_identity_weights_asymmetric():
    """Centre-tap / shifted permutation weights: the output must reproduce the input (hi + lo) to fp32 rounding -- catches operand swaps
    and a dropped lo plane (which would leave a 2^-11 relative error)."""
    ci = co = 64
    x = torch.randn(1, 12, 40, ci).to(DEV)
    perm = torch.randperm(ci)
    wt = torch.zeros(co, ci, 3, 3)
    wt[torch.arange(co), perm, 1, 1] = 1.0
    y = ops.conv3x3_rows(x, wt.to(DEV), None, 1, False)
    ref = x[..., perm.to(DEV)]
    assert (y - ref).abs().max().item() <= 2 ** -21 * ref.abs().max().item()
    wt = torch.zeros(co, ci, 3, 3)
    wt[torch.arange(co), perm, 0, 2] = 1.0           # tap (dy=-1, dx=+1): shifted copy with zero border
    y = ops.conv3x3_rows(x, wt.to(DEV), None, 1, False)
    ref = torch.zeros_like(x)
    ref[:, 1:, :-1] = x[:, :-1, 1:][..., perm.to(DEV)]
    assert (y - ref).abs().max().item() <= 2 ** -21 * x.abs().max().item()
    assert torch.equal(y[:, 0], torch.zeros_like(y[:, 0])) and torch.equal(y[:, :, -1], torch.zeros_like(y[:, :, -1]))


@pytest.mark.parametrize('b,t,h,w,ci,co', [(2, 5, 16, 40, 32, 32), (1, 3, 9, 33, 32, 64), (1, 1, 8, 8, 32, 32), (1, 2, 36, 36, 64, 128)])
def test_conv3x3x3_split_forward(b, t, h, w, ci, co):
    g = torch.Generator(device='cpu').manual_seed(7)
    x = torch.randn(b * t, h, w, ci, generator=g).to(DEV)
    wt = (torch.randn(co, ci, 3, 3, 3, generator=g) / (5 * ci ** 0.5)).to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    y = ops.conv3x3_rows(x, wt, bias, t, True)
    ref = _ref64(x, wt, bias, True, frames=t)
    assert _rel(y, ref) <= TOL


@pytest.mark.parametrize('kt,ci,co,h,w', [(1, 32, 64, 12, 36), (3, 32, 32, 12, 36), (1, 64, 64, 19, 45), (1, 32, 32, 33, 70), (1, 128, 64, 18, 18),
                                          (1, 256, 128, 9, 20), (3, 32, 64, 7, 33), (1, 64, 32, 40, 40)])
def test_conv3x3_split_backward(kt, ci, co, h, w):
    b, t = 2, 3
    g = torch.Generator(device='cpu').manual_seed(11 + ci + co)
    x = torch.randn(b * t, h, w, ci, generator=g).to(DEV).requires_grad_(True)
    shape = (co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)
    wt = (torch.randn(*shape, generator=g) / (4 * ci ** 0.5)).to(DEV).requires_grad_(True)
    bias = torch.randn(co, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(b * t, h, w, co, generator=g).to(DEV)
    y = ops.conv3x3_rows(x, wt, bias, t if kt == 3 else 1, True)
    y.backward(gy)
    xr = x.detach().double().requires_grad_(True)
    wr = wt.detach().double().requires_grad_(True)
    br = bias.detach().double().requires_grad_(True)
    if kt == 3:
        x5 = xr.view(b, t, h, w, ci).permute(0, 4, 1, 2, 3)
        yr = F.conv3d(x5, wr, br, padding=1).permute(0, 2, 3, 4, 1).reshape(b * t, h, w, co)
    else:
        yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1).permute(0, 2, 3, 1)
    # the kernel's own ReLU mask: outputs within rounding of zero may differ between the two
    (yr * (y.detach() > 0)).backward(gy.double())
    assert _rel(x.grad, xr.grad) <= TOL
    assert _rel(wt.grad, wr.grad) <= TOL
    assert _rel(bias.grad, br.grad) <= TOL


def test_conv3x3_split_module_dispatch_and_mode_switch():
    conv = torch.nn.Conv2d(32, 64, 3, padding=1).to(DEV)
    x = torch.randn(2, 32, 16, 32, device=DEV).to(memory_format=torch.channels_last)
    assert ops.conv3x3_native(x, conv) == 'split'
    y = ops.conv3x3(x, conv, relu=True)
    ref = torch.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1))
    assert y.dtype == torch.float32 and _rel(y, ref) <= TOL
    ops.set_split(False)
    assert ops.conv3x3_native(x, conv) is None                    # plain fp32 mode: the library
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert ops.conv3x3_native(x, conv) == 'bf16'
    with pytest.raises(native.NativeError):
        native.conv3x3_split(torch.zeros(1, 8, 8, 32), (torch.zeros(2, 9, 32, 32, dtype=torch.float16), torch.zeros(32)), None, 1, False)


def test_conv3x3_split_weights_follow_parameter_updates():
    """The prepared hi / lo planes are cached per weight version: an optimizer step must invalidate them."""
    conv = torch.nn.Conv2d(32, 32, 3, padding=1).to(DEV)
    x = torch.randn(1, 32, 8, 32, device=DEV).to(memory_format=torch.channels_last)
    y0 = ops.conv3x3(x, conv)
    with torch.no_grad():
        conv.weight.mul_(2.0)
    y1 = ops.conv3x3(x, conv)
    ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    assert _rel(y1, ref) <= TOL and not torch.allclose(y0, y1)


def test_absmax256():
    for n in (1, 3, 4, 1000, 12345, 1 << 20):
        x = torch.randn(n, device=DEV) * 3
        assert float(native.absmax256(x).max()) == float(x.abs().max())
    x = torch.zeros(4096, device=DEV)
    assert float(native.absmax256(x).max()) == 0.0
    x[77] = float('nan')
    assert float(native.absmax256(x).max()) == float('inf')        # a NaN poisons the maximum (fmaxf alone would drop it)
    x[77] = -float('inf')
    assert float(native.absmax256(x).max()) == float('inf')
    y = torch.randn(64, 64, device=DEV)
    y[0, 63] = 100.0
    assert float(native.absmax256(y[:, :32]).max()) == float(y[:, :32].abs().max())     # a strided view is measured on its own elements


@pytest.mark.parametrize('scale_x,scale_w', [(1e-8, 1.0), (1e8, 1e-6), (1.0, 1e5), (3e-5, 7e3)])
def test_conv3x3_split_dynamic_range(scale_x, scale_w):
    """Gradient-sized and huge operands: the per-tensor / per-row power-of-two scales keep the relative accuracy."""
    g = torch.Generator(device='cpu').manual_seed(3)
    x = (torch.randn(2, 20, 40, 64, generator=g) * scale_x).to(DEV).requires_grad_(True)
    wt = (torch.randn(64, 64, 3, 3, generator=g) / 24 * scale_w).to(DEV).requires_grad_(True)
    gy = (torch.randn(2, 20, 40, 64, generator=g) * scale_x).to(DEV)
    y = ops.conv3x3_rows(x, wt, None, 1, False)
    y.backward(gy)
    xr, wr = x.detach().double().requires_grad_(True), wt.detach().double().requires_grad_(True)
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, None, padding=1).permute(0, 2, 3, 1)
    yr.backward(gy.double())
    assert _rel(y, yr.detach()) <= TOL and _rel(x.grad, xr.grad) <= TOL and _rel(wt.grad, wr.grad) <= TOL


def test_conv3x3_split_outlier_and_nonfinite():
    g = torch.Generator(device='cpu').manual_seed(4)
    x = torch.randn(1, 16, 32, 32, generator=g).to(DEV)
    x[0, 5, 7, 3] = 1e4                                           # one element 1e4 x the rest: the others keep >= 2^-22 * 1e4 absolute accuracy
    wt = (torch.randn(32, 32, 3, 3, generator=g) / 17).to(DEV)
    y = ops.conv3x3_rows(x, wt, None, 1, False)
    ref = _ref64(x, wt, None, False)
    far = torch.ones_like(ref, dtype=torch.bool)
    far[0, 3:8, 5:10] = False                                     # outputs the outlier does not reach: errors relative to THEIR size
    assert _rel(y, ref) <= TOL
    assert float((y.double() - ref)[far].abs().max()) <= 2 ** -22 * 1e4 * 9 * 32 * float(wt.abs().max())
    x[0, 2, 2, 0] = float('nan')
    y = ops.conv3x3_rows(x, wt, None, 1, False)
    assert not torch.isfinite(y).all()                            # a non-finite input reaches the output, as in fp32 arithmetic
    x[0, 2, 2, 0] = float('inf')
    assert not torch.isfinite(ops.conv3x3_rows(x, wt, None, 1, False)).all()
    z = torch.zeros(1, 8, 8, 32, device=DEV)
    assert torch.equal(ops.conv3x3_rows(z, wt, None, 1, False), torch.zeros(1, 8, 8, 32, device=DEV))     # all-zero tensor: scale 1


def test_split_kernels_report_their_output_maximum_and_tags_follow_the_tensor():
    x = torch.randn(2, 24, 40, 32, device=DEV)
    wt = torch.randn(64, 32, 3, 3, device=DEV) / 17
    wps = native.conv3x3_split_prepare_weights(wt)[0]
    y, y_amax = native.conv3x3_split(x, wps, None, 1, True, want_amax=True)
    assert float(y_amax.max()) == float(y.abs().max())
    conv = torch.nn.Conv2d(32, 32, 3, padding=1).to(DEV)
    xn = x.permute(0, 3, 1, 2)
    a = ops.conv3x3(xn, conv, relu=True)
    assert ops.amax_tag(xn) is not None and float(ops.amax_tag(xn).max()) == float(x.abs().max())       # measured once, remembered on the tensor
    assert ops.amax_tag(a) is not None and float(ops.amax_tag(a).max()) == float(a.abs().max())          # produced by the kernel, carried through the permute
    a.add_(1.0)
    assert ops.amax_tag(a) is None                                                                        # an in-place write invalidates the tag


@pytest.mark.parametrize('n,t,h,w,ci,co,kt', [(2, 1, 16, 32, 32, 32, 1), (1, 1, 19, 45, 64, 32, 1), (3, 1, 8, 33, 32, 64, 1), (2, 1, 40, 40, 64, 64, 1),
                                              (6, 3, 12, 36, 32, 32, 3), (4, 2, 33, 70, 32, 64, 3), (2, 1, 100, 70, 32, 32, 1), (5, 5, 9, 9, 64, 64, 3),
                                              (1, 1, 1, 1, 32, 32, 1), (2, 2, 3, 300, 64, 32, 3)])
def test_conv3x3_split_resident_kernel(n, t, h, w, ci, co, kt, monkeypatch):
    """The persistent kernel of the narrow layers (weights of a slice resident in LDS, patches prefetched across tiles), forced on at
    test sizes: forward and masked data gradient against float64, and bit-equal maxima reporting."""
    monkeypatch.setenv('PCACC_CONV_RES', '2')
    native.reload_switches()                                   # the launchers read their switches once per process
    g = torch.Generator(device='cpu').manual_seed(n + h + w + ci + co)
    x = torch.randn(n, h, w, ci, generator=g).to(DEV).requires_grad_(True)
    shape = (co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)
    wt = (torch.randn(*shape, generator=g) / (4 * ci ** 0.5)).to(DEV).requires_grad_(True)
    bias = torch.randn(co, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(n, h, w, co, generator=g).to(DEV)
    y = ops.conv3x3_rows(x, wt, bias, t if kt == 3 else 1, True)
    y.backward(gy)
    xr, wr, br = (v.detach().double().requires_grad_(True) for v in (x, wt, bias))
    if kt == 3:
        yr = F.conv3d(xr.view(n // t, t, h, w, ci).permute(0, 4, 1, 2, 3), wr, br, padding=1).permute(0, 2, 3, 4, 1).reshape(n, h, w, co)
    else:
        yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1).permute(0, 2, 3, 1)
    (yr * (y.detach() > 0)).backward(gy.double())
    assert _rel(y, torch.relu(yr.detach())) <= TOL and _rel(x.grad, xr.grad) <= TOL and _rel(wt.grad, wr.grad) <= TOL
    assert float(ops.amax_tag(y).max()) == float(y.abs().max())
    monkeypatch.setenv('PCACC_CONV_RES', '0')
    native.reload_switches()
    y0 = ops.conv3x3_rows(x.detach(), wt.detach(), bias.detach(), t if kt == 3 else 1, True)
    assert _rel(y0, y.detach()) <= 1e-6                        # the two kernels sum in different orders


@pytest.mark.parametrize('n,h,w,ci,cu', [(2, 18, 18, 512, 256), (1, 9, 20, 64, 32), (3, 36, 36, 256, 128), (1, 72, 72, 128, 64), (2, 33, 7, 64, 64),
                                         (1, 144, 144, 64, 32), (1, 5, 5, 128, 128)])
def test_upconv2x2_split(n, h, w, ci, cu):
    """nn.ConvTranspose2d(kernel 2, stride 2) (models/unet.py:22-30) in the fp32x3 mode: forward, data gradient, weight and bias
    gradients against the module in float64."""
    g = torch.Generator(device='cpu').manual_seed(ci + cu + h)
    conv = torch.nn.ConvTranspose2d(ci, cu, kernel_size=2, stride=2).to(DEV)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(n, ci, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(n, cu, 2 * h, 2 * w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    calls = []
    orig = native.upconv2x2_split
    native.upconv2x2_split = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        y = ops.upconv2x2(x, conv)
        y.backward(gy)
    finally:
        native.upconv2x2_split = orig
    assert len(calls) == 2 and y.shape == (n, cu, 2 * h, 2 * w)
    c64 = torch.nn.ConvTranspose2d(ci, cu, kernel_size=2, stride=2).to(DEV).double()
    c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    xr = x.detach().double().requires_grad_(True)
    yr = c64(xr)
    yr.backward(gy.double())
    assert _rel(y, yr.detach()) <= TOL and _rel(x.grad, xr.grad) <= TOL
    assert _rel(conv.weight.grad, c64.weight.grad) <= TOL and _rel(conv.bias.grad, c64.bias.grad) <= TOL
    assert float(ops.amax_tag(y).max()) == float(y.abs().max())


@pytest.mark.parametrize('ci,co,kt,h,w,res', [(32, 32, 1, 24, 40, '2'), (32, 32, 1, 24, 40, '0'), (64, 32, 1, 17, 33, '2'), (32, 64, 1, 20, 36, '0'),
                                              (32, 32, 3, 16, 40, '2'), (128, 64, 1, 18, 18, '0')])
def test_conv_split_out_mask(ci, co, kt, h, w, res, monkeypatch):
    """pcacc_conv3x3_split_outmask: the data gradient stored as zero where the mask map is <= 0 == the plain kernel followed by
    aten::threshold_backward, bit for bit (streaming and resident kernels, with and without the input-side mask); the reported maximum
    is that of the masked result."""
    monkeypatch.setenv('PCACC_CONV_RES', res)
    native.reload_switches()
    b, t = 2, 3
    g = torch.Generator(device='cpu').manual_seed(ci + 5 * co + kt)
    gy = torch.randn(b * t, h, w, ci, generator=g).to(DEV)
    y = torch.relu(torch.randn(b * t, h, w, ci, generator=g)).to(DEV)
    wt = torch.randn(*((co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)), generator=g).to(DEV) / (3 * (ci * kt) ** 0.5)
    wf, _ = native.conv3x3_split_prepare_weights(wt)
    x = torch.relu(torch.randn(b * t, h, w, co, generator=g)).to(DEV)
    x[0, 0, 0, :3] = torch.tensor([float('nan'), -1.0, -0.0])
    frames = t if kt == 3 else 1
    am = native.absmax256(gy)
    for mask in (None, y):
        plain = native.conv3x3_split(gy, wf, None, frames, False, mask=mask, amax=am)
        got, got_amax = native.conv3x3_split(gy, wf, None, frames, False, mask=mask, amax=am, want_amax=True, out_mask=x)
        want = torch.ops.aten.threshold_backward(plain, x, 0)
        assert torch.equal(got, want)
        assert float(got_amax.max()) == float(want[torch.isfinite(want)].abs().max()) or not torch.isfinite(want).all()


@pytest.mark.parametrize('n,h,w,ca,cb,co,res', [(2, 40, 36, 32, 32, 32, '2'), (2, 40, 36, 32, 32, 32, '0'), (1, 72, 72, 128, 128, 128, '1'),
                                                (2, 36, 36, 256, 256, 256, '1'), (2, 64, 64, 64, 64, 64, '2'), (1, 33, 45, 32, 64, 64, '1'),
                                                (1, 20, 28, 64, 32, 32, '2'), (3, 18, 18, 96, 32, 64, '1')])
@pytest.mark.parametrize('relu', [False, True])
def test_conv3x3_split_on_two_inputs(n, h, w, ca, cb, co, res, relu, monkeypatch):
    """pcacc_conv3x3_split_cat: the convolution of cat(a, b) read in place == the float64 convolution of the concatenation at the layer tolerance,
    == the one-input kernel on the materialised concatenation up to the summation order (the channel boundary may force narrower slices), bf16
    second output = rounding of the first, maxima = those of the result.  res: PCACC_CONV_RES ('0' never / '2' always the resident kernel)."""
    monkeypatch.setenv('PCACC_CONV_RES', res)
    native.reload_switches()
    try:
        g = torch.Generator().manual_seed(ca + cb + h)
        a = torch.randn(n, h, w, ca, generator=g).to(DEV)
        b = (3.0 * torch.randn(n, h, w, cb, generator=g)).to(DEV)
        wt = (torch.randn(co, ca + cb, 3, 3, generator=g) / (3 * (ca + cb) ** 0.5)).to(DEV)
        bias = torch.randn(co, generator=g).to(DEV)
        wps = native.conv3x3_split_prepare_weights(wt)[0]
        cat = torch.cat((a, b), -1)
        amax = torch.maximum(native.absmax256(a), native.absmax256(b))
        y, ya, y16 = native.conv3x3_split_cat(a, b, amax, wps, bias, relu, want_bf16=True)
        ref = _ref64(cat, wt, bias, relu)
        assert _rel(y, ref) < TOL
        one = native.conv3x3_split(cat, wps, bias, 1, relu, amax=amax)
        assert _rel(y, one.double()) < 1e-6
        assert torch.equal(y16, y.to(torch.bfloat16))
        assert float(ya.max()) == float(y.abs().max())
        y2, _ = native.conv3x3_split_cat(a, b, amax, wps, bias, relu)
        assert torch.equal(y2, y)
    finally:
        monkeypatch.delenv('PCACC_CONV_RES')
        native.reload_switches()


def test_xcd_contiguous_block_walk_changes_no_result(monkeypatch):
    """Round 5: the multi-group convolution kernels map blockIdx to a logical block id such that the output-channel groups of a tile / the (co, ci)
    blocks of a strip run on ONE XCD and share their operand reads in its L2 (common.h: pcacc_xcd_block).  A pure re-ordering of workgroups: forward
    results, weight gradients (same partial slots, same reduce order) and maxima are bit-identical to launch order (PCACC_XCD_REMAP=0), fp32x3 and bf16."""
    g = torch.Generator(device='cpu').manual_seed(5)
    got = {}
    try:
        for remap in ('1', '0'):
            monkeypatch.setenv('PCACC_XCD_REMAP', remap)
            native.reload_switches()
            out = []
            for n, h, w, ci, co in ((3, 36, 36, 128, 256), (2, 18, 18, 256, 512), (5, 19, 23, 64, 128)):
                gg = torch.Generator(device='cpu').manual_seed(n + h + ci)
                x = torch.randn(n, h, w, ci, generator=gg).to(DEV)
                gy = torch.randn(n, h, w, co, generator=gg).to(DEV)
                wt = (torch.randn(co, ci, 3, 3, generator=gg) / (3 * ci ** 0.5)).to(DEV)
                bias = torch.randn(co, generator=gg).to(DEV)
                wf, wb = native.conv3x3_split_prepare_weights(wt)
                y, ya = native.conv3x3_split(x, wf, bias, 1, True, want_amax=True)
                out += [y, ya.max(), native.conv3x3_split(gy, wb, None, 1, False, mask=y)]
                out += list(native.conv3x3_wgrad_split(gy, x, mask=y))
                xb, gb = x.to(torch.bfloat16), gy.to(torch.bfloat16)
                if native.conv3x3_deep_supported(h, w, ci, co):
                    out.append(native.conv3x3(xb, native.conv3x3_prepare_weights(wt), bias, 1, True))     # the strip kernel of conv_deep.hip (c_in >= 128)
                if native.conv3x3_wgrad_deep_supported(h, w, ci, co):
                    out += list(native.conv3x3_wgrad_deep(gb, xb))
            got[remap] = out
    finally:
        monkeypatch.delenv('PCACC_XCD_REMAP')
        native.reload_switches()
    assert len(got['1']) == len(got['0']) and len(got['1']) >= 20
    for a, b in zip(got['1'], got['0']):
        assert torch.equal(a, b)
