"""csrc/upconv_bf16.hip: nn.ConvTranspose2d(kernel 2, stride 2) (models/unet.py:22-30) on bf16 channels-last rows -- forward, data gradient (also from a
channel slice of a wider map: the decoder's concatenation gradient), weight / bias gradient -- against the library's fp32 transposed convolution on the
same bf16-rounded operands.  Products of bf16 values are exact in fp32, so the only differences are the fp32 summation order and the bf16 rounding of
the stored result."""
import pytest
import torch
import torch.nn.functional as F

from pcaccumulation_amd import native, ops

pytestmark = pytest.mark.gpu

SHAPES = [(3, 5, 7, 64, 32), (2, 9, 9, 128, 64), (1, 18, 18, 256, 128), (2, 4, 6, 512, 256), (2, 11, 13, 64, 64), (1, 40, 36, 64, 32)]


def _case(n, h, w, c_in, c_up, seed=0, channels_last_weight=False):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(seed + c_in + h)
    x = torch.randn(n, h, w, c_in, generator=g).to(dev).to(torch.bfloat16)
    wt = (torch.randn(c_in, c_up, 2, 2, generator=g) / c_in ** 0.5).to(dev)
    if channels_last_weight:
        wt = wt.contiguous(memory_format=torch.channels_last)
    b = torch.randn(c_up, generator=g).to(dev)
    dy = torch.randn(n, 2 * h, 2 * w, c_up, generator=g).to(dev).to(torch.bfloat16)
    return x, wt, b, dy


def _w16(wt):
    return wt.to(torch.bfloat16).float()


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('cl', [False, True])
def test_forward(shape, cl):
    x, wt, b, _ = _case(*shape, channels_last_weight=cl)
    fwd, _ = native.upconv2x2_bf16_prepare_weights(wt)
    y = native.upconv2x2_bf16(x, fwd, b, 0)
    ref = F.conv_transpose2d(x.float().permute(0, 3, 1, 2), _w16(wt), b, stride=2).permute(0, 2, 3, 1)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    err = (y.float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2 ** -8, float(err)                      # one bf16 rounding of the stored value
    # and exactly the rounding of a result that differs from the library's by fp32 summation order only
    assert float((y.float() - ref.to(torch.bfloat16).float()).abs().max() / ref.abs().max()) < 2 ** -7


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('sliced', [False, True])
def test_data_gradient(shape, sliced):
    n, h, w, c_in, c_up = shape
    x, wt, b, dy = _case(*shape)
    if sliced:                                                  # dy = the first c_up channels of a [.., 2 c_up] map, read in place
        wide = torch.randn(n, 2 * h, 2 * w, 2 * c_up, device=dy.device).to(torch.bfloat16)
        wide[..., :c_up] = dy
        dy_in = wide[..., :c_up]
        assert not dy_in.is_contiguous()
    else:
        dy_in = dy
    _, bwd = native.upconv2x2_bf16_prepare_weights(wt)
    gx = native.upconv2x2_bf16(dy_in, bwd, None, 1)
    ref = F.conv2d(dy.float().permute(0, 3, 1, 2), _w16(wt), None, stride=2).permute(0, 2, 3, 1)      # adjoint of the transposed convolution
    assert gx.shape == (n, h, w, c_in)
    assert float((gx.float() - ref).abs().max() / ref.abs().max()) < 2 ** -8


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('sliced,cl', [(False, False), (True, True)])
def test_weight_and_bias_gradient(shape, sliced, cl):
    n, h, w, c_in, c_up = shape
    x, wt, b, dy = _case(*shape, channels_last_weight=cl)
    dy_in = dy
    if sliced:
        wide = torch.randn(n, 2 * h, 2 * w, 2 * c_up, device=dy.device).to(torch.bfloat16)
        wide[..., c_up:] = dy
        dy_in = wide[..., c_up:]                                # the second half: a non-zero channel offset
    gw, gb = native.upconv2x2_bf16_wgrad(dy_in, x, want_bias=True, like=wt)
    xf = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    wref = wt.detach().clone().contiguous().requires_grad_(True)
    bref = b.clone().requires_grad_(True)
    y = F.conv_transpose2d(xf, wref, bref, stride=2)
    y.backward(dy.float().permute(0, 3, 1, 2))
    assert gw.stride() == wt.stride() and gw.dtype == torch.float32
    assert float((gw - wref.grad).abs().max() / wref.grad.abs().max()) < 1e-5
    assert float((gb - bref.grad).abs().max() / bref.grad.abs().max()) < 1e-5


def test_autograd_path_matches_the_module():
    """ops.upconv2x2 on bf16 channels-last maps (bf16 compute mode) == the module in fp32 up to bf16 rounding, gradients included; the layer takes the
    own kernels (no library call)."""
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    conv = torch.nn.ConvTranspose2d(128, 64, 2, 2).to(dev)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(2, 128, 12, 10, device=dev).contiguous(memory_format=torch.channels_last)
    x16 = x.to(torch.bfloat16).requires_grad_(True)
    y = ops.upconv2x2(x16, conv)
    assert y.dtype == torch.bfloat16 and type(y.grad_fn).__name__ == 'PermuteBackward0'
    g = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (x16, conv.weight, conv.bias), g)
    xr = x16.detach().float().requires_grad_(True)
    yr = conv(xr)
    rx, rw, rb = torch.autograd.grad(yr, (xr, conv.weight, conv.bias), g.float())
    rel = lambda a, r: float((a.float() - r).abs().max() / r.abs().max())
    assert rel(y, yr) < 1e-2 and rel(gx, rx) < 1e-2 and rel(gw, rw) < 1e-2 and rel(gb, rb) < 1e-3
    assert gw.stride() == conv.weight.stride()


def test_unsupported_channel_counts_are_refused():
    assert not native.upconv2x2_bf16_supported(32, 16) and not native.upconv2x2_bf16_supported(96, 48) and native.upconv2x2_bf16_supported(64, 32)
