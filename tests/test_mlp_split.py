"""fp32x3 per-point linear layers (include/pcacc.h pcacc_rows_linear_split / _cat_split / pcacc_rows_wgrad_split / _cat_split,
csrc/mlp_split.hip): fp32 rows, products on the 16-bit matrix cores from scaled fp16 hi / lo halves.  Reference: the same layer in
float64 through autograd.  Tolerance 3e-6 of the largest entry (22 significant bits per factor, fp32 accumulation) -- the bf16 row
kernels are compared at 1e-2."""
import pytest
import torch

from pcaccumulation_amd import native, ops
from pcaccumulation_amd.ops import PillarIndex

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
TOL = 3e-6


@pytest.fixture(autouse=True)
def _split_mode():
    ops.set_split(True)
    yield
    ops.set_split(False)


def _rel(a, ref):
    return ((a.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-300)).item()


@pytest.mark.parametrize('k,n,rows', [(32, 32, 5000), (32, 64, 4097), (64, 32, 128), (64, 64, 33333), (128, 128, 9001), (32, 128, 2500),
                                      (128, 64, 6000), (64, 128, 2049), (128, 32, 3000)])
@pytest.mark.parametrize('pre_relu,post_relu,with_res', [(False, False, False), (True, True, True), (True, False, True)])
def test_linear_rows_split_autograd(k, n, rows, pre_relu, post_relu, with_res):
    g = torch.Generator(device='cpu').manual_seed(k * 7 + n + rows)
    layer = torch.nn.Linear(k, n).to(DEV)
    x = torch.randn(rows, k, generator=g).to(DEV).requires_grad_(True)
    res = torch.randn(rows, n, generator=g).to(DEV).requires_grad_(True) if with_res else None
    gy = torch.randn(rows, n, generator=g).to(DEV)
    y = ops.linear_rows(x, layer, pre_relu=pre_relu, post_relu=post_relu, residual=res)
    assert y.dtype == torch.float32
    y.backward(gy)
    xr = x.detach().double().requires_grad_(True)
    wr, br = layer.weight.detach().double().requires_grad_(True), layer.bias.detach().double().requires_grad_(True)
    rr = res.detach().double().requires_grad_(True) if with_res else None
    h = torch.relu(xr) if pre_relu else xr
    yr = torch.nn.functional.linear(h, wr, br)
    if with_res:
        yr = yr + rr
    # the kernel's own ReLU decisions: outputs within rounding of zero may differ between the two
    yr_out = yr * (y.detach() > 0) if post_relu else yr
    yr_out.backward(gy.double())
    assert _rel(y, torch.relu(yr.detach()) if post_relu else yr.detach()) <= TOL
    assert _rel(x.grad, xr.grad) <= TOL
    assert _rel(layer.weight.grad, wr.grad) <= TOL and _rel(layer.bias.grad, br.grad) <= TOL
    if with_res:
        assert _rel(res.grad, rr.grad) <= TOL


def test_linear_rows_split_dispatch():
    layer = torch.nn.Linear(64, 32).to(DEV)
    x = torch.randn(4096, 64, device=DEV)
    calls = []
    orig = native.rows_linear_split
    native.rows_linear_split = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        ops.linear_rows(x, layer)
        assert calls, 'fp32x3 mode: the split kernel takes 64 -> 32'
        calls.clear()
        ops.linear_rows(torch.randn(4096, 9, device=DEV), torch.nn.Linear(9, 64).to(DEV))       # few input features: the fp32 streaming kernel
        ops.set_split(False)
        ops.linear_rows(x, layer)                                                                  # fp32 mode: the fp32 vector kernel
        assert not calls
    finally:
        native.rows_linear_split = orig


def test_pfn_block_on_two_piece_fp32_rows():
    """ResnetBlockFC.forward_pooled in the fp32x3 mode (gather + concatenation folded into the split row kernels,
    pcacc_rows_linear_cat_split / pcacc_rows_wgrad_cat_split) against the block on the materialised cat(x, pooled[p2v]) in float64."""
    from pcaccumulation_amd.pillar_encoder import ResnetBlockFC
    torch.manual_seed(1)
    n, m = 50_000, 9_000
    p2v = torch.randint(0, m, (n,), device=DEV, dtype=torch.int32)
    p2v[:m] = torch.arange(m, device=DEV, dtype=torch.int32)
    pidx = PillarIndex.from_point_map(p2v, m)
    block = ResnetBlockFC(64, 32).to(DEV)
    torch.nn.init.normal_(block.fc_1.weight, std=0.2)
    x = torch.randn(n, 32, device=DEV).requires_grad_(True)
    pooled = (torch.randn(m, 32, device=DEV) * 40).requires_grad_(True)            # the two pieces differ in magnitude: one scale for both
    g = torch.randn(n, 32, device=DEV)
    assert ops.linear_rows_cat_available(x, pooled, block.fc_0)
    y = block.forward_pooled(x, pooled, pidx)
    y.backward(g)
    got = (y.detach(), x.grad.clone(), pooled.grad.clone(), [p.grad.clone() for p in block.parameters()])
    b64 = ResnetBlockFC(64, 32).to(DEV).double()
    b64.load_state_dict({k: v.double() for k, v in block.state_dict().items()})
    xr, pr = x.detach().double().requires_grad_(True), pooled.detach().double().requires_grad_(True)
    ops.set_split(False)
    cat = torch.cat([xr, pr[p2v.long()]], dim=1)
    net = torch.nn.functional.linear(torch.relu(cat), b64.fc_0.weight, b64.fc_0.bias)
    yr = torch.nn.functional.linear(torch.relu(net), b64.fc_1.weight, b64.fc_1.bias) + torch.nn.functional.linear(cat, b64.shortcut.weight)
    yr.backward(g.double())
    assert _rel(got[0], yr.detach()) <= TOL
    assert _rel(got[1], xr.grad) <= 4 * TOL and _rel(got[2], pr.grad) <= 4 * TOL        # two chained layers; per-pillar sums in another order
    for a, p in zip(got[3], b64.parameters()):
        assert _rel(a, p.grad) <= 4 * TOL


@pytest.mark.parametrize('scale', [1e-7, 1e6])
def test_rows_split_dynamic_range(scale):
    g = torch.Generator(device='cpu').manual_seed(5)
    rows, k, n = 7000, 64, 64
    x = (torch.randn(rows, k, generator=g) * scale).to(DEV)
    dy = (torch.randn(rows, n, generator=g) * scale).to(DEV)
    w = (torch.randn(n, k, generator=g) / 8).to(DEV)
    w[3] *= 1e4                                                                    # one output row far larger than the rest: per-row scales
    y = native.rows_linear_split(x, native.absmax256(x), w)
    y2, y_amax = native.rows_linear_split(x, native.absmax256(x), w, want_amax=True)
    assert torch.equal(y, y2) and float(y_amax.max()) == float(y.abs().max())
    assert _rel(y[:, :3], x.double() @ w[:3].double().t()) <= TOL and _rel(y, x.double() @ w.double().t()) <= TOL
    gw, gb = native.rows_wgrad_split(dy, native.absmax256(dy), x, native.absmax256(x), split=True)
    assert _rel(gw, dy.double().t() @ x.double()) <= TOL and _rel(gb, dy.double().sum(0)) <= TOL
    assert torch.equal(native.rows_wgrad_split(dy[:0], native.absmax256(dy), x[:0], native.absmax256(x)), torch.zeros(n, k + 1, device=DEV))


@pytest.mark.parametrize('two_piece,n', [(False, 4096), (False, 33333), (True, 50_001), (True, 2177)])
def test_pfn_block_split_fused_kernels(two_piece, n):
    """The fused fp32x3 block (csrc/pfn_block_split.hip: one forward kernel, one data-gradient kernel, sign masks instead of x and h)
    against float64 autograd of models/pillar_encoder.py:45-55 -- rows not a multiple of the 128-row tile, both input forms, rows of very
    different magnitude (the intermediates are scaled per wave)."""
    from pcaccumulation_amd.pillar_encoder import ResnetBlockFC
    torch.manual_seed(3 + n)
    m = max(n // 6, 1)
    block = ResnetBlockFC(64, 32).to(DEV)
    torch.nn.init.normal_(block.fc_1.weight, std=0.2)
    row_scale = torch.ones(n, 1, device=DEV)
    row_scale[n // 2:] = 1e-3                                                      # whole waves of small rows next to waves of large ones
    calls = []
    orig_f, orig_d = native.pfn_block_split_forward, native.pfn_block_split_dgrad
    native.pfn_block_split_forward = lambda *a, **k: (calls.append('f'), orig_f(*a, **k))[1]
    native.pfn_block_split_dgrad = lambda *a, **k: (calls.append('d'), orig_d(*a, **k))[1]
    try:
        if two_piece:
            p2v = torch.randint(0, m, (n,), device=DEV, dtype=torch.int32)
            p2v[:m] = torch.arange(m, device=DEV, dtype=torch.int32)
            pidx = PillarIndex.from_point_map(p2v, m)
            x = (torch.randn(n, 32, device=DEV) * row_scale).requires_grad_(True)
            pooled = (torch.randn(m, 32, device=DEV) * 7).requires_grad_(True)
            y = block.forward_pooled(x, pooled, pidx)
        else:
            x = (torch.randn(n, 64, device=DEV) * row_scale).requires_grad_(True)
            pooled = None
            y = block(x)
        g = torch.randn(n, 32, device=DEV) * row_scale
        y.backward(g)
    finally:
        native.pfn_block_split_forward, native.pfn_block_split_dgrad = orig_f, orig_d
    assert calls == ['f', 'd'], 'fp32x3 mode: the block runs on its fused kernels'
    assert float(ops.amax_of(y.detach()).max()) == float(y.detach().abs().max()), 'the store phase reports the output maximum'
    b64 = ResnetBlockFC(64, 32).to(DEV).double()
    b64.load_state_dict({k: v.double() for k, v in block.state_dict().items()})
    xr = x.detach().double().requires_grad_(True)
    pr = pooled.detach().double().requires_grad_(True) if two_piece else None
    cat = torch.cat([xr, pr[p2v.long()]], dim=1) if two_piece else xr
    net = torch.nn.functional.linear(torch.relu(cat), b64.fc_0.weight, b64.fc_0.bias)
    yr = torch.nn.functional.linear(torch.relu(net), b64.fc_1.weight, b64.fc_1.bias) + torch.nn.functional.linear(cat, b64.shortcut.weight)
    yr.backward(g.double())
    assert _rel(y.detach(), yr.detach()) <= TOL
    half = n // 2                                                                  # the small rows are held to their own magnitude
    assert _rel(y.detach()[half:] - block.fc_1.bias, yr.detach()[half:] - b64.fc_1.bias) <= 2e-4   # bias 0.1 next to 1e-3 rows: fp32 rounding of the sum
    assert _rel(x.grad, xr.grad) <= 4 * TOL and _rel(x.grad[half:], xr.grad[half:]) <= 4 * TOL
    if two_piece:
        assert _rel(pooled.grad, pr.grad) <= 4 * TOL
    for a, p in zip(block.parameters(), b64.parameters()):
        assert _rel(a.grad, p.grad) <= 4 * TOL


def test_pfn_block_split_matches_unfused_masks():
    """xmask / hmask are exactly the signs the unfused layers would use (x > 0, h > 0), nonfinite input propagates."""
    torch.manual_seed(9)
    n = 1000
    x = torch.randn(n, 64, device=DEV)
    x[5, 7] = 0.0
    w0, ws, w1 = torch.randn(32, 64, device=DEV) / 8, torch.randn(32, 64, device=DEV) / 8, torch.randn(32, 32, device=DEV) / 6
    b0, b1 = torch.randn(32, device=DEV), torch.randn(32, device=DEV)
    out, hr, xmask, hmask, out_amax, hr_amax = native.pfn_block_split_forward(x, native.absmax256(x), None, None, None, w0, b0, ws, w1, b1)
    bits = torch.arange(64, device=DEV)
    assert torch.equal(((xmask[:, None] >> bits) & 1).bool(), x > 0)
    assert torch.equal(((hmask[:, None].long() >> bits[:32]) & 1).bool(), hr > 0)
    h64 = torch.relu(x.double()) @ w0.double().t() + b0.double()
    assert _rel(hr, torch.relu(h64)) <= TOL and float(hr_amax.max()) == float(hr.max())
    x[17, 3] = float('nan')
    out2 = native.pfn_block_split_forward(x, native.absmax256(x), None, None, None, w0, b0, ws, w1, b1)[0]
    assert torch.isnan(out2[17]).all()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pfn_pool_block_is_pool_then_block(dtype):
    """models/pillar_encoder.py:116-118 as one autograd node (max-pool + broadcast + concatenation + block; the max-pool's gradient added
    into the rows' direct gradient by pcacc_segment_max_backward_acc) == the same three steps as separate nodes, bit for bit
    in the output and the data gradient."""
    from pcaccumulation_amd.pillar_encoder import ResnetBlockFC
    torch.manual_seed(11)
    n, m = 60_001, 9_000
    p2v = torch.randint(0, m, (n,), device=DEV, dtype=torch.int32)
    p2v[:m] = torch.arange(m, device=DEV, dtype=torch.int32)
    pidx = PillarIndex.from_point_map(p2v, m)
    block = ResnetBlockFC(64, 32).to(DEV)
    torch.nn.init.normal_(block.fc_1.weight, std=0.2)
    ops.set_split(dtype == torch.float32)
    x0 = torch.randn(n, 32, device=DEV).to(dtype)
    g = torch.randn(n, 32, device=DEV).to(dtype)
    res = []
    for fused in (True, False):
        x = x0.clone().requires_grad_(True)
        block.zero_grad()
        assert ops.pfn_pool_block_available(block, x, pidx)
        if fused:
            y = ops.pfn_block(block, x, None, pidx, pool=True)
        else:
            y = block.forward_pooled(x, ops.carry_amax(x, ops.segment_max(x, pidx)), pidx)
        y.backward(g)
        res.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in block.parameters()]))
        if dtype == torch.float32:
            assert float(ops.amax_of(x.grad).max()) >= float(x.grad.abs().max()) if not fused else float(ops.amax_of(x.grad).max()) == float(x.grad.abs().max())
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):                                         # weight gradients: slices summed by atomics
        assert _rel(a, b) <= 1e-5
