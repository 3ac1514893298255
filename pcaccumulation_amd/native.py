"""ctypes binding of libpcacc_hip.so (include/pcacc.h).

PyTorch is plumbing here: it owns device memory and the stream; the entry points get raw device
pointers, sizes and the current HIP stream.  There is NO fallback: if the library is missing or a
tensor lives on the CPU the call raises -- the oracle under oracle/ is test infrastructure and is
never reached from this module.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PCACC_LIB') or os.path.join(_HERE, 'csrc', 'libpcacc_hip.so')   # PCACC_LIB: experiment builds (tools/)

F32, BF16 = 0, 1
_ERR = {-1: 'PCACC_E_ARG', -2: 'PCACC_E_WORKSPACE', -3: 'PCACC_E_LAUNCH'}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load libpcacc_hip.so (after torch, so that it binds to the HIP runtime torch already loaded)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError('libpcacc_hip.so is not built (%s): run `python -c "import __graft_entry__ as g; '
                              'g.build()"` or `make -C pcaccumulation_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        for name in EXPORTS:
            try:
                getattr(_lib, name).restype = ctypes.c_int
            except AttributeError:
                if not os.environ.get('PCACC_LIB'):               # an experiment build of an earlier round (A/B runs) may lack entry points added since
                    raise
        _lib.pcacc_target.restype = ctypes.c_char_p
    return _lib


# every symbol include/pcacc.h declares (tests check the .so exports exactly these)
EXPORTS = [
    'pcacc_reload_switches', 'pcacc_cat2_rows', 'pcacc_collate_voxelize_workspace_bytes', 'pcacc_collate_voxelize', 'pcacc_rows_linear_split_dual', 'pcacc_rows_linear_few_dual', 'pcacc_pfn_block_split_forward_dual', 'pcacc_pool_skip_relu_backward_strided_y32', 'pcacc_conv3x3_split_dual', 'pcacc_conv3x3_split_cat', 'pcacc_upconv2x2_split_dual', 'pcacc_voxelize_workspace_bytes', 'pcacc_voxelize', 'pcacc_cell_index',
    'pcacc_frame_pillars_workspace_bytes', 'pcacc_frame_pillars', 'pcacc_compact_mask_workspace_bytes', 'pcacc_compact_mask',
    'pcacc_csr_workspace_bytes', 'pcacc_csr_build', 'pcacc_segment_mean3_maxlabel',
    'pcacc_segment_workspace_bytes', 'pcacc_segment_max', 'pcacc_segment_max_backward', 'pcacc_segment_sum', 'pcacc_scatter_sum_small_workspace_bytes', 'pcacc_scatter_sum_small',
    'pcacc_pfn_features', 'pcacc_pfn_features_ordered', 'pcacc_rows_linear', 'pcacc_rows_wgrad', 'pcacc_pillar_scatter', 'pcacc_gather_rows',
    'pcacc_bilinear_gather', 'pcacc_bilinear_gather_backward', 'pcacc_bev_warp', 'pcacc_bev_warp_dual', 'pcacc_rigid_transform',
    'pcacc_sinkhorn_kabsch_workspace_bytes', 'pcacc_sinkhorn_kabsch', 'pcacc_chamfer_workspace_bytes', 'pcacc_chamfer_forward', 'pcacc_chamfer_backward',
    'pcacc_cluster_workspace_bytes', 'pcacc_cluster', 'pcacc_conv3x3_prepare_weights', 'pcacc_conv3x3_bf16',
    'pcacc_rows_linear_bf16', 'pcacc_rows_linear_mixed', 'pcacc_rows_wgrad_mixed',
    'pcacc_segment_max_t', 'pcacc_segment_max_dual', 'pcacc_segment_max_canvas', 'pcacc_segment_max_canvas_backward', 'pcacc_segment_max_backward_t', 'pcacc_segment_max_backward_acc', 'pcacc_segment_sum_t', 'pcacc_rows_wgrad_bf16_workspace_bytes', 'pcacc_rows_wgrad_bf16', 'pcacc_sample_subsets', 'pcacc_conv3x3_wgrad_workspace_bytes', 'pcacc_conv3x3_wgrad_bf16', 'pcacc_upload_words', 'pcacc_bilinear_base_cells', 'pcacc_bilinear_sorted_workspace_bytes', 'pcacc_bilinear_gather_backward_sorted', 'pcacc_prep_points',
    'pcacc_kabsch_cov_forward', 'pcacc_kabsch_cov_backward', 'pcacc_kabsch_rt_forward', 'pcacc_kabsch_rt_backward', 'pcacc_ego_affinity_forward', 'pcacc_ego_affinity_backward_workspace_bytes', 'pcacc_ego_affinity_backward', 'pcacc_ego_perm_forward', 'pcacc_ego_perm_backward',
    'pcacc_sinkhorn_train_workspace_bytes', 'pcacc_sinkhorn_forward', 'pcacc_sinkhorn_backward',
    'pcacc_seg_loss_workspace_bytes', 'pcacc_seg_loss_forward', 'pcacc_seg_loss_backward',
    'pcacc_offset_loss_workspace_bytes', 'pcacc_offset_loss_forward', 'pcacc_offset_loss_backward',
    'pcacc_frames_max', 'pcacc_frames_max_backward', 'pcacc_rows_linear_cat_bf16', 'pcacc_rows_wgrad_cat_bf16',
    'pcacc_pillar_scatter_timed', 'pcacc_pillar_scatter_t', 'pcacc_timer_create', 'pcacc_timer_elapsed_us', 'pcacc_timer_destroy',
    'pcacc_svd3', 'pcacc_svd3_backward', 'pcacc_conv3x3_deep_supported', 'pcacc_conv3x3_deep_bf16', 'pcacc_conv3x3_prepare_weights_pair',
    'pcacc_conv3x3_masked_bf16',
    'pcacc_conv3x3_wgrad_deep_supported', 'pcacc_conv3x3_wgrad_deep_workspace_bytes', 'pcacc_conv3x3_wgrad_deep_bf16', 'pcacc_bn_rows_workspace_bytes', 'pcacc_bn_rows_forward', 'pcacc_bn_rows_backward',
    'pcacc_tube_rows', 'pcacc_tube_code', 'pcacc_tube_code_backward', 'pcacc_tube_pose_forward', 'pcacc_tube_gap_forward', 'pcacc_tube_finish',
    'pcacc_tube_gap_backward', 'pcacc_tube_pose_backward', 'pcacc_rows_wgrad_few_supported', 'pcacc_rows_wgrad_few_workspace_bytes', 'pcacc_rows_wgrad_few',
    'pcacc_maxpool2x2_bf16', 'pcacc_pool_skip_relu_backward_bf16',
    'pcacc_pfn_block_forward', 'pcacc_pfn_block_backward_workspace_bytes', 'pcacc_pfn_block_backward', 'pcacc_conv3x3_split_outmask', 'pcacc_conv3x3_outmask_supported', 'pcacc_conv3x3_outmask_bf16', 'pcacc_maxpool2x2_f32', 'pcacc_pool_skip_relu_backward_f32', 'pcacc_pool_skip_relu_backward_strided_bf16', 'pcacc_pool_skip_relu_backward_strided_f32', 'pcacc_bn_relu_rows_forward', 'pcacc_bn_relu_rows_backward', 'pcacc_bn_rows_backward_m', 'pcacc_bn_rows_forward_dual', 'pcacc_pfn_block_split_forward', 'pcacc_pfn_block_split_dgrad', 'pcacc_inv4x4',
    'pcacc_conv3x3_split_prepare_weights', 'pcacc_conv3x3_split_supported', 'pcacc_conv3x3_split', 'pcacc_conv3x3_wgrad_split_workspace_bytes',
    'pcacc_conv3x3_wgrad_split', 'pcacc_absmax256',
    'pcacc_rows_linear_split', 'pcacc_rows_linear_cat_split', 'pcacc_rows_wgrad_split_workspace_bytes', 'pcacc_rows_wgrad_split',
    'pcacc_rows_wgrad_cat_split', 'pcacc_upconv2x2_split_prepare_weights', 'pcacc_prepare_weights_batch', 'pcacc_upconv2x2_bf16_supported', 'pcacc_upconv2x2_bf16_prepare_weights', 'pcacc_upconv2x2_bf16',
    'pcacc_upconv2x2_bf16_wgrad_workspace_bytes', 'pcacc_upconv2x2_bf16_wgrad', 'pcacc_upconv2x2_split_supported', 'pcacc_upconv2x2_split',
    'pcacc_upconv2x2_wgrad_split_workspace_bytes', 'pcacc_upconv2x2_wgrad_split',
    'pcacc_head_conv3x3_supported', 'pcacc_head_conv3x3_forward', 'pcacc_head_conv3x3_dgrad', 'pcacc_head_conv3x3_wgrad',
    'pcacc_head_conv3x3_wgrad_workspace_bytes',
]


def x3_experiment(word):
    """Precision-map experiment (csrc/common.h, -DPCACC_X3_EXPERIMENT build only): the fp32x3 kernels launched on the current stream after this call drop /
    round the operand halves `word` names.  Raises on the shipped library, which has no such entry points."""
    l = lib()
    if not hasattr(l, 'pcacc_x3_experiment_conv'):
        raise NativeError('x3_experiment: this is not the experiment build of libpcacc_hip.so (tools/r06_precision_map.py builds and loads it)')
    for name in ('pcacc_x3_experiment_conv', 'pcacc_x3_experiment_rows', 'pcacc_x3_experiment_pfn'):
        fn = getattr(l, name)
        fn.restype = ctypes.c_int
        _check(fn(int(word), _stream()), name)


def reload_switches():
    """The launchers read their A/B environment switches once per process (include/pcacc.h: pcacc_reload_switches); call this after changing one."""
    lib().pcacc_reload_switches()


def _check(rc, what):
    if rc != 0:
        raise NativeError('%s failed: %s' % (what, _ERR.get(rc, rc)))


def _dev(t, dtype=None, what='tensor'):
    if not t.is_cuda:
        raise NativeError('%s must live on the GPU (got %s); the HIP path has no CPU fallback' % (what, t.device))
    if dtype is not None and t.dtype != dtype:
        raise NativeError('%s must be %s, got %s' % (what, dtype, t.dtype))
    if not t.is_contiguous():
        raise NativeError('%s must be contiguous' % what)
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """The current HIP stream of the current device as a void*.  torch.cuda.current_stream() builds a Stream object through three Python layers
    (9 us per call, ~750 calls per step: 3 ms of host time at a step that is host-bound below four sequences); the raw accessors are plain C calls."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _i64(v):
    return ctypes.c_int64(int(v))


def _dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise NativeError('canvas / feature map must be float32 or bfloat16, got %s' % t.dtype)


# ---------------------------------------------------------------------------------------------------
def voxelize(points, voxel_size, pc_range, grid, nt, max_voxels):
    """points [n,4] f32 cuda -> (coords [max_voxels,4] i32, p2v [n] i32, num_voxels [1] i32), all on device."""
    n = points.shape[0]
    nx, ny, nz = (int(g) for g in grid)
    dev = points.device
    coords = torch.empty((max_voxels, 4), dtype=torch.int32, device=dev)
    p2v = torch.empty((n,), dtype=torch.int32, device=dev)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_voxelize_workspace_bytes(_i64(n), nx, ny, nz, int(nt), ctypes.byref(need)), 'voxelize_workspace')
    ws = _ws(need.value, dev)
    vs = (ctypes.c_float * 3)(*[float(v) for v in voxel_size])
    rg = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    _check(lib().pcacc_voxelize(_dev(points, torch.float32, 'points'), _i64(n), vs, rg, nx, ny, nz, int(nt),
                                int(max_voxels), _dev(coords), _dev(p2v), _dev(num), _dev(ws),
                                ctypes.c_size_t(ws.numel()), _stream()), 'voxelize')
    return coords, p2v, num


def collate_voxelize(samples, voxel_size, pc_range, grid, nt):
    """Voxelise and collate a list of device samples (dicts with input_points [n,3] f64, time_indice [n,1] i64 and optionally sd_labels /
    inst_labels / fb_labels [n,1] i64) in one set of launches (include/pcacc.h: pcacc_collate_voxelize).  -> dict with the collated
    input_points [N,3] f64, time_indice [N,2] f64, the label tensors [N,1] i64, coords [N,5] f64 (rows beyond the pillar count unused),
    point_to_voxel_map [N,1] i32, num_voxels [B] i32 (device)."""
    B = len(samples)
    dev = samples[0]['input_points'].device
    nx, ny, nz = (int(g) for g in grid)
    keep = []                                                                  # contiguous views must outlive the launch

    def ptrs(key, dtype):
        arr = (ctypes.c_void_p * B)()
        for i, s in enumerate(samples):
            t = s[key]
            t = t.reshape(t.shape[0], -1) if t.dim() > 1 else t
            if t.dtype != dtype or not t.is_contiguous():
                t = t.to(dtype).contiguous()
            if not t.is_cuda:
                raise NativeError('collate_voxelize: %s must be a GPU tensor' % key)
            keep.append(t)
            arr[i] = t.data_ptr()
        return arr
    counts = (ctypes.c_int64 * B)(*[int(s['input_points'].shape[0]) for s in samples])
    n = sum(counts)
    pts = ptrs('input_points', torch.float64)
    tim = ptrs('time_indice', torch.int64)
    labels = {k: (ptrs(k, torch.int64) if all(k in s for s in samples) else None) for k in ('sd_labels', 'inst_labels', 'fb_labels')}
    out = {'input_points': torch.empty((n, 3), dtype=torch.float64, device=dev), 'time_indice': torch.empty((n, 2), dtype=torch.float64, device=dev)}
    for k, v in labels.items():
        if v is not None:
            out[k] = torch.empty((n, 1), dtype=torch.int64, device=dev)
    coords = torch.empty((n, 5), dtype=torch.float64, device=dev)
    p2v = torch.empty((n, 1), dtype=torch.int32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_collate_voxelize_workspace_bytes(_i64(n), B, nx, ny, nz, int(nt), ctypes.byref(need)), 'collate_voxelize_workspace')
    ws = _ws(need.value, dev)
    vs = (ctypes.c_float * 3)(*[float(v) for v in voxel_size])
    rg = (ctypes.c_float * 6)(*[float(v) for v in pc_range])
    opt = lambda k: _dev(out[k]) if labels[k] is not None else None
    _check(lib().pcacc_collate_voxelize(pts, tim, labels['sd_labels'], labels['inst_labels'], labels['fb_labels'], counts, B, vs, rg, nx, ny, nz, int(nt),
                                        _dev(out['input_points']), _dev(out['time_indice']), opt('sd_labels'), opt('inst_labels'), opt('fb_labels'),
                                        _dev(coords), _dev(p2v), _dev(num), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'collate_voxelize')
    out.update(coords=coords, point_to_voxel_map=p2v, num_voxels=num, _keep=keep)
    return out


def cell_index(coords, nx, ny, nt, n_batch):
    """coords [m,5] (b,z,y,x,t) f64 or i32 -> (cell [m] i32, cell2pillar [n_batch*nt*ny*nx] i32)."""
    m = coords.shape[0]
    dev = coords.device
    if coords.dtype == torch.float64:
        is_f64 = 1
    elif coords.dtype == torch.int32:
        is_f64 = 0
    else:
        raise NativeError('coordinates must be float64 (collate layout) or int32, got %s' % coords.dtype)
    cell = torch.empty((m,), dtype=torch.int32, device=dev)
    c2p = torch.empty((n_batch * nt * ny * nx,), dtype=torch.int32, device=dev)
    _check(lib().pcacc_cell_index(_dev(coords, None, 'coordinates'), is_f64, _i64(m), int(nx), int(ny), int(nt),
                                  int(n_batch), _dev(cell), _dev(c2p), _stream()), 'cell_index')
    return cell, c2p


def frame_pillars(cell2pillar, cells_per_frame, m):
    dev = cell2pillar.device
    n_cells = cell2pillar.numel()
    sorted_p = torch.zeros((m,), dtype=torch.int32, device=dev)      # entries beyond the occupied-cell count (duplicate cells) stay valid ids
    offs = torch.empty((n_cells // cells_per_frame + 1,), dtype=torch.int32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_frame_pillars_workspace_bytes(_i64(n_cells), ctypes.byref(need)), 'frame_pillars_workspace')
    ws = _ws(need.value, dev)
    _check(lib().pcacc_frame_pillars(_dev(cell2pillar, torch.int32), _i64(n_cells), _i64(cells_per_frame),
                                     _dev(sorted_p), _dev(offs), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()),
           'frame_pillars')
    return sorted_p, offs


def compact_mask(mask, size):
    """mask [n] bool / uint8 (contiguous) -> [size] int64: the indices of its non-zero entries, ascending (torch.nonzero_static(mask, size=size)[:, 0] for a
    `size` that equals the number of non-zeros -- the forward knows it from its host sync; entries beyond the count hold -1, as nonzero_static's do).
    A mask view that does not start on a 16-byte boundary is copied first (the kernel reads 16 mask bytes per lane)."""
    if mask.dtype not in (torch.bool, torch.uint8) or mask.dim() != 1:
        raise NativeError('compact_mask: a 1-D bool / uint8 mask expected, got %s %s' % (mask.dtype, tuple(mask.shape)))
    n, dev = mask.shape[0], mask.device
    out = torch.empty((int(size),), dtype=torch.int64, device=dev)
    if size == 0:
        return out
    if n == 0:
        return out.fill_(-1)
    if mask.data_ptr() % 16 or not mask.is_contiguous():
        mask = mask.clone(memory_format=torch.contiguous_format)
        if mask.data_ptr() % 16:
            raise NativeError('compact_mask: the allocator returned an unaligned block')
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_compact_mask_workspace_bytes(_i64(n), ctypes.byref(need)), 'compact_mask_workspace')
    ws = _ws(need.value, dev)
    _check(lib().pcacc_compact_mask(_dev(mask, None, 'mask'), _i64(n), _dev(out), _i64(size), None, _dev(ws), ctypes.c_size_t(ws.numel()), _stream()),
           'compact_mask')
    return out


def csr_build(p2v, m):
    n = p2v.shape[0]
    dev = p2v.device
    offs = torch.empty((m + 1,), dtype=torch.int32, device=dev)
    order = torch.empty((n,), dtype=torch.int32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_csr_workspace_bytes(_i64(n), _i64(m), ctypes.byref(need)), 'csr_workspace')
    ws = _ws(need.value, dev)
    _check(lib().pcacc_csr_build(_dev(p2v, torch.int32, 'p2v'), _i64(n), _i64(m), _dev(offs), _dev(order), _dev(ws),
                                 ctypes.c_size_t(ws.numel()), _stream()), 'csr_build')
    return offs, order


def segment_mean3_maxlabel(points, labels, offs, order, m):
    dev = points.device
    mean = torch.empty((m, 3), dtype=torch.float32, device=dev)
    lab_out = torch.empty((m,), dtype=torch.int64, device=dev) if labels is not None else None
    _check(lib().pcacc_segment_mean3_maxlabel(
        _dev(points, torch.float32, 'points'), _dev(labels, torch.int64, 'labels') if labels is not None else None,
        _dev(offs, torch.int32), _dev(order, torch.int32), _i64(m), _dev(mean),
        _dev(lab_out) if lab_out is not None else None, _stream()), 'segment_mean3_maxlabel')
    return mean, lab_out


def _segment_ws(n, m, c, dev):
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_segment_workspace_bytes(_i64(n), _i64(m), int(c), ctypes.byref(need)), 'segment_workspace')
    return _ws(need.value, dev)


def _seg_two_level(n, m):
    return m > 0 and n // m > 16


def segment_max(src, offs, order, m):
    """Per-segment channel-wise max and arg.  Rows f32 or bf16; [m,c] comes back in the rows' type for short segments and in
    f32 for long ones (two-level reduction), arg [m,c] i32."""
    n, c = src.shape
    if src.dtype not in (torch.float32, torch.bfloat16):
        src = src.float()
    out_dtype = torch.float32 if _seg_two_level(n, m) else src.dtype
    out = torch.empty((m, c), dtype=out_dtype, device=src.device)
    arg = torch.empty((m, c), dtype=torch.int32, device=src.device)
    ws = _segment_ws(n, m, c, src.device)
    _check(lib().pcacc_segment_max_t(_dev(src, None, 'src'), _dtype_code(src), int(c), _dev(offs, torch.int32), _dev(order, torch.int32),
                                     _i64(n), _i64(m), _dev(out), _dev(arg), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()),
           'segment_max')
    return out, arg


def segment_max_dual(src, offs, order, m):
    """segment_max on f32 rows of short segments -> (out [m,c] f32, out as bf16, arg [m,c] i32): the 'mixed' mode's shadow from the same store."""
    n, c = src.shape
    if _seg_two_level(n, m):
        raise NativeError('segment_max_dual: long segments (n / m > 16) have no second output')
    out = torch.empty((m, c), dtype=torch.float32, device=src.device)
    out16 = torch.empty((m, c), dtype=torch.bfloat16, device=src.device)
    arg = torch.empty((m, c), dtype=torch.int32, device=src.device)
    ws = _segment_ws(n, m, c, src.device)
    _check(lib().pcacc_segment_max_dual(_dev(src, torch.float32, 'src'), int(c), _dev(offs, torch.int32), _dev(order, torch.int32), _i64(n), _i64(m),
                                        _dev(out), _dev(out16), _dev(arg), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'segment_max_dual')
    return out, out16, arg


def segment_max_backward(grad_out, arg, p2v, n, out_dtype=None):
    """grad of segment_max w.r.t. its rows: [n,c] in out_dtype (default: grad_out's type); both f32 or bf16."""
    c = grad_out.shape[1]
    if grad_out.dtype not in (torch.float32, torch.bfloat16):
        grad_out = grad_out.float()
    out_dtype = out_dtype or grad_out.dtype
    g = torch.empty((n, c), dtype=out_dtype, device=grad_out.device)
    _check(lib().pcacc_segment_max_backward_t(_dev(grad_out, None, 'grad_out'), _dtype_code(grad_out), _dev(arg, torch.int32),
                                              _dev(p2v, torch.int32), _i64(n), int(c), _dev(g), _dtype_code(g), _stream()),
           'segment_max_backward')
    return g


def segment_max_backward_acc(grad_out, arg, p2v, grad_src, want_amax=False):
    """grad_src [n,c] += grad of segment_max w.r.t. its rows, in place (f32 / bf16 each); want_amax: -> absmax256 array of the sums."""
    n, c = grad_src.shape
    if grad_out.dtype not in (torch.float32, torch.bfloat16):
        grad_out = grad_out.float()
    amax = _zero256(grad_src.device) if want_amax else None
    _check(lib().pcacc_segment_max_backward_acc(_dev(grad_out, None, 'grad_out'), _dtype_code(grad_out), _dev(arg, torch.int32), _dev(p2v, torch.int32),
                                                _i64(n), int(c), _dev(grad_src, None, 'grad_src'), _dtype_code(grad_src),
                                                _dev(amax) if want_amax else None, _stream()), 'segment_max_backward_acc')
    return amax


def segment_max_canvas(src, offs, order, m, cell2pillar):
    """The encoder's last pooling written straight into the BEV canvas ('mixed' mode): src [n,c] f32 -> (canvas32 [n_cells,c] f32, canvas16 [n_cells,c] bf16,
    arg [m,c] i32).  bench.py times this dispatch through `scatter_timer` (its roofline object)."""
    n, c = src.shape
    n_cells = cell2pillar.numel()
    dev = src.device
    canvas32 = torch.empty((n_cells, c), dtype=torch.float32, device=dev)
    canvas16 = torch.empty((n_cells, c), dtype=torch.bfloat16, device=dev)
    arg = torch.empty((m, c), dtype=torch.int32, device=dev)
    t = None
    if scatter_timer is not None and c >= 32:
        t = KernelTimer()
        scatter_timer.append((t, n_cells, c, m, 'fused', n))
    _check(lib().pcacc_segment_max_canvas(_dev(src, torch.float32, 'src'), int(c), _dev(offs, torch.int32), _dev(order, torch.int32), _i64(n), _i64(m),
                                          _dev(cell2pillar, torch.int32), _i64(n_cells), _dev(canvas32), _dev(canvas16), _dev(arg),
                                          t.start if t else None, t.stop if t else None, _stream()), 'segment_max_canvas')
    return canvas32, canvas16, arg


def segment_max_canvas_backward(grad_canvas, arg, p2v, cell, n, out_dtype=None):
    """grad of the point rows from the canvas gradient [n_cells,c] (f32 / bf16), read in place through the pillars' cell numbers."""
    c = grad_canvas.shape[1]
    out_dtype = out_dtype or grad_canvas.dtype
    out = torch.empty((n, c), dtype=out_dtype, device=grad_canvas.device)
    _check(lib().pcacc_segment_max_canvas_backward(_dev(grad_canvas, None, 'grad_canvas'), _dtype_code(grad_canvas), _dev(arg, torch.int32), _dev(p2v, torch.int32),
                                                   _dev(cell, torch.int32), _i64(n), int(c), _dev(out), _dtype_code(out), _stream()), 'segment_max_canvas_backward')
    return out


def segment_sum(src, offs, order, m):
    """Per-segment sums (accumulated in f32): [m,c] in the rows' type for short segments, f32 for long ones."""
    n, c = src.shape
    if src.dtype not in (torch.float32, torch.bfloat16):
        src = src.float()
    out_dtype = torch.float32 if _seg_two_level(n, m) else src.dtype
    out = torch.empty((m, c), dtype=out_dtype, device=src.device)
    ws = _segment_ws(n, m, c, src.device)
    _check(lib().pcacc_segment_sum_t(_dev(src, None, 'src'), _dtype_code(src), int(c), _dev(offs, torch.int32), _dev(order, torch.int32),
                                     _i64(n), _i64(m), _dev(out), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'segment_sum')
    return out


# bench.py sets this to a list to time the dominant kernel live: (KernelTimer, n_cells, c, m, dtype) per launch, only for
# canvases of >= 32 channels.  The two events are attached to the dispatch (pcacc_pillar_scatter_timed).
scatter_timer = None


class KernelTimer(object):
    """A pair of HIP events for one dispatch; elapsed_us() after the stream has been synchronised."""

    def __init__(self):
        self.start, self.stop = ctypes.c_void_p(), ctypes.c_void_p()
        _check(lib().pcacc_timer_create(ctypes.byref(self.start), ctypes.byref(self.stop)), 'timer_create')

    def elapsed_us(self):
        us = ctypes.c_float(0.0)
        _check(lib().pcacc_timer_elapsed_us(self.start, self.stop, ctypes.byref(us)), 'timer_elapsed_us')
        return us.value

    def __del__(self):
        try:
            lib().pcacc_timer_destroy(self.start, self.stop)
        except Exception:
            pass


def pillar_scatter(feats, cell2pillar, out_dtype=torch.float32):
    """feats [m,c] f32 (or bf16 with a bf16 canvas: rows copied as they are) -> canvas [n_cells, c] (channels-last) of out_dtype."""
    c = feats.shape[1]
    n_cells = cell2pillar.numel()
    canvas = torch.empty((n_cells, c), dtype=out_dtype, device=feats.device)
    t = None
    if scatter_timer is not None and c >= 32:
        t = KernelTimer()
        scatter_timer.append((t, n_cells, c, feats.shape[0], out_dtype, feats.dtype))
    _check(lib().pcacc_pillar_scatter_t(_dev(feats, None, 'feats'), _dtype_code(feats), _dev(cell2pillar, torch.int32), _i64(n_cells),
                                        int(c), _dev(canvas), _dtype_code(canvas), t.start if t else None, t.stop if t else None,
                                        _stream()), 'pillar_scatter')
    return canvas


def cat2_rows(a, b):
    """cat((a, b), -1) for two contiguous [..., ca] / [..., cb] tensors of one dtype whose rows are multiples of 16 bytes (channels-last maps)."""
    ca, cb = a.shape[-1], b.shape[-1]
    rows = a.numel() // ca
    if a.dtype != b.dtype or tuple(a.shape[:-1]) != tuple(b.shape[:-1]) or not a.is_contiguous() or not b.is_contiguous():
        raise NativeError('cat2_rows: two contiguous tensors of one dtype and equal leading dimensions expected')
    out = torch.empty(tuple(a.shape[:-1]) + (ca + cb,), dtype=a.dtype, device=a.device)
    _check(lib().pcacc_cat2_rows(_dev(a, None, 'a'), int(ca * a.element_size()), _dev(b, None, 'b'), int(cb * b.element_size()), _i64(rows), _dev(out),
                                 _stream()), 'cat2_rows')
    return out


def gather_rows(src, idx):
    """out[i] = src[idx[i]] for a 2-D src with row size a multiple of 4 bytes; idx i32 (-1 -> zeros)."""
    row_bytes = src.shape[1] * src.element_size()
    out = torch.empty((idx.shape[0], src.shape[1]), dtype=src.dtype, device=src.device)
    _check(lib().pcacc_gather_rows(_dev(src, None, 'src'), int(row_bytes), _dev(idx, torch.int32, 'idx'),
                                   _i64(idx.shape[0]), _dev(out), _stream()), 'gather_rows')
    return out


def bilinear_gather(fmap, points, map_idx, x_scale, y_scale):
    """fmap [n_maps,h,w,c] channels-last f32/bf16; points [k,3] f32; map_idx [k] i32 -> [k,c] f32."""
    n_maps, h, w, c = fmap.shape
    k = points.shape[0]
    out = torch.empty((k, c), dtype=torch.float32, device=fmap.device)
    _check(lib().pcacc_bilinear_gather(_dev(fmap, None, 'fmap'), _dtype_code(fmap), n_maps, h, w, c,
                                       _dev(points, torch.float32, 'points'), _dev(map_idx, torch.int32, 'map_idx'),
                                       _i64(k), ctypes.c_float(x_scale), ctypes.c_float(y_scale), _dev(out), _stream()),
           'bilinear_gather')
    return out


def bilinear_gather_backward(grad_out, shape, points, map_idx, x_scale, y_scale):
    n_maps, h, w, c = shape
    g = torch.zeros(shape, dtype=torch.float32, device=grad_out.device)
    _check(lib().pcacc_bilinear_gather_backward(_dev(grad_out, torch.float32, 'grad_out'), n_maps, h, w, c,
                                                _dev(points, torch.float32), _dev(map_idx, torch.int32),
                                                _i64(points.shape[0]), ctypes.c_float(x_scale), ctypes.c_float(y_scale),
                                                _dev(g), _stream()), 'bilinear_gather_backward')
    return g


def bev_warp(bev, inv_pose, x_reso, y_reso, x_min, y_min):
    """bev [B,T,H,W,C] channels-last; inv_pose [B,T,4,4] f32 -> warped [B,T,H,W,C]."""
    b, t, h, w, c = bev.shape
    out = torch.empty_like(bev)
    _check(lib().pcacc_bev_warp(_dev(bev, None, 'bev'), _dtype_code(bev), b, t, h, w, c,
                                _dev(inv_pose, torch.float32, 'inv_pose'), ctypes.c_float(x_reso), ctypes.c_float(y_reso),
                                ctypes.c_float(x_min), ctypes.c_float(y_min), _dev(out), _stream()), 'bev_warp')
    return out



def bev_warp_dual(bev, inv_pose, x_reso, y_reso, x_min, y_min):
    """bev_warp on an fp32 map -> (warped f32, warped as bf16): the 'mixed' mode's shadow of the warped features from the same store."""
    b, nt, h, w, c = bev.shape
    out = torch.empty_like(bev)
    out16 = torch.empty(bev.shape, dtype=torch.bfloat16, device=bev.device)
    _check(lib().pcacc_bev_warp_dual(_dev(bev, torch.float32, 'bev'), b, nt, h, w, c, _dev(inv_pose, torch.float32, 'inv_pose'), ctypes.c_float(x_reso),
                                     ctypes.c_float(y_reso), ctypes.c_float(x_min), ctypes.c_float(y_min), _dev(out), _dev(out16), _stream()), 'bev_warp_dual')
    return out, out16


def rigid_transform(points, frame_idx, tsfm):
    out = torch.empty_like(points)
    _check(lib().pcacc_rigid_transform(_dev(points, torch.float32, 'points'), _dev(frame_idx, torch.int32, 'frame_idx'),
                                       _dev(tsfm, torch.float32, 'tsfm'), _i64(points.shape[0]), _dev(out), _stream()),
           'rigid_transform')
    return out


def chamfer_forward(xyz1, xyz2):
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dev = xyz1.device
    d1 = torch.empty((b, n), dtype=torch.float32, device=dev)
    d2 = torch.empty((b, m), dtype=torch.float32, device=dev)
    i1 = torch.empty((b, n), dtype=torch.int32, device=dev)
    i2 = torch.empty((b, m), dtype=torch.int32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_chamfer_workspace_bytes(b, n, m, ctypes.byref(need)), 'chamfer_workspace')
    ws = _ws(need.value, dev)
    _check(lib().pcacc_chamfer_forward(_dev(xyz1, torch.float32, 'xyz1'), _dev(xyz2, torch.float32, 'xyz2'), b, n, m,
                                       _dev(d1), _dev(i1), _dev(d2), _dev(i2), _dev(ws), ctypes.c_size_t(ws.numel()),
                                       _stream()), 'chamfer_forward')
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, gd1, gd2, i1, i2):
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    _check(lib().pcacc_chamfer_backward(_dev(xyz1, torch.float32), _dev(xyz2, torch.float32), b, n, m,
                                        _dev(gd1, torch.float32, 'grad_dist1'), _dev(i1, torch.int32),
                                        _dev(gd2, torch.float32, 'grad_dist2'), _dev(i2, torch.int32),
                                        _dev(g1), _dev(g2), _stream()), 'chamfer_backward')
    return g1, g2


ROWS_LINEAR_K = (2, 3, 4, 9, 32, 64, 128)


def rows_linear_supported(k, n):
    return k in ROWS_LINEAR_K and 1 <= n <= 128


def _row_dtype_bit(t, bit, what):
    if t is None or t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return bit
    raise NativeError('%s must be float32 or bfloat16, got %s' % (what, t.dtype))


def rows_linear(x, w, bias=None, residual=None, pre_relu=False, post_relu=False, in_mask=None, out_mask=None, out_dtype=None):
    """y = post(pre(x) @ w^T + bias + residual); x [rows,k], w [n,k] f32 (see include/pcacc.h).  Row tensors may be f32 or
    bf16; all-bf16 calls with k, n in {32,64,128} run on the matrix cores, everything else in fp32 arithmetic."""
    rows, k = x.shape
    n = w.shape[0]
    out_dtype = out_dtype or x.dtype
    y = torch.empty((rows, n), dtype=out_dtype, device=x.device)
    opt = lambda t, what: _dev(t, None, what) if t is not None else None
    flags = (1 if pre_relu else 0) | (2 if post_relu else 0)
    tensors = (x, in_mask, residual, out_mask, y)
    all_bf16 = all(t is None or t.dtype == torch.bfloat16 for t in tensors)
    if all_bf16 and k in (32, 64, 128) and n in (32, 64, 128):
        _check(lib().pcacc_rows_linear_bf16(_dev(x, torch.bfloat16, 'x'), opt(in_mask, 'in_mask'), _dev(w, torch.float32, 'w'),
                                            _dev(bias, torch.float32, 'bias') if bias is not None else None, opt(residual, 'residual'),
                                            opt(out_mask, 'out_mask'), _dev(y), _i64(rows), int(k), int(n), flags, _stream()),
               'rows_linear_bf16')
        return y
    dt = (_row_dtype_bit(x, 1, 'x') | _row_dtype_bit(in_mask, 2, 'in_mask') | _row_dtype_bit(residual, 4, 'residual')
          | _row_dtype_bit(out_mask, 8, 'out_mask') | _row_dtype_bit(y, 16, 'y'))
    _check(lib().pcacc_rows_linear_mixed(_dev(x, None, 'x'), opt(in_mask, 'in_mask'), _dev(w, torch.float32, 'w'),
                                         _dev(bias, torch.float32, 'bias') if bias is not None else None, opt(residual, 'residual'),
                                         opt(out_mask, 'out_mask'), _dev(y), _i64(rows), int(k), int(n), flags, dt, _stream()),
           'rows_linear')
    return y


def _split_aug(out, n, k, split, native_split):
    """split: (dW [n,k], db [n]); contiguous views of `out` when the kernel wrote the split layout, slices of the [n,k+1] rows otherwise."""
    if not split:
        return out
    if native_split:
        flat = out.view(-1)
        return flat[:n * k].view(n, k), flat[n * k:]
    return out[:, :-1], out[:, -1]


def rows_wgrad(dy, x, dy_mask=None, x_relu=False, split=False):
    """[n, k+1] f32 = dYeff^T @ [Xeff | 1]: weight gradient with the bias gradient in the last column (dy, x, dy_mask f32 or bf16).
    split=True: (dW [n,k], db [n]) -- contiguous tensors where the kernel can write them so (no copy when autograd adopts them)."""
    rows, n = dy.shape
    k = x.shape[1]
    out = torch.empty((n, k + 1), dtype=torch.float32, device=dy.device)
    flags = (1 if x_relu else 0) | (2 if split else 0)
    if all(t is None or t.dtype == torch.bfloat16 for t in (dy, dy_mask, x)) and k % 32 == 0 and n % 32 == 0:
        need = ctypes.c_size_t(0)
        _check(lib().pcacc_rows_wgrad_bf16_workspace_bytes(_i64(rows), int(k), int(n), ctypes.byref(need)), 'rows_wgrad_bf16_workspace')
        ws = _ws(need.value, dy.device)
        _check(lib().pcacc_rows_wgrad_bf16(_dev(dy, torch.bfloat16, 'dy'), _dev(dy_mask, torch.bfloat16, 'dy_mask') if dy_mask is not None else None,
                                           _dev(x, torch.bfloat16, 'x'), flags, _i64(rows), int(k), int(n), _dev(out),
                                           _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'rows_wgrad_bf16')
        return _split_aug(out, n, k, split, True)
    dt = _row_dtype_bit(dy, 1, 'dy') | _row_dtype_bit(dy_mask, 2, 'dy_mask') | _row_dtype_bit(x, 4, 'x')
    if lib().pcacc_rows_wgrad_few_supported(int(k), int(n)):
        need = ctypes.c_size_t(0)
        _check(lib().pcacc_rows_wgrad_few_workspace_bytes(_i64(rows), int(k), int(n), ctypes.byref(need)), 'rows_wgrad_few_workspace')
        ws = _ws(need.value, dy.device)
        _check(lib().pcacc_rows_wgrad_few(_dev(dy, None, 'dy'), _dev(dy_mask, None, 'dy_mask') if dy_mask is not None else None, _dev(x, None, 'x'),
                                          flags, _i64(rows), int(k), int(n), _dev(out), dt, _dev(ws), ctypes.c_size_t(ws.numel()),
                                          _stream()), 'rows_wgrad_few')
        return _split_aug(out, n, k, split, rows > 0)
    _check(lib().pcacc_rows_wgrad_mixed(_dev(dy, None, 'dy'), _dev(dy_mask, None, 'dy_mask') if dy_mask is not None else None,
                                        _dev(x, None, 'x'), 1 if x_relu else 0, _i64(rows), int(k), int(n), _dev(out), dt, _stream()),
           'rows_wgrad')
    return _split_aug(out, n, k, split, False)


def pfn_features(points, p2v, pillar_mean, coords, time_indice, vx, vy, x_offset, y_offset, scale, n_frames, order=None):
    """[n,9] f32 pillar-encoder inputs (models/pillar_encoder.py:98-110); time_indice [n,2] f64 (b,t).  order [n] i32: row r = point order[r]."""
    n = points.shape[0]
    out = torch.empty((n, 9), dtype=torch.float32, device=points.device)
    if coords.dtype == torch.float64:
        is_f64 = 1
    elif coords.dtype == torch.int32:
        is_f64 = 0
    else:
        raise NativeError('coordinates must be float64 or int32')
    if time_indice.dtype != torch.float64 or not time_indice.is_contiguous() or time_indice.shape[1] != 2:
        raise NativeError('time_indice must be a contiguous float64 [n,2] tensor')
    tcol = ctypes.c_void_p(time_indice.data_ptr() + 8)                   # column 1 of row 0
    if order is not None and order.shape[0] != n:
        raise NativeError('pfn_features: order has %d entries for %d points' % (order.shape[0], n))
    if order is None:
        _check(lib().pcacc_pfn_features(_dev(points, torch.float32, 'points'), _dev(p2v, torch.int32, 'p2v'),
                                        _dev(pillar_mean, torch.float32, 'pillar_mean'), _dev(coords, None, 'coordinates'), is_f64,
                                        tcol if n else None, _i64(2), _i64(n), ctypes.c_double(vx), ctypes.c_double(vy),
                                        ctypes.c_double(x_offset), ctypes.c_double(y_offset), ctypes.c_float(scale),
                                        ctypes.c_float(n_frames), _dev(out), _stream()), 'pfn_features')
        return out
    _check(lib().pcacc_pfn_features_ordered(_dev(points, torch.float32, 'points'), _dev(p2v, torch.int32, 'p2v'),
                                            _dev(pillar_mean, torch.float32, 'pillar_mean'), _dev(coords, None, 'coordinates'), is_f64,
                                            tcol if n else None, _i64(2), _i64(n), ctypes.c_double(vx), ctypes.c_double(vy),
                                            ctypes.c_double(x_offset), ctypes.c_double(y_offset), ctypes.c_float(scale),
                                            ctypes.c_float(n_frames), _dev(order, torch.int32, 'order') if order is not None else None, _dev(out), _stream()),
           'pfn_features')
    return out


def scatter_sum_small(src, idx, m):
    """out[m,c] = sum of src rows per idx, for m*c <= 8192 (LDS-privatised, no CSR)."""
    n, c = src.shape
    out = torch.empty((m, c), dtype=torch.float32, device=src.device)
    if os.environ.get('PCACC_R05_ABI'):                           # A/B against a round-5 library (PCACC_LIB): its entry point took no workspace
        _check(lib().pcacc_scatter_sum_small(_dev(src, torch.float32, 'src'), _dev(idx, torch.int32, 'idx'), _i64(n), int(c), int(m), _dev(out), _stream()),
               'scatter_sum_small')
        return out
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_scatter_sum_small_workspace_bytes(_i64(n), int(c), int(m), ctypes.byref(need)), 'scatter_sum_small_workspace')
    ws = _ws(need.value, src.device)
    _check(lib().pcacc_scatter_sum_small(_dev(src, torch.float32, 'src'), _dev(idx, torch.int32, 'idx'), _i64(n), int(c), int(m),
                                         _dev(out), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'scatter_sum_small')
    return out


def sinkhorn_kabsch(feats_s, feats_t, coor_s, coor_t, thr2, params, n_iters):
    """Forward-only ego-motion matching for P pairs: returns (perm [P,k,k], pose [P,4,4]); see include/pcacc.h (A8)."""
    P, k, c = feats_s.shape
    dev = feats_s.device
    perm = torch.empty((P, k, k), dtype=torch.float32, device=dev)
    pose = torch.empty((P, 4, 4), dtype=torch.float32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_sinkhorn_kabsch_workspace_bytes(int(P), int(k), ctypes.byref(need)), 'sinkhorn_kabsch_workspace')
    ws = _ws(need.value, dev)
    _check(lib().pcacc_sinkhorn_kabsch(_dev(feats_s, torch.float32, 'feats_s'), _dev(feats_t, torch.float32, 'feats_t'),
                                       _dev(coor_s, torch.float32, 'coor_s'), _dev(coor_t, torch.float32, 'coor_t'),
                                       _dev(thr2, torch.float32, 'thr2'), _dev(params, torch.float32, 'params'), int(P), int(k), int(c),
                                       int(n_iters), _dev(perm), _dev(pose), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()),
           'sinkhorn_kabsch')
    return perm, pose


def cluster(points, offset, sel, batch, n_batches, voxel_size, eps, min_samples, min_p_cluster):
    """Instance labels [N] i64 of the selected points, clustered per sample; see include/pcacc.h (C1)."""
    n = points.shape[0]
    labels = torch.empty((n,), dtype=torch.int64, device=points.device)
    if n == 0:
        return labels
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_cluster_workspace_bytes(_i64(n), ctypes.byref(need)), 'cluster_workspace')
    ws = _ws(need.value, points.device)
    _check(lib().pcacc_cluster(_dev(points, torch.float32, 'points'),
                               _dev(offset, torch.float32, 'offset') if offset is not None else None,
                               _dev(sel, torch.uint8, 'sel'), _dev(batch, torch.int32, 'batch'), _i64(n), int(n_batches),
                               ctypes.c_float(voxel_size), ctypes.c_double(eps), int(min_samples), int(min_p_cluster),
                               _dev(labels), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'cluster')
    return labels


def conv3x3_supported(c_in, c_out):
    return c_in >= 32 and c_out >= 32 and c_in % 32 == 0 and c_out % 32 == 0


def conv3x3_deep_supported(h, w, c_in, c_out):
    """True for the layer shapes the strip kernel of csrc/conv_deep.hip takes (c_in >= 128, small images)."""
    return bool(lib().pcacc_conv3x3_deep_supported(int(h), int(w), int(c_in), int(c_out)))


def conv3x3_prepare_weights(weight, transpose=False):
    """weight f32 [O,I,3,3] or [O,I,3,3,3] (contiguous) -> bf16 [kt*9, O', I'] for pcacc_conv3x3_bf16 (A6/A9 in pcacc.h)."""
    o, i = weight.shape[0], weight.shape[1]
    kt = 3 if weight.dim() == 5 else 1
    shape = (kt * 9, i, o) if transpose else (kt * 9, o, i)
    out = torch.empty(shape, dtype=torch.bfloat16, device=weight.device)
    _check(lib().pcacc_conv3x3_prepare_weights(_dev(weight, torch.float32, 'weight'), int(o), int(i), kt, 1 if transpose else 0,
                                               _dev(out), _stream()), 'conv3x3_prepare_weights')
    return out


def conv3x3_prepare_weights_pair(weight):
    """weight f32 [O,I,3,3] / [O,I,3,3,3] in ANY dense storage order -> (forward form bf16 [kt*9,O,I], data-gradient form bf16 [kt*9,I,O]),
    one launch (pcacc_conv3x3_prepare_weights_pair)."""
    o, i = weight.shape[0], weight.shape[1]
    kt = 3 if weight.dim() == 5 else 1
    fwd = torch.empty((kt * 9, o, i), dtype=torch.bfloat16, device=weight.device)
    bwd = torch.empty((kt * 9, i, o), dtype=torch.bfloat16, device=weight.device)
    strides = (ctypes.c_int64 * weight.dim())(*weight.stride())
    if not weight.is_cuda or weight.dtype != torch.float32:
        raise NativeError('conv3x3_prepare_weights_pair: weight must be a float32 GPU tensor')
    _check(lib().pcacc_conv3x3_prepare_weights_pair(ctypes.c_void_p(weight.data_ptr()), int(o), int(i), kt, strides, _dev(fwd), _dev(bwd),
                                                    _stream()), 'conv3x3_prepare_weights_pair')
    return fwd, bwd


def conv3x3_outmask_supported(h, w, c_in, c_out, kt=1):
    return bool(lib().pcacc_conv3x3_outmask_supported(int(h), int(w), int(c_in), int(c_out), int(kt)))


def conv3x3(x_rows, wp, bias, frames, relu, mask=None, out_mask=None):
    """x_rows bf16 [n_img,h,w,c_in] contiguous, wp from conv3x3_prepare_weights -> bf16 [n_img,h,w,c_out].
    mask (same shape as x_rows): x_rows is the gradient of a ReLU layer whose forward output is `mask`; elements where mask <= 0 are
    read as zero (ReLU backward fused into the staging).  out_mask (shape of the result; no bias / relu / mask with it): the result is
    stored as zero where out_mask <= 0 (conv3x3_outmask_supported)."""
    n_img, h, w, c_in = x_rows.shape
    taps, c_out, wc_in = wp.shape
    if wc_in != c_in:
        raise NativeError('conv3x3: weights prepared for %d input channels, input has %d' % (wc_in, c_in))
    out = torch.empty((n_img, h, w, c_out), dtype=torch.bfloat16, device=x_rows.device)
    if out_mask is not None:
        if bias is not None or relu or mask is not None or tuple(out_mask.shape) != tuple(out.shape):
            raise NativeError('conv3x3: out_mask goes with no bias / relu / input mask and has the shape of the result')
        _check(lib().pcacc_conv3x3_outmask_bf16(_dev(x_rows, torch.bfloat16, 'x'), _dev(wp, torch.bfloat16, 'wp'),
                                                _dev(out_mask, torch.bfloat16, 'out_mask'), _dev(out), int(n_img), int(frames), int(h), int(w),
                                                int(c_in), int(c_out), taps // 9, _stream()), 'conv3x3_outmask')
        return out
    if mask is not None and mask.shape != x_rows.shape:
        raise NativeError('conv3x3: mask shape %s != input shape %s' % (tuple(mask.shape), tuple(x_rows.shape)))
    _check(lib().pcacc_conv3x3_masked_bf16(_dev(x_rows, torch.bfloat16, 'x'), _dev(mask, torch.bfloat16, 'mask') if mask is not None else None,
                                           _dev(wp, torch.bfloat16, 'wp'), _dev(bias, torch.float32, 'bias') if bias is not None else None,
                                           _dev(out), int(n_img), int(frames), int(h), int(w), int(c_in), int(c_out), taps // 9,
                                           1 if relu else 0, _stream()), 'conv3x3')
    return out


def sample_subsets(counts, k, seed):
    """counts [D] i32 on the device -> [D, k] i64: k distinct indices below counts[d] per row (see include/pcacc.h)."""
    d = counts.shape[0]
    out = torch.empty((d, k), dtype=torch.int64, device=counts.device)
    _check(lib().pcacc_sample_subsets(_dev(counts, torch.int32, 'counts'), int(d), int(k), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)),
                                      _dev(out), _stream()), 'sample_subsets')
    return out


def conv3x3_wgrad_supported(c_in, c_out):
    return c_in in (32, 64) and c_out in (32, 64)


def conv3x3_wgrad(dy_rows, x_rows, frames=1, dt=0):
    """dy_rows [n_img,h,w,c_out], x_rows [n_img,h,w,c_in] bf16 -> (dw [c_out, 9, c_in] f32, db [c_out] f32) for frame tap dt of a
    kt=3 layer (db is the full bias gradient for dt = 0)."""
    n_img, h, w, c_out = dy_rows.shape
    c_in = x_rows.shape[3]
    dw = torch.empty((c_out * 9 * c_in + c_out,), dtype=torch.float32, device=dy_rows.device)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_conv3x3_wgrad_workspace_bytes(int(n_img), int(h), int(w), int(c_in), int(c_out), ctypes.byref(need)),
           'conv3x3_wgrad_workspace')
    ws = _ws(need.value, dy_rows.device)
    _check(lib().pcacc_conv3x3_wgrad_bf16(_dev(dy_rows, torch.bfloat16, 'dy'), _dev(x_rows, torch.bfloat16, 'x'), _dev(dw), int(n_img),
                                          int(frames), int(dt), int(h), int(w), int(c_in), int(c_out), _dev(ws),
                                          ctypes.c_size_t(ws.numel()), _stream()), 'conv3x3_wgrad')
    return dw[:c_out * 9 * c_in].view(c_out, 9, c_in), dw[c_out * 9 * c_in:]


def conv3x3_wgrad_deep_supported(h, w, c_in, c_out):
    return bool(lib().pcacc_conv3x3_wgrad_deep_supported(int(h), int(w), int(c_in), int(c_out)))


def conv3x3_wgrad_deep(dy_rows, x_rows, mask=None):
    """dy_rows [n_img,h,w,c_out], x_rows [n_img,h,w,c_in] bf16 -> (dw [c_out, 9, c_in] f32, db [c_out] f32); csrc/conv_deep.hip."""
    n_img, h, w, c_out = dy_rows.shape
    c_in = x_rows.shape[3]
    dw = torch.empty((c_out, 9, c_in), dtype=torch.float32, device=dy_rows.device)
    db = torch.empty((c_out,), dtype=torch.float32, device=dy_rows.device)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_conv3x3_wgrad_deep_workspace_bytes(int(n_img), int(h), int(w), int(c_in), int(c_out), ctypes.byref(need)),
           'conv3x3_wgrad_deep_workspace')
    ws = _ws(need.value, dy_rows.device)
    _check(lib().pcacc_conv3x3_wgrad_deep_bf16(_dev(dy_rows, torch.bfloat16, 'dy'), _dev(mask, torch.bfloat16, 'mask') if mask is not None else None,
                                               _dev(x_rows, torch.bfloat16, 'x'), _dev(dw), _dev(db),
                                               int(n_img), int(h), int(w), int(c_in), int(c_out), _dev(ws), ctypes.c_size_t(ws.numel()),
                                               _stream()), 'conv3x3_wgrad_deep')
    return dw, db


# ---- fp32x3: the same layers at fp32 accuracy, scaled fp16 hi / lo products on the matrix cores (csrc/conv_split.hip) ---------------------
def absmax256(t):
    """256 partial maxima of |t| (f32, contiguous or any dense layout) on the device: what the fp32x3 kernels derive a tensor's scale from."""
    out = torch.empty((256,), dtype=torch.float32, device=t.device)
    if not t.is_cuda or t.dtype != torch.float32:
        raise NativeError('absmax256: float32 GPU tensor expected, got %s on %s' % (t.dtype, t.device))
    if not (t.is_contiguous() or torch.ops.aten.is_non_overlapping_and_dense(t)):
        t = t.contiguous()                                     # the kernel walks the storage: it must hold exactly the tensor's elements
    _check(lib().pcacc_absmax256(ctypes.c_void_p(t.data_ptr()), _i64(t.numel()), _dev(out), _stream()), 'absmax256')
    return out


_ZERO_POOL = {}


def _zero256(device):
    """A zeroed [256] f32 array for a kernel's output maxima: rows of a [128, 256] block cleared by ONE fill launch (a torch.zeros per
    kernel call was ~100 fill launches per step).  A row is handed out once; the block lives as long as any of its rows."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0)
    blk = _ZERO_POOL.get(key)
    if blk is None or blk[1] >= blk[0].shape[0]:
        blk = [torch.zeros((128, 256), dtype=torch.float32, device=device), 0]
        _ZERO_POOL[key] = blk
    row = blk[0][blk[1]]
    blk[1] += 1
    return row


def conv3x3_split_supported(h, w, c_in, c_out):
    return bool(lib().pcacc_conv3x3_split_supported(int(h), int(w), int(c_in), int(c_out)))


def conv3x3_split_prepare_weights(weight):
    """weight f32 [O,I,3,3] / [O,I,3,3,3] in ANY dense storage order -> ((planes fp16 [2,kt*9,O,I], scale f32 [O]), (planes fp16
    [2,kt*9,I,O], scale f32 [I])): forward and data-gradient form; plane 0 = hi, plane 1 = lo of the row-scaled weight, scale = 1 / row scale."""
    o, i = weight.shape[0], weight.shape[1]
    kt = 3 if weight.dim() == 5 else 1
    if not weight.is_cuda or weight.dtype != torch.float32:
        raise NativeError('conv3x3_split_prepare_weights: weight must be a float32 GPU tensor')
    fwd = torch.empty((2, kt * 9, o, i), dtype=torch.float16, device=weight.device)
    bwd = torch.empty((2, kt * 9, i, o), dtype=torch.float16, device=weight.device)
    sf = torch.empty((o,), dtype=torch.float32, device=weight.device)
    sb = torch.empty((i,), dtype=torch.float32, device=weight.device)
    strides = (ctypes.c_int64 * weight.dim())(*weight.stride())
    _check(lib().pcacc_conv3x3_split_prepare_weights(ctypes.c_void_p(weight.data_ptr()), int(o), int(i), kt, strides, _dev(fwd), _dev(sf),
                                                     _dev(bwd), _dev(sb), _stream()), 'conv3x3_split_prepare_weights')
    return (fwd, sf), (bwd, sb)


def conv3x3_split(x_rows, wps, bias, frames, relu, mask=None, amax=None, want_amax=False, out_mask=None, want_bf16=False):
    """x_rows f32 [n_img,h,w,c_in] contiguous, wps = (planes, scale) from conv3x3_split_prepare_weights -> f32 [n_img,h,w,c_out]; mask as in
    conv3x3 (f32); amax = absmax256(x_rows) when the caller has it already.  want_amax: -> (out, absmax256 array of out), the maxima
    collected by the kernel's epilogue."""
    wp, wscale = wps
    n_img, h, w, c_in = x_rows.shape
    _, taps, c_out, wc_in = wp.shape
    if wc_in != c_in:
        raise NativeError('conv3x3_split: weights prepared for %d input channels, input has %d' % (wc_in, c_in))
    if mask is not None and mask.shape != x_rows.shape:
        raise NativeError('conv3x3_split: mask shape %s != input shape %s' % (tuple(mask.shape), tuple(x_rows.shape)))
    xp = _dev(x_rows, torch.float32, 'x')
    if amax is None:
        amax = absmax256(x_rows)
    out = torch.empty((n_img, h, w, c_out), dtype=torch.float32, device=x_rows.device)
    out_amax = _zero256(x_rows.device) if want_amax else None
    if out_mask is not None:                                   # result zeroed where out_mask <= 0 (no bias / relu with it)
        if bias is not None or relu or tuple(out_mask.shape) != tuple(out.shape):
            raise NativeError('conv3x3_split: out_mask goes with no bias / relu and has the shape of the result')
        _check(lib().pcacc_conv3x3_split_outmask(xp, _dev(amax, torch.float32, 'amax'), _dev(mask, torch.float32, 'mask') if mask is not None else None,
                                                 _dev(wp, torch.float16, 'wp'), _dev(wscale, torch.float32, 'wscale'),
                                                 _dev(out_mask, torch.float32, 'out_mask'), _dev(out), _dev(out_amax) if want_amax else None, int(n_img),
                                                 int(frames), int(h), int(w), int(c_in), int(c_out), taps // 9, _stream()), 'conv3x3_split_outmask')
        return (out, out_amax) if want_amax else out
    if want_bf16:                                              # 'mixed' mode: -> (out, out_amax, bf16 copy of out written by the same epilogue)
        out16 = torch.empty((n_img, h, w, c_out), dtype=torch.bfloat16, device=x_rows.device)
        _check(lib().pcacc_conv3x3_split_dual(xp, _dev(amax, torch.float32, 'amax'), _dev(mask, torch.float32, 'mask') if mask is not None else None,
                                              _dev(wp, torch.float16, 'wp'), _dev(wscale, torch.float32, 'wscale'),
                                              _dev(bias, torch.float32, 'bias') if bias is not None else None,
                                              _dev(out), _dev(out_amax) if want_amax else None, _dev(out16), int(n_img), int(frames), int(h), int(w),
                                              int(c_in), int(c_out), taps // 9, 1 if relu else 0, _stream()), 'conv3x3_split_dual')
        return out, out_amax, out16
    _check(lib().pcacc_conv3x3_split(xp, _dev(amax, torch.float32, 'amax'), _dev(mask, torch.float32, 'mask') if mask is not None else None,
                                     _dev(wp, torch.float16, 'wp'), _dev(wscale, torch.float32, 'wscale'),
                                     _dev(bias, torch.float32, 'bias') if bias is not None else None,
                                     _dev(out), _dev(out_amax) if want_amax else None, int(n_img), int(frames), int(h), int(w), int(c_in), int(c_out),
                                     taps // 9, 1 if relu else 0, _stream()), 'conv3x3_split')
    return (out, out_amax) if want_amax else out


def conv3x3_wgrad_split(dy_rows, x_rows, frames=1, dt=0, mask=None, dy_amax=None, x_amax=None):
    """dy_rows [n_img,h,w,c_out], x_rows [n_img,h,w,c_in] f32 -> (dw [c_out, 9, c_in] f32, db [c_out] f32) for frame tap dt (db is the
    full bias gradient for dt = 0); mask = forward output of the ReLU layer whose gradient dy_rows is (None: no ReLU)."""
    n_img, h, w, c_out = dy_rows.shape
    c_in = x_rows.shape[3]
    dyp, xp = _dev(dy_rows, torch.float32, 'dy'), _dev(x_rows, torch.float32, 'x')
    dy_amax = absmax256(dy_rows) if dy_amax is None else dy_amax
    x_amax = absmax256(x_rows) if x_amax is None else x_amax
    dw = torch.empty((c_out, 9, c_in), dtype=torch.float32, device=dy_rows.device)
    db = torch.empty((c_out,), dtype=torch.float32, device=dy_rows.device)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_conv3x3_wgrad_split_workspace_bytes(int(n_img), int(h), int(w), int(c_in), int(c_out), ctypes.byref(need)),
           'conv3x3_wgrad_split_workspace')
    ws = _ws(need.value, dy_rows.device)
    _check(lib().pcacc_conv3x3_wgrad_split(dyp, _dev(dy_amax, torch.float32, 'dy_amax'), _dev(mask, torch.float32, 'mask') if mask is not None else None,
                                           xp, _dev(x_amax, torch.float32, 'x_amax'), _dev(dw), _dev(db), int(n_img), int(frames), int(dt),
                                           int(h), int(w), int(c_in), int(c_out), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()),
           'conv3x3_wgrad_split')
    return dw, db


def upconv2x2_split_supported(h, w, c_in, c_up):
    return bool(lib().pcacc_upconv2x2_split_supported(int(h), int(w), int(c_in), int(c_up)))


def conv3x3_split_cat(a_rows, b_rows, amax, wps, bias, relu, want_bf16=False):
    """conv3x3_split on cat(a_rows, b_rows) along the channels without the concatenation: a_rows f32 [n,h,w,c_a], b_rows f32 [n,h,w,c_b], both
    contiguous; amax bounds both.  -> (out f32 [n,h,w,c_out], its absmax256 array[, out as bf16])."""
    wp, wscale = wps
    n_img, h, w, c_a = a_rows.shape
    c_in = c_a + b_rows.shape[3]
    _, taps, c_out, wc_in = wp.shape
    if wc_in != c_in or taps != 9 or tuple(b_rows.shape[:3]) != (n_img, h, w):
        raise NativeError('conv3x3_split_cat: weights prepared for %d input channels (%d taps), inputs have %d + %d' % (wc_in, taps, c_a, c_in - c_a))
    out = torch.empty((n_img, h, w, c_out), dtype=torch.float32, device=a_rows.device)
    out16 = torch.empty((n_img, h, w, c_out), dtype=torch.bfloat16, device=a_rows.device) if want_bf16 else None
    out_amax = _zero256(a_rows.device)
    _check(lib().pcacc_conv3x3_split_cat(_dev(a_rows, torch.float32, 'a'), _dev(b_rows, torch.float32, 'b'), int(c_a), _dev(amax, torch.float32, 'amax'),
                                         _dev(wp), _dev(wscale), _opt(bias, torch.float32, 'bias'), _dev(out), _dev(out_amax),
                                         _dev(out16) if out16 is not None else None, int(n_img), int(h), int(w), int(c_in), int(c_out), 1 if relu else 0,
                                         _stream()), 'conv3x3_split_cat')
    return (out, out_amax, out16) if want_bf16 else (out, out_amax)


def upconv2x2_split_prepare_weights(weight):
    """nn.ConvTranspose2d(k=2, s=2).weight f32 [c_in, c_up, 2, 2] (any dense layout) -> ((planes fp16 [2, 4 c_up, c_in], scale [4 c_up]),
    (planes fp16 [2, c_in, 4 c_up], scale [c_in])): forward and data-gradient form."""
    ci, cu = weight.shape[0], weight.shape[1]
    if not weight.is_cuda or weight.dtype != torch.float32 or tuple(weight.shape[2:]) != (2, 2):
        raise NativeError('upconv2x2_split_prepare_weights: float32 GPU weight [c_in, c_up, 2, 2] expected')
    fwd = torch.empty((2, 4 * cu, ci), dtype=torch.float16, device=weight.device)
    bwd = torch.empty((2, ci, 4 * cu), dtype=torch.float16, device=weight.device)
    sf = torch.empty((4 * cu,), dtype=torch.float32, device=weight.device)
    sb = torch.empty((ci,), dtype=torch.float32, device=weight.device)
    strides = (ctypes.c_int64 * 4)(*weight.stride())
    _check(lib().pcacc_upconv2x2_split_prepare_weights(ctypes.c_void_p(weight.data_ptr()), int(ci), int(cu), strides, _dev(fwd), _dev(sf), _dev(bwd),
                                                       _dev(sb), _stream()), 'upconv2x2_split_prepare_weights')
    return (fwd, sf), (bwd, sb)


def upconv2x2_bf16_supported(c_in, c_up):
    return bool(lib().pcacc_upconv2x2_bf16_supported(int(c_in), int(c_up)))


def upconv2x2_bf16_prepare_weights(weight):
    """nn.ConvTranspose2d(k=2, s=2).weight f32 [c_in, c_up, 2, 2] (any dense layout) -> (bf16 [4 c_up, c_in], bf16 [c_in, 4 c_up])."""
    ci, cu = weight.shape[0], weight.shape[1]
    if not weight.is_cuda or weight.dtype != torch.float32 or tuple(weight.shape[2:]) != (2, 2):
        raise NativeError('upconv2x2_bf16_prepare_weights: float32 GPU weight [c_in, c_up, 2, 2] expected')
    fwd = torch.empty((4 * cu, ci), dtype=torch.bfloat16, device=weight.device)
    bwd = torch.empty((ci, 4 * cu), dtype=torch.bfloat16, device=weight.device)
    strides = (ctypes.c_int64 * 4)(*weight.stride())
    _check(lib().pcacc_upconv2x2_bf16_prepare_weights(ctypes.c_void_p(weight.data_ptr()), int(ci), int(cu), strides, _dev(fwd), _dev(bwd), _stream()),
           'upconv2x2_bf16_prepare_weights')
    return fwd, bwd


def _rows_pitch(rows, what):
    """rows: a [n, h, w, c] bf16 view whose pixels are `pitch` elements apart (a channel slice of a wider channels-last map, or a dense map) ->
    pitch, or None when the view is anything else."""
    n, h, w, c = rows.shape
    st = rows.stride()
    pitch = st[2]
    if st[3] != 1 or pitch < c or st[1] != w * pitch or st[0] != h * w * pitch or pitch % 8 or rows.data_ptr() % 16:
        return None
    return pitch


def upconv2x2_bf16(x_rows, wp, bias, direction):
    """direction 0: x_rows bf16 [n,h,w,c_in], wp = forward form -> bf16 [n,2h,2w,c_up] (+ bias f32 [c_up]); direction 1: x_rows = dy [n,2h,2w,c_up]
    (a dense map or a channel slice of a wider one), wp = data-gradient form -> bf16 [n,h,w,c_in]."""
    if x_rows.dtype != torch.bfloat16 or not x_rows.is_cuda:
        raise NativeError('upconv2x2_bf16: bf16 GPU rows expected')
    pitch = _rows_pitch(x_rows, 'x')
    if pitch is None:
        x_rows = x_rows.contiguous()
        pitch = x_rows.shape[3]
    n = x_rows.shape[0]
    if direction == 0:
        h, w, c_in = x_rows.shape[1:]
        c_up = wp.shape[0] // 4
        out = torch.empty((n, 2 * h, 2 * w, c_up), dtype=torch.bfloat16, device=x_rows.device)
        out_pitch = c_up
    else:
        h, w, c_up = x_rows.shape[1] // 2, x_rows.shape[2] // 2, x_rows.shape[3]
        c_in = wp.shape[0]
        out = torch.empty((n, h, w, c_in), dtype=torch.bfloat16, device=x_rows.device)
        out_pitch = c_in
    _check(lib().pcacc_upconv2x2_bf16(ctypes.c_void_p(x_rows.data_ptr()), _dev(wp, torch.bfloat16, 'wp'), _opt(bias, torch.float32, 'bias') if direction == 0 else None,
                                      _dev(out), int(n), int(h), int(w), int(c_in), int(c_up), int(direction), int(pitch), int(out_pitch), _stream()),
           'upconv2x2_bf16')
    return out


def upconv2x2_bf16_wgrad(dy_rows, x_rows, want_bias=True, like=None):
    """dy_rows bf16 [n,2h,2w,c_up] (dense or a channel slice), x_rows bf16 [n,h,w,c_in] -> (dw f32 [c_in,c_up,2,2] in the memory layout of `like`
    (the weight; default contiguous), db f32 [c_up] or None)."""
    n, h, w, c_in = x_rows.shape
    c_up = dy_rows.shape[3]
    dp, xp = _rows_pitch(dy_rows, 'dy'), _rows_pitch(x_rows, 'x')
    if dp is None:
        dy_rows, dp = dy_rows.contiguous(), c_up
    if xp is None:
        x_rows, xp = x_rows.contiguous(), c_in
    if dy_rows.dtype != torch.bfloat16 or x_rows.dtype != torch.bfloat16 or tuple(dy_rows.shape[:3]) != (n, 2 * h, 2 * w):
        raise NativeError('upconv2x2_bf16_wgrad: bf16 rows [n,2h,2w,c_up] and [n,h,w,c_in] expected')
    if like is not None and tuple(like.shape) == (c_in, c_up, 2, 2):
        dw = torch.empty_strided(like.shape, like.stride(), dtype=torch.float32, device=x_rows.device)
    else:
        dw = torch.empty((c_in, c_up, 2, 2), dtype=torch.float32, device=x_rows.device)
    dws = (ctypes.c_int64 * 4)(*dw.stride())
    db = torch.empty((c_up,), dtype=torch.float32, device=x_rows.device) if want_bias else None
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_upconv2x2_bf16_wgrad_workspace_bytes(int(n), int(h), int(w), int(c_in), int(c_up), ctypes.byref(need)), 'upconv2x2_bf16_wgrad_workspace')
    ws = _ws(need.value, x_rows.device)
    _check(lib().pcacc_upconv2x2_bf16_wgrad(ctypes.c_void_p(dy_rows.data_ptr()), int(dp), ctypes.c_void_p(x_rows.data_ptr()), int(xp), ctypes.c_void_p(dw.data_ptr()),
                                            dws, _dev(db) if db is not None else None, int(n), int(h), int(w), int(c_in), int(c_up), _dev(ws),
                                            ctypes.c_size_t(ws.numel()), _stream()), 'upconv2x2_bf16_wgrad')
    return dw, db


def prepare_weights_batch(jobs, n_jobs, total_blocks):
    """jobs: int64 GPU tensor [n_jobs, 16] (include/pcacc.h: pcacc_prepare_weights_batch) -- every prepared form of every listed weight, one launch."""
    if not jobs.is_cuda or jobs.dtype != torch.int64 or not jobs.is_contiguous() or jobs.numel() != 16 * n_jobs:
        raise NativeError('prepare_weights_batch: jobs must be a contiguous int64 GPU tensor [n_jobs, 16]')
    _check(lib().pcacc_prepare_weights_batch(ctypes.c_void_p(jobs.data_ptr()), int(n_jobs), int(total_blocks), _stream()), 'prepare_weights_batch')


def upconv2x2_split(x_rows, amax, wps, bias, direction, want_bf16=False, into=None):
    """direction 0: x_rows f32 [n,h,w,c_in] -> ([n,2h,2w,c_up], its absmax256 array); direction 1: x_rows = dy [n,2h,2w,c_up] -> ([n,h,w,c_in], amax).
    want_bf16 (direction 0): -> (out, amax, bf16 copy of out written by the same epilogue)."""
    wp, wscale = wps
    n = x_rows.shape[0]
    if direction == 0:
        h, w, c_in = x_rows.shape[1:]
        c_up = wp.shape[1] // 4
        out = torch.empty((n, 2 * h, 2 * w, c_up), dtype=torch.float32, device=x_rows.device) if into is None else None
    else:
        h, w, c_up = x_rows.shape[1] // 2, x_rows.shape[2] // 2, x_rows.shape[3]
        c_in = wp.shape[1]
        out = torch.empty((n, h, w, c_in), dtype=torch.float32, device=x_rows.device)
    out_amax = _zero256(x_rows.device)
    if want_bf16:
        if direction != 0:
            raise NativeError('upconv2x2_split: the bf16 second output goes with direction 0')
        pitch = 0
        if into is not None:                                   # (f32 buffer, bf16 buffer) [n,2h,2w,wide]: the results are their first c_up channels
            out, out16 = into
            pitch = out.shape[3]
            if tuple(out.shape) != (n, 2 * h, 2 * w, pitch) or tuple(out16.shape) != tuple(out.shape) or not out.is_contiguous() or not out16.is_contiguous():
                raise NativeError('upconv2x2_split: concatenation buffers must be contiguous [n,2h,2w,wide] tensors of one shape')
        else:
            out16 = torch.empty(out.shape, dtype=torch.bfloat16, device=x_rows.device)
        _check(lib().pcacc_upconv2x2_split_dual(_dev(x_rows, torch.float32, 'x'), _dev(amax, torch.float32, 'amax'), _dev(wp, torch.float16, 'wp'),
                                                _dev(wscale, torch.float32, 'wscale'), _opt(bias, torch.float32, 'bias'), _dev(out), _dev(out_amax),
                                                _dev(out16), int(n), int(h), int(w), int(c_in), int(c_up), int(pitch), _stream()), 'upconv2x2_split_dual')
        return out, out_amax, out16
    _check(lib().pcacc_upconv2x2_split(_dev(x_rows, torch.float32, 'x'), _dev(amax, torch.float32, 'amax'), _dev(wp, torch.float16, 'wp'),
                                       _dev(wscale, torch.float32, 'wscale'), _opt(bias, torch.float32, 'bias'), _dev(out), _dev(out_amax), int(n),
                                       int(h), int(w), int(c_in), int(c_up), int(direction), _stream()), 'upconv2x2_split')
    return out, out_amax


def upconv2x2_wgrad_split(dy_rows, dy_amax, x_rows, x_amax):
    """dy [n,2h,2w,c_up], x [n,h,w,c_in] f32 -> (dw [c_in, c_up, 2, 2] f32 (a permuted view), db [c_up] f32)."""
    n, h, w, c_in = x_rows.shape
    c_up = dy_rows.shape[3]
    dw = torch.empty((4 * c_up, c_in), dtype=torch.float32, device=x_rows.device)
    db4 = torch.empty((4 * c_up,), dtype=torch.float32, device=x_rows.device)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_upconv2x2_wgrad_split_workspace_bytes(int(n), int(h), int(w), int(c_in), int(c_up), ctypes.byref(need)), 'upconv2x2_wgrad_workspace')
    ws = _ws(need.value, x_rows.device)
    _check(lib().pcacc_upconv2x2_wgrad_split(_dev(dy_rows, torch.float32, 'dy'), _dev(dy_amax, torch.float32, 'dy_amax'), _dev(x_rows, torch.float32, 'x'),
                                             _dev(x_amax, torch.float32, 'x_amax'), _dev(dw), _dev(db4), int(n), int(h), int(w), int(c_in), int(c_up),
                                             _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'upconv2x2_wgrad_split')
    return dw.view(2, 2, c_up, c_in).permute(3, 2, 0, 1), db4.view(4, c_up).sum(0)


def head_conv3x3_supported(c_in, c_out):
    return bool(lib().pcacc_head_conv3x3_supported(int(c_in), int(c_out)))


def head_conv3x3_forward(x_rows, weight, bias):
    """x_rows [n,h,w,c_in] f32 / bf16, weight f32 [c_out,c_in,3,3] (any dense layout), c_out <= 4 -> y [n,h,w,c_out] f32."""
    n, h, w, c_in = x_rows.shape
    c_out = weight.shape[0]
    y = torch.empty((n, h, w, c_out), dtype=torch.float32, device=x_rows.device)
    strides = (ctypes.c_int64 * 4)(*weight.stride())
    _check(lib().pcacc_head_conv3x3_forward(_dev(x_rows, None, 'x'), _dtype_code(x_rows), ctypes.c_void_p(weight.data_ptr()), strides,
                                            _opt(bias, torch.float32, 'bias'), _dev(y), int(n), int(h), int(w), int(c_in), int(c_out), _stream()),
           'head_conv3x3_forward')
    return y


def head_conv3x3_dgrad(dy_rows, weight, c_in, out_dtype):
    n, h, w, c_out = dy_rows.shape
    dx = torch.empty((n, h, w, c_in), dtype=out_dtype, device=dy_rows.device)
    strides = (ctypes.c_int64 * 4)(*weight.stride())
    _check(lib().pcacc_head_conv3x3_dgrad(_dev(dy_rows, torch.float32, 'dy'), ctypes.c_void_p(weight.data_ptr()), strides, _dev(dx), _dtype_code(dx),
                                          int(n), int(h), int(w), int(c_in), int(c_out), _stream()), 'head_conv3x3_dgrad')
    return dx


def head_conv3x3_wgrad(dy_rows, x_rows, want_bias=True):
    n, h, w, c_out = dy_rows.shape
    c_in = x_rows.shape[3]
    dw = torch.empty((c_out, c_in, 3, 3), dtype=torch.float32, device=dy_rows.device)
    db = torch.empty((c_out,), dtype=torch.float32, device=dy_rows.device) if want_bias else None
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_head_conv3x3_wgrad_workspace_bytes(int(n), int(h), int(w), int(c_in), int(c_out), ctypes.byref(need)), 'head_conv3x3_wgrad_workspace')
    ws = _ws(need.value, dy_rows.device)
    _check(lib().pcacc_head_conv3x3_wgrad(_dev(dy_rows, torch.float32, 'dy'), _dev(x_rows, None, 'x'), _dtype_code(x_rows), _dev(dw),
                                          _dev(db) if db is not None else None, int(n), int(h), int(w), int(c_in), int(c_out), _dev(ws),
                                          ctypes.c_size_t(ws.numel()), _stream()), 'head_conv3x3_wgrad')
    return dw, db


def rows_split_supported(k, n):
    """Widths the fp32x3 row kernels take (csrc/mlp_split.hip)."""
    return k in (32, 64, 128) and n in (32, 64, 128)


def rows_linear_split(x, x_amax, w, bias=None, residual=None, pre_relu=False, post_relu=False, in_mask=None, out_mask=None, want_amax=False,
                      want_bf16=False):
    """rows_linear on f32 rows at fp32 accuracy on the matrix cores; x_amax = absmax256(x) (of the tensor before ReLU / mask: an upper bound
    is what the scale needs).  want_amax: -> (y, absmax256 array of y) from the kernel's store phase.  want_bf16 ('mixed' mode):
    -> (y, y_amax, y as bf16 from the same epilogue)."""
    rows, k = x.shape
    n = w.shape[0]
    y = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    y_amax = _zero256(x.device) if want_amax or want_bf16 else None
    flags = (1 if pre_relu else 0) | (2 if post_relu else 0)
    if want_bf16:
        y16 = torch.empty((rows, n), dtype=torch.bfloat16, device=x.device)
        _check(lib().pcacc_rows_linear_split_dual(_dev(x, torch.float32, 'x'), _dev(x_amax, torch.float32, 'x_amax'), _opt(in_mask, torch.float32, 'in_mask'),
                                                  _dev(w, torch.float32, 'w'), _opt(bias, torch.float32, 'bias'), _opt(residual, torch.float32, 'residual'),
                                                  _opt(out_mask, torch.float32, 'out_mask'), _dev(y), _dev(y16), _dev(y_amax), _i64(rows), int(k),
                                                  int(n), flags, _stream()), 'rows_linear_split_dual')
        return y, y_amax, y16
    _check(lib().pcacc_rows_linear_split(_dev(x, torch.float32, 'x'), _dev(x_amax, torch.float32, 'x_amax'), _opt(in_mask, torch.float32, 'in_mask'),
                                         _dev(w, torch.float32, 'w'), _opt(bias, torch.float32, 'bias'), _opt(residual, torch.float32, 'residual'),
                                         _opt(out_mask, torch.float32, 'out_mask'), _dev(y), _opt(y_amax, torch.float32, 'y_amax'), _i64(rows), int(k),
                                         int(n), flags, _stream()), 'rows_linear_split')
    return (y, y_amax) if want_amax else y


def rows_linear_few_dual(x, w, bias=None, residual=None, pre_relu=False, post_relu=False):
    """rows_linear on f32 rows with k <= 9 inputs in plain fp32 arithmetic -> (y f32, absmax256 array of y, y as bf16), one store phase
    ('mixed' mode: the pillar encoder's position layer)."""
    rows, k = x.shape
    n = w.shape[0]
    y = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    y16 = torch.empty((rows, n), dtype=torch.bfloat16, device=x.device)
    y_amax = _zero256(x.device)
    flags = (1 if pre_relu else 0) | (2 if post_relu else 0)
    _check(lib().pcacc_rows_linear_few_dual(_dev(x, torch.float32, 'x'), _dev(w, torch.float32, 'w'), _opt(bias, torch.float32, 'bias'),
                                            _opt(residual, torch.float32, 'residual'), _dev(y), _dev(y16), _dev(y_amax), _i64(rows), int(k), int(n),
                                            flags, _stream()), 'rows_linear_few_dual')
    return y, y_amax, y16


def rows_wgrad_split(dy, dy_amax, x, x_amax, dy_mask=None, x_relu=False, split=False):
    """rows_wgrad on f32 rows at fp32 accuracy on the matrix cores (split: (dW [n,k], db [n]) contiguous)."""
    rows, n = dy.shape
    k = x.shape[1]
    out = torch.empty((n, k + 1), dtype=torch.float32, device=dy.device)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_rows_wgrad_split_workspace_bytes(_i64(rows), int(k), int(n), ctypes.byref(need)), 'rows_wgrad_split_workspace')
    ws = _ws(need.value, dy.device)
    _check(lib().pcacc_rows_wgrad_split(_dev(dy, torch.float32, 'dy'), _dev(dy_amax, torch.float32, 'dy_amax'), _opt(dy_mask, torch.float32, 'dy_mask'),
                                        _dev(x, torch.float32, 'x'), _dev(x_amax, torch.float32, 'x_amax'), (1 if x_relu else 0) | (2 if split else 0),
                                        _i64(rows), int(k), int(n), _dev(out), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'rows_wgrad_split')
    return _split_aug(out, n, k, split, rows > 0)


def rows_linear_cat_split(xa, xa_amax, xb, xb_amax, b_index, w, bias=None, residual=None, pre_relu=False, post_relu=False, want_amax=False):
    """y = post(pre(cat(xa, xb[b_index])) @ w^T + bias + residual), f32 rows, the concatenation read in place (fp32x3)."""
    rows, ka = xa.shape
    k, n = ka + xb.shape[1], w.shape[0]
    y = torch.empty((rows, n), dtype=torch.float32, device=xa.device)
    y_amax = _zero256(xa.device) if want_amax else None
    flags = (1 if pre_relu else 0) | (2 if post_relu else 0)
    _check(lib().pcacc_rows_linear_cat_split(_dev(xa, torch.float32, 'xa'), _dev(xa_amax, torch.float32, 'xa_amax'), _dev(xb, torch.float32, 'xb'),
                                             _dev(xb_amax, torch.float32, 'xb_amax'), _opt(b_index, torch.int32, 'b_index'), int(ka), None,
                                             _dev(w, torch.float32, 'w'), _opt(bias, torch.float32, 'bias'), _opt(residual, torch.float32, 'residual'),
                                             None, None, _dev(y), None, 0, _opt(y_amax, torch.float32, 'y_amax'), _i64(rows), int(k), int(n), flags,
                                             _stream()), 'rows_linear_cat_split')
    return (y, y_amax) if want_amax else y


def rows_linear_cat_backward_split(gy, gy_amax, w_t, dy_mask, xa, xb, b_index, pre_relu):
    """(d xa [rows,ka], d of the gathered xb rows [rows,kb]) = split of (gy masked where dy_mask <= 0) @ w_t^T, each masked where the
    forward input was <= 0 when pre_relu; f32 rows (fp32x3)."""
    rows, n = gy.shape
    ka, kb = xa.shape[1], xb.shape[1]
    ga = torch.empty((rows, ka), dtype=torch.float32, device=gy.device)
    gb = torch.empty((rows, kb), dtype=torch.float32, device=gy.device)
    g_amax = _zero256(gy.device)                                               # of both pieces together: an upper bound for either
    _check(lib().pcacc_rows_linear_cat_split(_dev(gy, torch.float32, 'gy'), _dev(gy_amax, torch.float32, 'gy_amax'), None, None,
                                             _opt(b_index, torch.int32, 'b_index'), 0, _opt(dy_mask, torch.float32, 'dy_mask'),
                                             _dev(w_t, torch.float32, 'w_t'), None, None,
                                             _dev(xa, torch.float32, 'xa') if pre_relu else None, _dev(xb, torch.float32, 'xb') if pre_relu else None,
                                             _dev(ga), _dev(gb), int(ka), _dev(g_amax), _i64(rows), int(n), int(ka + kb), 0, _stream()),
           'rows_linear_cat_backward_split')
    return ga, gb, g_amax


def rows_wgrad_cat_split(dy, dy_amax, xa, xa_amax, xb, xb_amax, b_index, dy_mask=None, x_relu=False, split=False):
    """[n, k+1] f32 weight (+ bias) gradient for x = cat(xa, xb[b_index]), f32 rows (fp32x3; split: see rows_wgrad)."""
    rows, n = dy.shape
    ka = xa.shape[1]
    k = ka + xb.shape[1]
    out = torch.empty((n, k + 1), dtype=torch.float32, device=dy.device)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_rows_wgrad_split_workspace_bytes(_i64(rows), int(k), int(n), ctypes.byref(need)), 'rows_wgrad_split_workspace')
    ws = _ws(need.value, dy.device)
    _check(lib().pcacc_rows_wgrad_cat_split(_dev(dy, torch.float32, 'dy'), _dev(dy_amax, torch.float32, 'dy_amax'), _opt(dy_mask, torch.float32, 'dy_mask'),
                                            _dev(xa, torch.float32, 'xa'), _dev(xa_amax, torch.float32, 'xa_amax'), _dev(xb, torch.float32, 'xb'),
                                            _dev(xb_amax, torch.float32, 'xb_amax'), _opt(b_index, torch.int32, 'b_index'), int(ka),
                                            (1 if x_relu else 0) | (2 if split else 0), _i64(rows), int(k), int(n), _dev(out), _dev(ws),
                                            ctypes.c_size_t(ws.numel()), _stream()), 'rows_wgrad_cat_split')
    return _split_aug(out, n, k, split, rows > 0)


def upload_small(values, dtype, device):
    """A small host list / numpy array / CPU tensor -> device tensor of `dtype` without a blocking copy (see pcacc_upload_words).
    CPU devices get a plain tensor."""
    host = torch.as_tensor(values, dtype=dtype).contiguous()
    if device.type != 'cuda':
        return host
    out = torch.empty(host.shape, dtype=dtype, device=device)
    nbytes = host.numel() * host.element_size()
    if nbytes == 0:
        return out
    if nbytes % 4:
        raise NativeError('upload_small: byte size must be a multiple of 4')
    _check(lib().pcacc_upload_words(ctypes.c_void_p(host.data_ptr()), _i64(nbytes // 4), _dev(out), _stream()), 'upload_words')
    return out


def bilinear_gather_backward_sorted(grad_out, shape, points, map_idx, x_scale, y_scale, out_dtype=torch.float32):
    """Gradient of bilinear_gather w.r.t. the map, atomic-free (see include/pcacc.h): grad_out [k,c] f32|bf16 ->
    [n_maps,h,w,c] in out_dtype."""
    n_maps, h, w, c = shape
    k = points.shape[0]
    dev = grad_out.device
    out = torch.empty((n_maps, h, w, c), dtype=out_dtype, device=dev)
    if k == 0:
        return out.zero_()
    cells = torch.empty((k,), dtype=torch.int32, device=dev)
    _check(lib().pcacc_bilinear_base_cells(_dev(points, torch.float32, 'points'), _dev(map_idx, torch.int32, 'map_idx'), _i64(k),
                                           int(n_maps), int(h), int(w), ctypes.c_float(x_scale), ctypes.c_float(y_scale),
                                           _dev(cells), _stream()), 'bilinear_base_cells')
    offs, order = csr_build(cells, n_maps * h * w + 1)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_bilinear_sorted_workspace_bytes(_i64(k), int(c), _dtype_code(grad_out), ctypes.byref(need)), 'bilinear_sorted_workspace')
    ws = _ws(need.value, dev)
    _check(lib().pcacc_bilinear_gather_backward_sorted(_dev(grad_out, None, 'grad_out'), _dtype_code(grad_out), int(n_maps), int(h),
                                                       int(w), int(c), _dev(points, torch.float32, 'points'), _dev(offs, torch.int32),
                                                       _dev(order, torch.int32), _i64(k), ctypes.c_float(x_scale), ctypes.c_float(y_scale),
                                                       _dev(out), _dtype_code(out), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()),
           'bilinear_gather_backward_sorted')
    return out


def prep_points(points, tsfm12, noise, noise_scale, scale, crop_xy, z_min, z_max, remove_ground, ground_z):
    """points [m,3] f64 -> (augmented points [m,3] f64, keep [m] u8); see include/pcacc.h (D1)."""
    m = points.shape[0]
    out = torch.empty_like(points)
    keep = torch.empty((m,), dtype=torch.uint8, device=points.device)
    _check(lib().pcacc_prep_points(_dev(points, torch.float64, 'points'), _dev(tsfm12, torch.float64, 'tsfm12') if tsfm12 is not None else None,
                                   _dev(noise, torch.float64, 'noise') if noise is not None else None, ctypes.c_double(noise_scale),
                                   ctypes.c_double(scale), ctypes.c_double(crop_xy), ctypes.c_double(z_min), ctypes.c_double(z_max),
                                   1 if remove_ground else 0, ctypes.c_double(ground_z), _i64(m), _dev(out), _dev(keep), _stream()),
           'prep_points')
    return out, keep


def _sinkhorn_ws(P, k, dev):
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_sinkhorn_train_workspace_bytes(int(P), int(k), ctypes.byref(need)), 'sinkhorn_train_workspace')
    return _ws(need.value, dev)


def sinkhorn_forward(log_alpha, n_iters):
    """log_alpha [P,k,k] f32 -> (log_perm [P,k,k], lse_rows, lse_cols [n_iters,P,k]); see include/pcacc.h."""
    P, k, _ = log_alpha.shape
    dev = log_alpha.device
    out = torch.empty_like(log_alpha)
    lr = torch.empty((n_iters, P, k), dtype=torch.float32, device=dev)
    lc = torch.empty((n_iters, P, k), dtype=torch.float32, device=dev)
    ws = _sinkhorn_ws(P, k, dev)
    _check(lib().pcacc_sinkhorn_forward(_dev(log_alpha, torch.float32, 'log_alpha'), int(P), int(k), int(n_iters), _dev(out), _dev(lr),
                                        _dev(lc), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'sinkhorn_forward')
    return out, lr, lc


def sinkhorn_backward(grad_log_perm, log_alpha, lse_rows, lse_cols):
    P, k, _ = log_alpha.shape
    n_iters = lse_rows.shape[0]
    g = torch.empty_like(log_alpha)
    ws = _sinkhorn_ws(P, k, log_alpha.device)
    _check(lib().pcacc_sinkhorn_backward(_dev(grad_log_perm, torch.float32, 'grad'), _dev(log_alpha, torch.float32, 'log_alpha'),
                                         _dev(lse_rows, torch.float32), _dev(lse_cols, torch.float32), int(P), int(k), int(n_iters),
                                         _dev(g), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'sinkhorn_backward')
    return g


# ---------------------------------------------------------------------------------------------------
def _opt(t, dtype, what):
    return _dev(t, dtype, what) if t is not None else None


def seg_loss_forward(logits, plane, labels, rows, n):
    """L1 (include/pcacc.h): logits = the whole f32 / bf16 logit tensor ([n_total,2] rows when plane == 0, NCHW planes of
    `plane` cells otherwise), labels [n_total] i64, rows [n] i64 or None.  Returns (loss [2] f32 = cross entropy, Lovasz;
    metric [4,2] f64; lovasz_grad [2,n] f32; saved [8] f32) -- the last two are for seg_loss_backward."""
    dev = logits.device
    loss = torch.empty((2,), dtype=torch.float32, device=dev)
    metric = torch.empty((4, 2), dtype=torch.float64, device=dev)
    lov = torch.empty((2, n), dtype=torch.float32, device=dev)
    saved = torch.zeros((8,), dtype=torch.float32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_seg_loss_workspace_bytes(_i64(n), ctypes.byref(need)), 'seg_loss_workspace')
    ws = _ws(need.value, dev)
    _check(lib().pcacc_seg_loss_forward(_dev(logits, None, 'logits'), _dtype_code(logits), _i64(plane), _dev(labels, torch.int64, 'labels'),
                                        _opt(rows, torch.int64, 'rows'), _i64(n), _dev(loss), _dev(metric), _dev(lov), _dev(saved), _dev(ws),
                                        ctypes.c_size_t(ws.numel()), _stream()), 'seg_loss_forward')
    return loss, metric, lov, saved


def seg_loss_backward(logits, plane, labels, rows, n, lovasz_grad, saved, grad_bce, grad_lovasz):
    """Gradient of grad_bce * cross entropy + grad_lovasz * Lovasz w.r.t. the whole logit tensor (0 outside the rows)."""
    grad = torch.empty_like(logits)
    _check(lib().pcacc_seg_loss_backward(_dev(logits, None, 'logits'), _dtype_code(logits), _i64(plane), _dev(labels, torch.int64, 'labels'),
                                         _opt(rows, torch.int64, 'rows'), _i64(n), _i64(logits.numel() // 2), _dev(lovasz_grad, torch.float32),
                                         _dev(saved, torch.float32), _opt(grad_bce, torch.float32, 'grad_bce'),
                                         _opt(grad_lovasz, torch.float32, 'grad_lovasz'), _dev(grad), _stream()), 'seg_loss_backward')
    return grad


def offset_loss_forward(points, time_indice, inst_labels, label_base, ego_motion, inst_motion, n_frames, transformed_points, offset_est, rows):
    """L2 (include/pcacc.h): returns (out [3] f32 = L1 term, direction term, mean L2 error; offset_gt [m,2] f32)."""
    dev = points.device
    n, k = points.shape[0], inst_motion.shape[0]
    m = rows.shape[0] if rows is not None else n
    out = torch.empty((3,), dtype=torch.float32, device=dev)
    gt = torch.empty((m, 2), dtype=torch.float32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_offset_loss_workspace_bytes(_i64(m), _i64(k), ctypes.byref(need)), 'offset_loss_workspace')
    ws = _ws(need.value, dev)
    _check(lib().pcacc_offset_loss_forward(_dev(points, torch.float32, 'points'), _dev(time_indice, torch.int64, 'time_indice'),
                                           _dev(inst_labels, torch.int64, 'inst_labels'), _dev(label_base, torch.int64, 'label_base'),
                                           _dev(ego_motion, torch.float32, 'ego_motion'), _dev(inst_motion, torch.float32, 'inst_motion'),
                                           int(n_frames), _i64(n), _i64(k), _dev(transformed_points, torch.float32, 'transformed_points'),
                                           _dev(offset_est, torch.float32, 'offset_est'), _opt(rows, torch.int64, 'rows'), _i64(m), _dev(out),
                                           _dev(gt), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'offset_loss_forward')
    return out, gt


def offset_loss_backward(offset_gt, offset_est, rows, grad_norm, grad_dir):
    grad = torch.empty_like(offset_est)
    m = offset_gt.shape[0]
    _check(lib().pcacc_offset_loss_backward(_dev(offset_gt, torch.float32), _dev(offset_est, torch.float32, 'offset_est'),
                                            _opt(rows, torch.int64, 'rows'), _i64(m), _i64(offset_est.shape[0]),
                                            _opt(grad_norm, torch.float32, 'grad_norm'), _opt(grad_dir, torch.float32, 'grad_dir'), _dev(grad),
                                            _stream()), 'offset_loss_backward')
    return grad


def frames_max(x):
    """x [S,T,...] f32 / bf16 contiguous -> (max over T [S,...], winning frame u8 [S,...]); include/pcacc.h A9."""
    S, T = x.shape[0], x.shape[1]
    plane = x[0, 0].numel()
    out = torch.empty((S,) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    arg = torch.empty((S,) + tuple(x.shape[2:]), dtype=torch.uint8, device=x.device)
    _check(lib().pcacc_frames_max(_dev(x, None, 'x'), _dtype_code(x), _i64(S), int(T), _i64(plane), _dev(out), _dev(arg), _stream()), 'frames_max')
    return out, arg


def frames_max_backward(grad_out, arg, frames):
    S = grad_out.shape[0]
    plane = grad_out[0].numel()
    g = torch.empty((S, frames) + tuple(grad_out.shape[1:]), dtype=grad_out.dtype, device=grad_out.device)
    _check(lib().pcacc_frames_max_backward(_dev(grad_out, None, 'grad_out'), _dev(arg, torch.uint8), _dtype_code(grad_out), _i64(S), int(frames),
                                           _i64(plane), _dev(g), _stream()), 'frames_max_backward')
    return g


def rows_linear_cat_supported(ka, kb, n):
    """Two-piece rows are a bf16 matrix-core path only (include/pcacc.h: pcacc_rows_linear_cat_bf16)."""
    return (ka + kb) in (32, 64, 128) and n in (32, 64, 128) and ka % 8 == 0 and kb % 8 == 0


def rows_linear_cat(xa, xb, b_index, w, bias=None, residual=None, pre_relu=False, post_relu=False):
    """y = post(pre(cat(xa, xb[b_index])) @ w^T + bias + residual), bf16 rows, the concatenation read in place."""
    rows, ka = xa.shape
    k, n = ka + xb.shape[1], w.shape[0]
    y = torch.empty((rows, n), dtype=torch.bfloat16, device=xa.device)
    flags = (1 if pre_relu else 0) | (2 if post_relu else 0)
    _check(lib().pcacc_rows_linear_cat_bf16(_dev(xa, torch.bfloat16, 'xa'), _dev(xb, torch.bfloat16, 'xb'), _opt(b_index, torch.int32, 'b_index'),
                                            int(ka), None, _dev(w, torch.float32, 'w'), _opt(bias, torch.float32, 'bias'),
                                            _opt(residual, torch.bfloat16, 'residual'), None, None, _dev(y), None, 0, _i64(rows), int(k), int(n),
                                            flags, _stream()), 'rows_linear_cat')
    return y


def rows_linear_cat_backward(gy, w_t, dy_mask, xa, xb, b_index, pre_relu):
    """(d xa [rows,ka], d of the gathered xb rows [rows,kb]) = split of (gy masked where dy_mask <= 0) @ w_t^T, each masked where the
    forward input was <= 0 when pre_relu.  w_t = W^T [k,n] f32."""
    rows, n = gy.shape
    ka, kb = xa.shape[1], xb.shape[1]
    ga = torch.empty((rows, ka), dtype=torch.bfloat16, device=gy.device)
    gb = torch.empty((rows, kb), dtype=torch.bfloat16, device=gy.device)
    _check(lib().pcacc_rows_linear_cat_bf16(_dev(gy, torch.bfloat16, 'gy'), None, _opt(b_index, torch.int32, 'b_index'), 0,
                                            _opt(dy_mask, torch.bfloat16, 'dy_mask'), _dev(w_t, torch.float32, 'w_t'), None, None,
                                            _dev(xa, torch.bfloat16, 'xa') if pre_relu else None, _dev(xb, torch.bfloat16, 'xb') if pre_relu else None,
                                            _dev(ga), _dev(gb), int(ka), _i64(rows), int(n), int(ka + kb), 0, _stream()), 'rows_linear_cat_backward')
    return ga, gb


def rows_wgrad_cat(dy, xa, xb, b_index, dy_mask=None, x_relu=False, split=False):
    """[n, k+1] f32 weight (+ bias) gradient for x = cat(xa, xb[b_index]), bf16 rows (split: see rows_wgrad)."""
    rows, n = dy.shape
    ka = xa.shape[1]
    k = ka + xb.shape[1]
    out = torch.empty((n, k + 1), dtype=torch.float32, device=dy.device)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_rows_wgrad_bf16_workspace_bytes(_i64(rows), int(k), int(n), ctypes.byref(need)), 'rows_wgrad_bf16_workspace')
    ws = _ws(need.value, dy.device)
    _check(lib().pcacc_rows_wgrad_cat_bf16(_dev(dy, torch.bfloat16, 'dy'), _opt(dy_mask, torch.bfloat16, 'dy_mask'), _dev(xa, torch.bfloat16, 'xa'),
                                           _dev(xb, torch.bfloat16, 'xb'), _opt(b_index, torch.int32, 'b_index'), int(ka),
                                           (1 if x_relu else 0) | (2 if split else 0),
                                           _i64(rows), int(k), int(n), _dev(out), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'rows_wgrad_cat')
    return _split_aug(out, n, k, split, rows > 0)


def svd3(a):
    """a [n,3,3] f32 -> (u [n,3,3], s [n,3], v [n,3,3]) with a = u diag(s) v^T (include/pcacc.h: pcacc_svd3)."""
    n = a.shape[0]
    u, v = torch.empty_like(a), torch.empty_like(a)
    s = torch.empty((n, 3), dtype=torch.float32, device=a.device)
    _check(lib().pcacc_svd3(_dev(a, torch.float32, 'a'), _i64(n), _dev(u), _dev(s), _dev(v), _stream()), 'svd3')
    return u, s, v


def svd3_backward(u, s, v, gu, gs, gv):
    ga = torch.empty_like(u)
    _check(lib().pcacc_svd3_backward(_dev(u, torch.float32), _dev(s, torch.float32), _dev(v, torch.float32), _opt(gu, torch.float32, 'grad_u'),
                                     _opt(gs, torch.float32, 'grad_s'), _opt(gv, torch.float32, 'grad_v'), _i64(u.shape[0]), _dev(ga), _stream()),
           'svd3_backward')
    return ga


def bn_rows_supported(x):
    c = x.shape[1] if x.dim() == 2 else 0
    v = 8 if x.dtype == torch.bfloat16 else 4
    return x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[0] >= 1 and 0 < c <= 256 and c % v == 0 and 256 % (c // v) == 0


def _bn_ws(rows, c, dev):
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_bn_rows_workspace_bytes(_i64(rows), int(c), ctypes.byref(need)), 'bn_rows_workspace')
    return _ws(need.value, dev)


def bn_rows_forward(x, gamma, beta, eps, momentum, running_mean, running_var, relu=False):
    """Training-mode BatchNorm1d over the rows of x [rows,c] (include/pcacc.h): -> (y like x, save_mean [c], save_invstd [c]);
    running_mean / running_var (f32, or None) are updated in place.  relu: y = max(bn(x), 0) in the same pass."""
    rows, c = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _bn_ws(rows, c, x.device)
    fn = lib().pcacc_bn_relu_rows_forward if relu else lib().pcacc_bn_rows_forward
    _check(fn(_dev(x, None, 'x'), _dtype_code(x), _i64(rows), int(c), _opt(gamma, torch.float32, 'gamma'),
                                       _opt(beta, torch.float32, 'beta'), ctypes.c_float(eps), ctypes.c_float(momentum),
                                       _opt(running_mean, torch.float32, 'running_mean'), _opt(running_var, torch.float32, 'running_var'),
                                       _dev(y), _dev(mean), _dev(invstd), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'bn_rows_forward')
    return y, mean, invstd


def bn_rows_forward_dual(x, gamma, beta, eps, momentum, running_mean, running_var, relu=False):
    """bn_rows_forward on f32 rows -> (y f32, y as bf16, absmax256 array of y, save_mean, save_invstd): the 'mixed' mode's shadow and the next fp32x3
    layer's scale from the same store phase."""
    rows, c = x.shape
    y = torch.empty_like(x)
    y16 = torch.empty((rows, c), dtype=torch.bfloat16, device=x.device)
    am = _zero256(x.device)
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _bn_ws(rows, c, x.device)
    _check(lib().pcacc_bn_rows_forward_dual(_dev(x, torch.float32, 'x'), _i64(rows), int(c), _opt(gamma, torch.float32, 'gamma'), _opt(beta, torch.float32, 'beta'),
                                            ctypes.c_float(eps), ctypes.c_float(momentum), _opt(running_mean, torch.float32, 'running_mean'),
                                            _opt(running_var, torch.float32, 'running_var'), 1 if relu else 0, _dev(y), _dev(y16), _dev(am), _dev(mean),
                                            _dev(invstd), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'bn_rows_forward_dual')
    return y, y16, am, mean, invstd


def bn_rows_backward(grad_y, x, gamma, save_mean, save_invstd, relu_beta=None, relu=False, want_amax=False):
    """-> (grad_x like x, grad_gamma [c] f32, grad_beta [c] f32[, absmax256 array of grad_x]).  relu: the backward of bn_rows_forward(relu=True)
    (relu_beta = its beta)."""
    rows, c = x.shape
    gx = torch.empty_like(x)
    gg = torch.empty((c,), dtype=torch.float32, device=x.device)
    gb = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _bn_ws(rows, c, x.device)
    if grad_y.dtype != x.dtype:
        raise NativeError('bn_rows_backward: grad_y must have the type of x')
    if want_amax:
        am = _zero256(x.device)
        _check(lib().pcacc_bn_rows_backward_m(_dev(grad_y, None, 'grad_y'), _dev(x, None, 'x'), _dtype_code(x), _i64(rows), int(c),
                                              _opt(gamma, torch.float32, 'gamma'), _opt(relu_beta, torch.float32, 'beta') if relu else None, 1 if relu else 0,
                                              _dev(save_mean, torch.float32), _dev(save_invstd, torch.float32), _dev(gx), _dev(am), _dev(gg), _dev(gb),
                                              _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'bn_rows_backward_m')
        return gx, gg, gb, am
    if relu:
        _check(lib().pcacc_bn_relu_rows_backward(_dev(grad_y, None, 'grad_y'), _dev(x, None, 'x'), _dtype_code(x), _i64(rows), int(c),
                                                 _opt(gamma, torch.float32, 'gamma'), _opt(relu_beta, torch.float32, 'beta'),
                                                 _dev(save_mean, torch.float32), _dev(save_invstd, torch.float32), _dev(gx), _dev(gg), _dev(gb),
                                                 _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'bn_relu_rows_backward')
        return gx, gg, gb
    _check(lib().pcacc_bn_rows_backward(_dev(grad_y, None, 'grad_y'), _dev(x, None, 'x'), _dtype_code(x), _i64(rows), int(c),
                                        _opt(gamma, torch.float32, 'gamma'), _dev(save_mean, torch.float32), _dev(save_invstd, torch.float32),
                                        _dev(gx), _dev(gg), _dev(gb), _dev(ws), ctypes.c_size_t(ws.numel()), _stream()), 'bn_rows_backward')
    return gx, gg, gb


# ---- TubeNet slot algebra (include/pcacc.h: pcacc_tube_*) ---------------------------------------------------------------------------
def tube_rows(xyz, slot, slot_centre, n_frames):
    """rows [n,4] f32 = (xyz - centre of the instance's anchor frame, t / n_frames)."""
    n = xyz.shape[0]
    rows = torch.empty((n, 4), dtype=torch.float32, device=xyz.device)
    _check(lib().pcacc_tube_rows(_dev(xyz, torch.float32, 'xyz'), _dev(slot, torch.int32, 'slot'), _dev(slot_centre, torch.float32, 'slot_centre'),
                                 _i64(n), int(n_frames), _dev(rows), _stream()), 'tube_rows')
    return rows


def tube_code(geo, motion, frame, n_frames):
    n_inst, c = geo.shape
    code = torch.empty((n_inst * n_frames, 4 * c), dtype=torch.float32, device=geo.device)
    _check(lib().pcacc_tube_code(_dev(geo, torch.float32, 'geo'), _dev(motion, torch.float32, 'motion'), _dev(frame, torch.float32, 'frame'),
                                 _i64(n_inst), int(n_frames), int(c), _dev(code), _stream()), 'tube_code')
    return code


def tube_code_backward(grad_code, n_inst, n_frames, c):
    dev = grad_code.device
    g_geo = torch.empty((n_inst, c), dtype=torch.float32, device=dev)
    g_motion = torch.empty((n_inst, c), dtype=torch.float32, device=dev)
    g_frame = torch.empty((n_inst * n_frames, c), dtype=torch.float32, device=dev)
    _check(lib().pcacc_tube_code_backward(_dev(grad_code, torch.float32, 'grad_code'), _i64(n_inst), int(n_frames), int(c), _dev(g_geo),
                                          _dev(g_motion), _dev(g_frame), _stream()), 'tube_code_backward')
    return g_geo, g_motion, g_frame


def tube_pose_forward(pose_vec, remaining, total, slot_centre, weights, n_frames):
    """-> pose_c [S,12], gt_c [S,12], step [S,4,4], remaining_out [S,4,4], total_out [S,4,4], loss_rt [2] f64, wsum [1]."""
    s = pose_vec.shape[0]
    dev = pose_vec.device
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    pose_c, gt_c, step, rem_out, total_out, wsum = f(s, 12), f(s, 12), f(s, 4, 4), f(s, 4, 4), f(s, 4, 4), f(1)
    loss_rt = torch.empty((2,), dtype=torch.float64, device=dev)
    _check(lib().pcacc_tube_pose_forward(_dev(pose_vec, torch.float32, 'pose_vec'), _dev(remaining, torch.float32, 'remaining'),
                                         _opt(total, torch.float32, 'total'), _dev(slot_centre, torch.float32, 'slot_centre'),
                                         _dev(weights, torch.float32, 'weights'), int(s), int(n_frames), _dev(pose_c), _dev(gt_c), _dev(step),
                                         _dev(rem_out), _dev(total_out), _dev(loss_rt), _dev(wsum), _stream()), 'tube_pose_forward')
    return pose_c, gt_c, step, rem_out, total_out, loss_rt, wsum


def tube_gap_forward(rows, slot, pose_c, gt_c):
    n = rows.shape[0]
    pp = torch.empty((n, 4), dtype=torch.float32, device=rows.device)
    _check(lib().pcacc_tube_gap_forward(_dev(rows, torch.float32, 'rows'), _dev(slot, torch.int32, 'slot'), _dev(pose_c, torch.float32, 'pose_c'),
                                        _dev(gt_c, torch.float32, 'gt_c'), _i64(n), _dev(pp), _stream()), 'tube_gap_forward')
    return pp


def tube_finish(slot_sums, count, weights, wsum):
    l12 = torch.empty((2,), dtype=torch.float32, device=slot_sums.device)
    _check(lib().pcacc_tube_finish(_dev(slot_sums, torch.float32, 'slot_sums'), int(slot_sums.shape[1]), _dev(count, torch.float32, 'count'),
                                   _dev(weights, torch.float32, 'weights'), _dev(wsum, torch.float32, 'wsum'), int(slot_sums.shape[0]),
                                   _dev(l12), _stream()), 'tube_finish')
    return l12


def tube_gap_backward(rows, slot, pose_c, gt_c, weights, count, wsum, grad_l1, grad_l2):
    n = rows.shape[0]
    g16 = torch.empty((n, 16), dtype=torch.float32, device=rows.device)
    _check(lib().pcacc_tube_gap_backward(_dev(rows, torch.float32, 'rows'), _dev(slot, torch.int32, 'slot'), _dev(pose_c, torch.float32, 'pose_c'),
                                         _dev(gt_c, torch.float32, 'gt_c'), _dev(weights, torch.float32, 'weights'),
                                         _dev(count, torch.float32, 'count'), _dev(wsum, torch.float32, 'wsum'),
                                         _opt(grad_l1, torch.float32, 'grad_l1'), _opt(grad_l2, torch.float32, 'grad_l2'), _i64(n), _dev(g16),
                                         _stream()), 'tube_gap_backward')
    return g16


def tube_pose_backward(pose_vec, remaining, slot_centre, weights, wsum, grad_pose, grad_rot, grad_trans, n_frames):
    s = pose_vec.shape[0]
    g_vec = torch.empty((s, 7), dtype=torch.float32, device=pose_vec.device)
    _check(lib().pcacc_tube_pose_backward(_dev(pose_vec, torch.float32, 'pose_vec'), _dev(remaining, torch.float32, 'remaining'),
                                          _dev(slot_centre, torch.float32, 'slot_centre'), _dev(weights, torch.float32, 'weights'),
                                          _dev(wsum, torch.float32, 'wsum'), _dev(grad_pose, torch.float32, 'grad_pose'), int(grad_pose.shape[1]),
                                          _opt(grad_rot, torch.float64, 'grad_rot'), _opt(grad_trans, torch.float64, 'grad_trans'), int(s),
                                          int(n_frames), _dev(g_vec), _stream()), 'tube_pose_backward')
    return g_vec


# ---- tail of a U-Net encoder stage (include/pcacc.h: pcacc_maxpool2x2_bf16, pcacc_pool_skip_relu_backward_bf16) -----------------------
def maxpool2x2(x_rows):
    """[n_img, h, w, c] bf16 / f32 channels-last -> [n_img, h/2, w/2, c]."""
    n, h, w, c = x_rows.shape
    out = torch.empty((n, h // 2, w // 2, c), dtype=x_rows.dtype, device=x_rows.device)
    if x_rows.dtype == torch.float32:
        _check(lib().pcacc_maxpool2x2_f32(_dev(x_rows, torch.float32, 'x'), _i64(n), int(h), int(w), int(c), _dev(out), _stream()), 'maxpool2x2')
    else:
        _check(lib().pcacc_maxpool2x2_bf16(_dev(x_rows, torch.bfloat16, 'x'), _i64(n), int(h), int(w), int(c), _dev(out), _stream()), 'maxpool2x2')
    return out


def _pixel_pitch(g, shape):
    """Elements between consecutive pixels of a [n,h,w,c] map whose channels are contiguous and whose pixels are evenly spaced (a dense map,
    or a channel slice of a wider dense map); None when the layout is anything else."""
    n, h, w, c = shape
    if tuple(g.shape) != tuple(shape) or g.stride(3) != 1:
        return None
    p = g.stride(2)
    if p < c or g.stride(1) != w * p or g.stride(0) != h * w * p or g.data_ptr() % 16:
        return None
    return p


def pool_skip_relu_backward(y_rows, grad_pooled, grad_skip, want_amax=False):
    """(un-pool(grad_pooled) + grad_skip) * (y > 0) in one pass; either gradient may be None.  bf16 or f32 rows (all of y's type);
    grad_skip may be a channel slice of a wider map (read in place); want_amax (f32): -> (grad, absmax256 array of it)."""
    n, h, w, c = y_rows.shape
    if y_rows.dtype == torch.float32 and any(g is not None and g.dtype == torch.bfloat16 for g in (grad_pooled, grad_skip)):
        # 'mixed' mode: the forward's own fp32 y decides the windows' winners, the gradients are bf16
        out = torch.empty(y_rows.shape, dtype=torch.bfloat16, device=y_rows.device)
        pitch, gs_ptr = c, None
        if grad_skip is not None:
            pitch = _pixel_pitch(grad_skip, y_rows.shape)
            if pitch is None or pitch % 8:
                grad_skip, pitch = grad_skip.contiguous(), c
            gs_ptr = ctypes.c_void_p(grad_skip.data_ptr())
        _check(lib().pcacc_pool_skip_relu_backward_strided_y32(_dev(y_rows, torch.float32, 'y'), _opt(grad_pooled, torch.bfloat16, 'grad_pooled'), gs_ptr,
                                                               _i64(pitch), _i64(n), int(h), int(w), int(c), _dev(out), _stream()),
               'pool_skip_relu_backward')
        return out
    out = torch.empty_like(y_rows)
    f32 = y_rows.dtype == torch.float32
    dt = torch.float32 if f32 else torch.bfloat16
    pitch, gs_ptr = c, None
    if grad_skip is not None:
        if grad_skip.dtype != dt or not grad_skip.is_cuda:
            raise NativeError('pool_skip_relu_backward: grad_skip must be a %s GPU tensor' % dt)
        pitch = _pixel_pitch(grad_skip, y_rows.shape)
        if pitch is None or pitch % (4 if f32 else 8):
            grad_skip, pitch = grad_skip.contiguous(), c
        gs_ptr = ctypes.c_void_p(grad_skip.data_ptr())
    if f32:
        amax = _zero256(y_rows.device) if want_amax else None
        _check(lib().pcacc_pool_skip_relu_backward_strided_f32(_dev(y_rows, torch.float32, 'y'), _opt(grad_pooled, torch.float32, 'grad_pooled'), gs_ptr,
                                                               _i64(pitch), _i64(n), int(h), int(w), int(c), _dev(out),
                                                               _dev(amax) if want_amax else None, _stream()), 'pool_skip_relu_backward')
        return (out, amax) if want_amax else out
    _check(lib().pcacc_pool_skip_relu_backward_strided_bf16(_dev(y_rows, torch.bfloat16, 'y'), _opt(grad_pooled, torch.bfloat16, 'grad_pooled'), gs_ptr,
                                                            _i64(pitch), _i64(n), int(h), int(w), int(c), _dev(out), _stream()),
           'pool_skip_relu_backward')
    return out


# ---- fused ResnetBlockFC of the pillar encoder (include/pcacc.h: pcacc_pfn_block_*) ---------------------------------------------------
PFN_BLOCK_SLICES = {'w1': (0, 1024, (32, 32)), 'b1': (1024, 1056, (32,)), 'ws': (1056, 3104, (32, 64)), 'w0': (3104, 5152, (32, 64)),
                    'b0': (5152, 5184, (32,))}


def pfn_block_forward(xa, pooled, p2v, w0, b0, ws, w1, b1):
    """-> (out [rows,32] bf16, relu(h) [rows,32] bf16)."""
    rows = xa.shape[0]
    out = torch.empty((rows, 32), dtype=torch.bfloat16, device=xa.device)
    hr = torch.empty((rows, 32), dtype=torch.bfloat16, device=xa.device)
    _check(lib().pcacc_pfn_block_forward(_dev(xa, torch.bfloat16, 'xa'), _opt(pooled, torch.bfloat16, 'pooled'), _opt(p2v, torch.int32, 'p2v'),
                                         _dev(w0, torch.float32, 'w0'), _opt(b0, torch.float32, 'b0'), _dev(ws, torch.float32, 'ws'),
                                         _dev(w1, torch.float32, 'w1'), _opt(b1, torch.float32, 'b1'), _dev(out), _dev(hr), _i64(rows),
                                         _stream()), 'pfn_block_forward')
    return out, hr


def pfn_block_backward(xa, pooled, p2v, hr, grad_out, w0, ws, w1):
    """-> (grad_xa, grad_xb rows or None, grad_params [5184] f32; PFN_BLOCK_SLICES names its parts)."""
    rows = xa.shape[0]
    dev = xa.device
    two = pooled is not None
    gxa = torch.empty((rows, 32 if two else 64), dtype=torch.bfloat16, device=dev)
    gxb = torch.empty((rows, 32), dtype=torch.bfloat16, device=dev) if two else None
    gp = torch.empty((5184,), dtype=torch.float32, device=dev)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_pfn_block_backward_workspace_bytes(_i64(rows), ctypes.byref(need)), 'pfn_block_backward_workspace')
    ws_buf = _ws(need.value, dev)
    _check(lib().pcacc_pfn_block_backward(_dev(xa, torch.bfloat16, 'xa'), _opt(pooled, torch.bfloat16, 'pooled'), _opt(p2v, torch.int32, 'p2v'),
                                          _dev(hr, torch.bfloat16, 'relu_h'), _dev(grad_out, torch.bfloat16, 'grad_out'),
                                          _dev(w0, torch.float32, 'w0'), _dev(ws, torch.float32, 'ws'), _dev(w1, torch.float32, 'w1'),
                                          _dev(gxa), _opt(gxb, torch.bfloat16, 'grad_xb'), _dev(gp), _i64(rows), _dev(ws_buf),
                                          ctypes.c_size_t(ws_buf.numel()), _stream()), 'pfn_block_backward')
    return gxa, gxb, gp


def pfn_block_split_forward(xa, xa_amax, pooled, pooled_amax, p2v, w0, b0, ws, w1, b1):
    """The block on f32 rows (fp32x3) -> (out, relu(h) [rows,32] f32, xmask [rows] i64, hmask [rows] i32, out_amax, hr_amax)."""
    rows = xa.shape[0]
    dev = xa.device
    out = torch.empty((rows, 32), dtype=torch.float32, device=dev)
    hr = torch.empty((rows, 32), dtype=torch.float32, device=dev)
    xmask = torch.empty((rows,), dtype=torch.int64, device=dev)
    hmask = torch.empty((rows,), dtype=torch.int32, device=dev)
    out_amax, hr_amax = _zero256(dev), _zero256(dev)
    _check(lib().pcacc_pfn_block_split_forward(_dev(xa, torch.float32, 'xa'), _dev(xa_amax, torch.float32, 'xa_amax'), _opt(pooled, torch.float32, 'pooled'),
                                               _opt(pooled_amax, torch.float32, 'pooled_amax'), _opt(p2v, torch.int32, 'p2v'),
                                               _dev(w0, torch.float32, 'w0'), _opt(b0, torch.float32, 'b0'), _dev(ws, torch.float32, 'ws'),
                                               _dev(w1, torch.float32, 'w1'), _opt(b1, torch.float32, 'b1'), _dev(out), _dev(hr), _dev(xmask),
                                               _dev(hmask), _dev(out_amax), _dev(hr_amax), _i64(rows), _stream()), 'pfn_block_split_forward')
    return out, hr, xmask, hmask, out_amax, hr_amax


def pfn_block_split_forward_dual(xa, xa_amax, pooled, pooled_amax, p2v, w0, b0, ws, w1, b1):
    """'mixed' mode: -> (out [rows,32] f32, its absmax256 array, out as bf16, relu(h) [rows,32] bf16) from one kernel."""
    rows, dev = xa.shape[0], xa.device
    out = torch.empty((rows, 32), dtype=torch.float32, device=dev)
    out16 = torch.empty((rows, 32), dtype=torch.bfloat16, device=dev)
    hr16 = torch.empty((rows, 32), dtype=torch.bfloat16, device=dev)
    out_amax = _zero256(dev)
    _check(lib().pcacc_pfn_block_split_forward_dual(_dev(xa, torch.float32, 'xa'), _dev(xa_amax, torch.float32, 'xa_amax'), _opt(pooled, torch.float32, 'pooled'),
                                                    _opt(pooled_amax, torch.float32, 'pooled_amax'), _opt(p2v, torch.int32, 'p2v'),
                                                    _dev(w0, torch.float32, 'w0'), _opt(b0, torch.float32, 'b0'), _dev(ws, torch.float32, 'ws'),
                                                    _dev(w1, torch.float32, 'w1'), _opt(b1, torch.float32, 'b1'), _dev(out), _dev(out16), _dev(hr16),
                                                    _dev(out_amax), _i64(rows), _stream()), 'pfn_block_split_forward_dual')
    return out, out_amax, out16, hr16


def pfn_block_split_dgrad(grad_out, grad_out_amax, xmask, hmask, w0, ws, w1, two_pieces):
    """-> (grad_xa, grad_xb rows or None, d(h) [rows,32], amax of the d(x) pieces together, amax of d(h))."""
    rows = grad_out.shape[0]
    dev = grad_out.device
    gxa = torch.empty((rows, 32 if two_pieces else 64), dtype=torch.float32, device=dev)
    gxb = torch.empty((rows, 32), dtype=torch.float32, device=dev) if two_pieces else None
    dh = torch.empty((rows, 32), dtype=torch.float32, device=dev)
    gx_amax, dh_amax = _zero256(dev), _zero256(dev)
    _check(lib().pcacc_pfn_block_split_dgrad(_dev(grad_out, torch.float32, 'grad_out'), _dev(grad_out_amax, torch.float32, 'grad_out_amax'),
                                             _dev(xmask, torch.int64, 'xmask'), _dev(hmask, torch.int32, 'hmask'), _dev(w0, torch.float32, 'w0'),
                                             _dev(ws, torch.float32, 'ws'), _dev(w1, torch.float32, 'w1'), _dev(gxa),
                                             _opt(gxb, torch.float32, 'grad_xb'), _dev(dh), _dev(gx_amax), _dev(dh_amax), _i64(rows), _stream()),
           'pfn_block_split_dgrad')
    return gxa, gxb, dh, gx_amax, dh_amax


# ---- matching stage of the ego head under autograd (include/pcacc.h: pcacc_ego_affinity_* / pcacc_ego_perm_*) -------------------------
def ego_affinity_forward(feats_s, feats_t, params):
    """feats [P,k,c] f32 (L2-normalised rows), params [2] f32 = (softplus(alpha), exp(beta) + 0.02) -> affinity [P,k,k]."""
    p, k, c = feats_s.shape
    out = torch.empty((p, k, k), dtype=torch.float32, device=feats_s.device)
    _check(lib().pcacc_ego_affinity_forward(_dev(feats_s, torch.float32, 'feats_s'), _dev(feats_t, torch.float32, 'feats_t'),
                                            _dev(params, torch.float32, 'params'), int(p), int(k), int(c), _dev(out), _stream()), 'ego_affinity_forward')
    return out


def ego_affinity_backward(grad_aff, aff, params):
    """-> (grad_dot [P,k,k], grad_params [2])."""
    gd = torch.empty_like(aff)
    gp = torch.empty((2,), dtype=torch.float32, device=aff.device)
    need = ctypes.c_size_t(0)
    _check(lib().pcacc_ego_affinity_backward_workspace_bytes(ctypes.byref(need)), 'ego_affinity_backward_workspace')
    ws = _ws(need.value, aff.device)
    _check(lib().pcacc_ego_affinity_backward(_dev(grad_aff, torch.float32, 'grad_affinity'), _dev(aff, torch.float32, 'affinity'),
                                             _dev(params, torch.float32, 'params'), _i64(aff.numel()), _dev(gd), _dev(gp), _dev(ws),
                                             ctypes.c_size_t(ws.numel()), _stream()), 'ego_affinity_backward')
    return gd, gp


def ego_perm_forward(log_perm, coor_s, coor_t, thr2):
    """-> (perm [P,k,k], rowsum [P,k], weighted_t [P,k,3], colsum [P,k])."""
    p, k, _ = log_perm.shape
    dev = log_perm.device
    perm = torch.empty_like(log_perm)
    rowsum = torch.empty((p, k), dtype=torch.float32, device=dev)
    colsum = torch.empty((p, k), dtype=torch.float32, device=dev)
    wt = torch.empty((p, k, 3), dtype=torch.float32, device=dev)
    _check(lib().pcacc_ego_perm_forward(_dev(log_perm, torch.float32, 'log_perm'), _dev(coor_s, torch.float32, 'coor_s'),
                                        _dev(coor_t, torch.float32, 'coor_t'), _dev(thr2, torch.float32, 'thr2'), int(p), int(k), _dev(perm),
                                        _dev(rowsum), _dev(wt), _dev(colsum), _stream()), 'ego_perm_forward')
    return perm, rowsum, wt, colsum


def ego_perm_backward(g_perm, g_rowsum, g_wt, g_colsum, perm, coor_t, rowsum, wt):
    p, k, _ = perm.shape
    out = torch.empty_like(perm)
    _check(lib().pcacc_ego_perm_backward(_opt(g_perm, torch.float32, 'grad_perm'), _opt(g_rowsum, torch.float32, 'grad_rowsum'),
                                         _opt(g_wt, torch.float32, 'grad_weighted_t'), _opt(g_colsum, torch.float32, 'grad_colsum'),
                                         _dev(perm, torch.float32, 'perm'), _dev(coor_t, torch.float32, 'coor_t'),
                                         _dev(rowsum, torch.float32, 'rowsum'), _dev(wt, torch.float32, 'weighted_t'), int(p), int(k), _dev(out),
                                         _stream()), 'ego_perm_backward')
    return out


def kabsch_cov_forward(x1, x2, w):
    """x1, x2 [P,k,3], w [P,k] f32 -> (cov [P,3,3], m1 [P,3], m2 [P,3], norm [P,2])."""
    p, k, _ = x1.shape
    dev = x1.device
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    cov, m1, m2, norm = f(p, 3, 3), f(p, 3), f(p, 3), f(p, 2)
    _check(lib().pcacc_kabsch_cov_forward(_dev(x1, torch.float32, 'x1'), _dev(x2, torch.float32, 'x2'), _dev(w, torch.float32, 'w'), int(p), int(k),
                                          _dev(cov), _dev(m1), _dev(m2), _dev(norm), _stream()), 'kabsch_cov_forward')
    return cov, m1, m2, norm


def kabsch_cov_backward(x1, x2, w, m1, m2, norm, g_cov, g_m1, g_m2):
    p, k, _ = x1.shape
    gx2 = torch.empty_like(x2)
    gw = torch.empty_like(w)
    _check(lib().pcacc_kabsch_cov_backward(_dev(x1, torch.float32, 'x1'), _dev(x2, torch.float32, 'x2'), _dev(w, torch.float32, 'w'),
                                           _dev(m1, torch.float32, 'm1'), _dev(m2, torch.float32, 'm2'), _dev(norm, torch.float32, 'norm'),
                                           _opt(g_cov, torch.float32, 'grad_cov'), _opt(g_m1, torch.float32, 'grad_m1'),
                                           _opt(g_m2, torch.float32, 'grad_m2'), int(p), int(k), _dev(gx2), _dev(gw), _stream()), 'kabsch_cov_backward')
    return gx2, gw


def kabsch_rt_forward(u, v, m1, m2):
    """u, v [n,3,3], m1, m2 [n,3] f32 -> (rot [n,3,3], trans [n,3])."""
    n = u.shape[0]
    rot = torch.empty((n, 3, 3), dtype=torch.float32, device=u.device)
    trans = torch.empty((n, 3), dtype=torch.float32, device=u.device)
    _check(lib().pcacc_kabsch_rt_forward(_dev(u, torch.float32, 'u'), _dev(v, torch.float32, 'v'), _dev(m1, torch.float32, 'm1'),
                                         _dev(m2, torch.float32, 'm2'), int(n), _dev(rot), _dev(trans), _stream()), 'kabsch_rt_forward')
    return rot, trans


def kabsch_rt_backward(u, v, m1, g_rot, g_trans):
    n = u.shape[0]
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=u.device)
    gu, gv, gm1, gm2 = f(n, 3, 3), f(n, 3, 3), f(n, 3), f(n, 3)
    _check(lib().pcacc_kabsch_rt_backward(_dev(u, torch.float32, 'u'), _dev(v, torch.float32, 'v'), _dev(m1, torch.float32, 'm1'),
                                          _opt(g_rot, torch.float32, 'grad_rot'), _opt(g_trans, torch.float32, 'grad_trans'), int(n), _dev(gu),
                                          _dev(gv), _dev(gm1), _dev(gm2), _stream()), 'kabsch_rt_backward')
    return gu, gv, gm1, gm2


def inv4x4(m):
    """[..., 4, 4] f32 -> inverses, one launch (include/pcacc.h: pcacc_inv4x4)."""
    x = m.contiguous().float()
    out = torch.empty_like(x)
    _check(lib().pcacc_inv4x4(_dev(x, torch.float32, 'm'), _i64(x.numel() // 16), _dev(out), _stream()), 'inv4x4')
    return out
