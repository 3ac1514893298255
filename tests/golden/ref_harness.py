"""Import the Python reference from /root/reference (this container only) to emit golden vectors.

NOT used by any test at run time: tests read the committed .npz fixtures.  Only
tests/golden/make_golden.py imports this file.  The reference needs modules that are not
installed here; each stand-in below is the published semantics of the call the hot path makes
(SURVEY.md 8c lists why each is faithful):

  torch_scatter.scatter(src, index, dim=0, dim_size=None, reduce=...)  -> scatter_reduce,
      include_self=False, empty segments stay 0 (torch_scatter's documented fill)
  numba.jit                                                              -> identity decorator
  open3d, nestargs, tensorboardX, IPython                                -> empty modules
  torchsparse.utils.quantize.sparse_quantize (v1.4.0, README.md:27)      -> floor + ravel hash + np.unique, the
      algorithm the reference also keeps at dataset_toolbox/prep_nuscene_waymo_sf/libs/spv_utils.py:7-75
  torch.utils.cpp_extension.load (Chamfer JIT at import)                 -> empty namespace
"""
import sys
import types

import torch

REF_ROOT = '/root/reference'


def _scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    assert dim == 0 and out is None
    index = index.long()
    n = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    red = {'sum': 'sum', 'add': 'sum', 'mean': 'mean', 'max': 'amax', 'min': 'amin'}[reduce]
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    return res.scatter_reduce(0, idx, src, red, include_self=False)


def _sparse_quantize(coords, voxel_size=1, *, return_index=False, return_inverse=False):
    import numpy as np
    c = np.floor(coords / voxel_size).astype(np.int32)
    k = c - c.min(0)
    k = k.astype(np.uint64)
    kmax = k.max(0).astype(np.uint64) + 1
    h = np.zeros(k.shape[0], dtype=np.uint64)
    for j in range(k.shape[1] - 1):
        h += k[:, j]
        h *= kmax[j + 1]
    h += k[:, -1]
    _, idx, inv = np.unique(h, return_index=True, return_inverse=True)
    out = [c[idx]]
    if return_index:
        out.append(idx)
    if return_inverse:
        out.append(inv)
    return out


def install():
    if REF_ROOT in sys.path:
        return
    ts = types.ModuleType('torch_scatter')
    ts.scatter = _scatter
    sys.modules['torch_scatter'] = ts

    nb = types.ModuleType('numba')

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.jit = nb.njit = jit
    sys.modules['numba'] = nb

    for name in ['open3d', 'nestargs', 'tensorboardX', 'IPython', 'IPython.display', 'torchsparse',
                 'torchsparse.utils', 'torchsparse.utils.quantize']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['IPython.display'].display = print

    sys.modules['torchsparse.utils.quantize'].sparse_quantize = _sparse_quantize
    sys.modules['tensorboardX'].SummaryWriter = object

    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: types.SimpleNamespace()
    sys.path.insert(0, REF_ROOT)


def collate(samples):
    install()
    from libs.dataloader import collate_fn
    return collate_fn(samples)


def voxeliser(cfg):
    install()
    from libs.voxel_generator import Voxelization
    return Voxelization(cfg['voxel_generator'])
