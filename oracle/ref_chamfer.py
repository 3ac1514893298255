"""Loader for oracle/_ref/cd_ref*.so: the reference's Chamfer C++ CPU path (chamfer_distance/chamfer_distance.cpp:59-177)
built by `make -C oracle ref` from the reference source in place.  TEST INFRASTRUCTURE: used to pin oracle.chamfer_* and to
emit tests/golden/chamfer.npz; never imported by the product."""
import glob
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return bool(glob.glob(os.path.join(_HERE, '_ref', 'cd_ref*.so')))


def load():
    import torch  # noqa: F401  (libtorch must be loaded first)
    path = glob.glob(os.path.join(_HERE, '_ref', 'cd_ref*.so'))[0]
    flags = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)      # the CUDA launcher symbols are undefined and never called
    try:
        spec = importlib.util.spec_from_file_location('cd_ref', path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(flags)
    return mod


def forward(xyz1, xyz2):
    """torch CPU tensors [b,n,3], [b,m,3] -> dist1, dist2, idx1, idx2 exactly as ChamferDistanceFunction.forward
    (chamfer_distance/chamfer_distance.py:11-23) allocates and fills them."""
    import torch
    cd = load()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1, d2 = torch.zeros(b, n), torch.zeros(b, m)
    i1, i2 = torch.zeros(b, n, dtype=torch.int), torch.zeros(b, m, dtype=torch.int)
    cd.forward(xyz1.contiguous(), xyz2.contiguous(), d1, d2, i1, i2)
    return d1, d2, i1, i2


def backward(xyz1, xyz2, gd1, gd2, i1, i2):
    import torch
    cd = load()
    g1, g2 = torch.zeros(xyz1.size()), torch.zeros(xyz2.size())
    cd.backward(xyz1.contiguous(), xyz2.contiguous(), g1, g2, gd1.contiguous(), gd2.contiguous(), i1, i2)
    return g1, g2
