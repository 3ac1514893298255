"""CPU-only: the C-ABI library loads and exports every symbol include/pcacc.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'pcacc.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pcacc_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    names = _declared()
    assert 'pcacc_voxelize' in names and 'pcacc_pillar_scatter' in names and 'pcacc_chamfer_forward' in names
    assert len(names) >= 20


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from pcaccumulation_amd import native
    assert os.path.exists(native.LIB_PATH)
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.pcacc_target.restype = ctypes.c_char_p
    assert lib.pcacc_target() == b'gfx950'
    assert sorted(native.EXPORTS + ['pcacc_target']) == _declared()


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing on the host."""
    import torch
    from pcaccumulation_amd import native
    with pytest.raises(native.NativeError):
        native.rigid_transform(torch.zeros(4, 3), torch.zeros(4, dtype=torch.int32), torch.zeros(1, 16))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'pcaccumulation_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(import oracle|from oracle)', src, flags=re.M), f
