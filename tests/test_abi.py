"""The C-ABI library loads and exports every symbol include/pca.h declares (no GPU needed, no compute)."""
import ctypes
import os
import re

from conftest import PKG, ROOT


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'pca.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pca_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported():
    so = os.path.join(PKG, 'pca_amd', 'libpca_hip.so')
    assert os.path.exists(so), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(so)
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert lib.pca_version() == 2
    lib.pca_bev_workspace_bytes.restype = ctypes.c_int64
    lib.pca_bev_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
    assert lib.pca_bev_workspace_bytes(1000, 256) > 1000 * 20


def test_binding_lists_every_export():
    from pca_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_functions()
    _lib.load()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M) or 'liboracle' in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pca_amd import _lib
    with pytest.raises(RuntimeError):
        _lib.Context.get()
