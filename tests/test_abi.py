"""CPU-side checks of the C-ABI boundary: the library loads (no GPU needed),
exports every symbol include/empanada_hip.h declares, and the Python binding
covers exactly that set."""
import os
import re

import pytest


def _declared():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'include', 'empanada_hip.h')).read()
    return sorted(set(re.findall(r'EMP_API[^;]*?\b(emp_\w+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from empanada_napari_amd import _abi
    lib = _abi.load(build_if_missing=True)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in the header but not exported'
    assert sorted(_abi.PROTOTYPES) == names, 'ctypes prototypes out of sync with the header'
    assert lib.emp_abi_version() == 5


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from empanada_napari_amd import engines, weights
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        engines.HipPanopticDeepLab({}, folded=True)


def test_product_package_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, 'empanada-napari_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')) and f != '_smoke.py':
                txt = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
