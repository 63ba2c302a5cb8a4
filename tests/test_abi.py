"""The C-ABI library loads and exports every symbol include/vfn_hip.h declares (no GPU needed)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'vfn_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(vfn_\w+)\s*\(', txt)))


def test_header_and_binding_agree():
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import _lib
    assert declared_symbols() == _lib.ALL_SYMBOLS


def test_library_exports_every_declared_symbol():
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    for s in declared_symbols():
        assert hasattr(L, s), s
    hdr = open(os.path.join(ROOT, 'include', 'vfn_hip.h')).read()
    assert L.vfn_abi_version() == _lib.ABI_VERSION == int(re.search(r'#define VFN_ABI_VERSION (\d+)', hdr).group(1))
    assert L.vfn_conv_cfg_count() == 62


def test_descriptor_sizes_match_the_library():
    """A binding whose ctypes struct drifted from include/vfn_hip.h is caught here (and at load time)."""
    import ctypes as C
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import _lib
    L = _lib.lib()
    for which, cls in _lib.DESC_IDS.items():
        assert L.vfn_sizeof_desc(which) == C.sizeof(cls), cls.__name__
    assert L.vfn_sizeof_desc(99) == -1
    names = __import__('vfloodnet_amd').ops.conv_cfg_names(0)
    assert names[10] == 'conv_igemm_kernel<64, 128, 2, 4, 0>' and names[13] == 'conv_igemm_dma_kernel<64, 64, 2, 2, 2>'


def test_missing_library_fails_loudly(monkeypatch):
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libvfn_hip.so')
    with pytest.raises(RuntimeError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'v-floodnet_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'import oracle' not in src and 'from oracle' not in src, fn
