"""The C-ABI library loads without a GPU and exports exactly the symbols include/pnpi.h declares (no compute calls here)."""
import os
import re

from pnpinversion_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pnpi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(pnpi_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    lib = _capi.load_library()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libpnpi.so does not export %s" % n
    assert names == set(_capi.SYMBOLS), (names ^ set(_capi.SYMBOLS))


def test_sd1_config_roundtrip():
    from pnpinversion_amd.config import SD1
    lib = _capi.load_library()
    c = _capi.ModelConfig()
    lib.pnpi_config_sd1(c)
    mine = SD1.to_c()
    for f, _ in _capi.ModelConfig._fields_:
        a, b = getattr(c, f), getattr(mine, f)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), f
        else:
            assert a == b, f


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(ImportError):
        _capi.load_library(str(tmp_path / "nope.so"))


def test_integration_doc_names_every_symbol():
    """INTEGRATION.md is the binding guide: every entry point of include/pnpi.h has to appear in it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "pnpi.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    syms = sorted(set(re.findall(r"\b(pnpi_[a-z0-9_]+)\s*\(", hdr)))
    assert len(syms) >= 39 and not [s for s in syms if s not in doc]
