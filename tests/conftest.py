import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests exercise the in-tree libpnpi.so.  It normally travels with the snapshot; if it is missing or older than its
    sources (fresh checkout) it is rebuilt here -- hipcc cross-compiles gfx950 without a GPU.  The product itself never builds or
    falls back implicitly: _capi.load_library() raises when the library is absent."""
    try:
        from pnpinversion_amd.build import build
        build(verbose=False)
    except Exception as e:  # keep collecting: the tests that need the library will say what is wrong
        print("[conftest] libpnpi.so build skipped / failed: %s" % e, file=sys.stderr)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
