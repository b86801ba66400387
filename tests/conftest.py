import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The in-tree library is git-ignored and travels to the GPU box as a binary: (re)build it whenever it is missing
    # or was built from other sources than the ones in the tree (bmx_build_id vs the sources' hash), so that a stale
    # binary is never what gets tested.  `make` is incremental; _lib.lib() raises if the ids still differ.
    import ctypes
    import subprocess
    from ballermixplus_amd import _lib
    so = _lib.LIB_PATH
    stale = True
    if os.path.exists(so):
        try:
            L = ctypes.CDLL(so)
            L.bmx_build_id.restype = ctypes.c_char_p
            stale = L.bmx_build_id().decode() != _lib.source_id()
        except (OSError, AttributeError):
            stale = True
    if stale:
        subprocess.run(['make', '-B', '-C', os.path.join(REPO, 'ballermixplus_amd', 'csrc')], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _has_gpu():
    try:
        from ballermixplus_amd import _lib
        return _lib.lib().bmx_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass without a device: they are skipped (not passed) on CPU boxes
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no HIP device in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
