import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


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
