import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the in-tree library is git-ignored: build it if a fresh checkout has not run build() yet
    so = os.path.join(REPO, 'ballermixplus_amd', 'libbmxscan.so')
    if not os.path.exists(so):
        import subprocess
        subprocess.run(['make', '-C', os.path.join(REPO, 'ballermixplus_amd', 'csrc')], check=False,
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
