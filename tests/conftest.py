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
    import subprocess
    from ballermixplus_amd import _lib
    so = _lib.LIB_PATH
    stale = True
    if os.path.exists(so):
        # asked in a CHILD process: a library opened here stays mapped, and the rebuilt file would never be seen by this
        # process (dlopen returns the handle it already has) -- which once turned a stale binary into "no GPU, all skipped"
        r = subprocess.run([sys.executable, '-c', 'import ctypes,sys; L=ctypes.CDLL(sys.argv[1]); L.bmx_build_id.restype=ctypes.c_char_p; '
                            'print(L.bmx_build_id().decode())', so], capture_output=True, text=True)
        stale = r.returncode != 0 or r.stdout.strip() != _lib.source_id()
    if stale:
        subprocess.run(['make', '-B', '-C', os.path.join(REPO, 'ballermixplus_amd', 'csrc')], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _has_gpu():
    """True when libbmxscan sees a HIP device.  A library that does not load (missing, stale, a declared symbol absent) is a
    build error and stops the run -- it must never turn into "no GPU, tests skipped"; and on a box that has a GPU device node
    (/dev/kfd) but where HIP reports no device, the count is retried (a fresh box may need a moment) and then reported as an error."""
    import time
    from ballermixplus_amd import _lib
    try:
        L = _lib.lib()
    except Exception as e:
        raise pytest.UsageError('libbmxscan.so cannot be used: %r' % (e,))
    import subprocess

    def probe():
        # in a child process: a HIP runtime whose first initialisation failed stays failed for the life of its process
        r = subprocess.run([sys.executable, '-c', 'import ctypes,sys; L=ctypes.CDLL(sys.argv[1]); print(L.bmx_device_count())', _lib.LIB_PATH],
                           capture_output=True, text=True, timeout=120)
        try:
            return int(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            return 0

    if not os.path.exists('/dev/kfd'):
        return L.bmx_device_count() > 0          # no GPU device node: a CPU box
    for attempt in range(10):
        if probe() > 0:
            return L.bmx_device_count() > 0
        time.sleep(3)
    raise pytest.UsageError('/dev/kfd exists but HIP reports no device after 10 attempts: refusing to skip the GPU tests silently')


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass without a device: they are skipped (not passed) on CPU boxes
    if _has_gpu():
        return
    sys.stderr.write('[conftest] no HIP device on this box: GPU tests are skipped\n')
    skip = pytest.mark.skip(reason='no HIP device in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
