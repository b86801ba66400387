"""CPU checks of the native pieces: the C-ABI library loads and exports every symbol that
include/bmxscan.h declares (no compute without a GPU), fails loudly without a device, and the
device math header -- compiled here for the host -- is bit-faithful to scipy's Cephes."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
from scipy.special import betaln

from util import REPO


def _declared_symbols():
    src = open(os.path.join(REPO, 'include', 'bmxscan.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(bmx_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from ballermixplus_amd import _lib
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
        assert n in _lib.PROTOTYPES, 'ctypes prototype missing for ' + n
    major, minor = C.c_int(), C.c_int()
    L.bmx_version(C.byref(major), C.byref(minor))
    assert (major.value, minor.value) == (1, 5)
    assert L.bmx_build_id().decode() == _lib.source_id()      # the binary under test was built from this tree
    assert L.bmx_alpha_cut() == 18.420680743952364


def test_no_cpu_fallback_without_a_device():
    from ballermixplus_amd import _lib, engine
    if _lib.lib().bmx_device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_lib.BmxError) as e:
        engine.Context(0)
    assert e.value.code == -2 and 'no CPU fallback' in str(e.value)
    h = C.c_void_p()
    assert _lib.lib().bmx_ctx_create(C.byref(h), 0) == -2 and not h.value
    assert _lib.lib().bmx_ctx_scan(None) == -1          # NULL context is rejected, not dereferenced


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'ballermixplus_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(root, f)).read()
                assert 'oracle' not in txt.replace('no oracle', ''), os.path.join(root, f)


@pytest.fixture(scope='module')
def hostmath(tmp_path_factory):
    d = tmp_path_factory.mktemp('hm')
    src = d / 'h.cpp'
    src.write_text('#include "bmx_math.h"\nextern "C" {\n'
                   'double h_crlog(double x){ return bmx::crlog(x);}\n'
                   'double h_lbeta(double a,double b){ return bmx::cephes::lbeta_pos(a,b);}\n'
                   'double h_pmf(int k,int n,double a,double b){ return bmx::betabinom_pmf(k,n,a,b);}\n}\n')
    so = d / 'libh.so'
    subprocess.run(['g++', '-O2', '-ffp-contract=off', '-mfma', '-shared', '-fPIC', '-I',
                    os.path.join(REPO, 'ballermixplus_amd', 'csrc'), '-o', str(so), str(src)], check=True)
    L = C.CDLL(str(so))
    L.h_crlog.restype = C.c_double
    L.h_crlog.argtypes = [C.c_double]
    L.h_lbeta.restype = C.c_double
    L.h_lbeta.argtypes = [C.c_double, C.c_double]
    L.h_pmf.restype = C.c_double
    L.h_pmf.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
    return L


def test_device_log_is_correctly_rounded(hostmath):
    mpmath = pytest.importorskip('mpmath')
    mpmath.mp.prec = 300
    rng = np.random.default_rng(7)
    xs = np.concatenate([np.exp(rng.uniform(-40, 40, 3000)), rng.uniform(0.5, 2, 3000),
                         rng.uniform(1e9, 4e10, 3000), np.arange(1, 200, dtype=float)])
    for x in xs:
        assert hostmath.h_crlog(float(x)) == float(mpmath.log(mpmath.mpf(float(x))))


def test_device_betaln_is_bit_identical_to_scipy(hostmath):
    """Same operation order as scipy 1.15.3's xsf/cephes lbeta -> bit-equal, the 1e9 noise included."""
    xs = [.05 * i for i in range(1, 11)]
    ab = [0.001, 0.05, 0.5, 1, 3, 9, 45, 95, 200, 500, 1e3, 1e4, 1e6, 1e9]
    args = []
    for n in (50, 100):
        for a in ab:
            for x0 in xs:
                for x in (x0, 1. - x0):
                    b = a / x - a
                    args.append((float(a), b))
                    args += [(k + a, n - k + b) for k in range(0, n + 1, 3)]
        args += [(float(n - k + 1), float(k + 1)) for k in range(n + 1)]
    A = np.array(args)
    ref = betaln(A[:, 0], A[:, 1])
    mine = np.array([hostmath.h_lbeta(p, q) for p, q in args])
    assert np.array_equal(ref, mine)


def test_device_pmf_matches_scipy(hostmath):
    from scipy.stats import betabinom
    for n, a, x in [(50, 1e9, 0.05), (100, 1e6, 0.5), (50, 0.001, 0.3), (100, 20, 0.95), (200, 1e4, 0.45)]:
        b = a / x - a
        k = np.arange(n + 1)
        ref = betabinom(n, a, b).pmf(k)
        mine = np.array([hostmath.h_pmf(int(kk), n, a, b) for kk in k])
        ok = ref > 1e-300
        assert np.max(np.abs(mine[ok] - ref[ok]) / ref[ok]) < 5e-16 * 8
