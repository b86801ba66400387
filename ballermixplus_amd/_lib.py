"""ctypes binding of libbmxscan.so (C ABI declared in include/bmxscan.h).

There is deliberately no fallback: if the HIP library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C ballermixplus_amd/csrc`)
importing this module raises, and every compute call raises when no GPU is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get('BMX_LIB_NAME', 'libbmxscan.so'))   # BMX_LIB_NAME: A/B builds


class BmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libbmxscan error %d: %s' % (code, msg))
        self.code = code


class BmxModel(C.Structure):
    _fields_ = [
        ('stat', C.c_int32), ('min_count', C.c_int32), ('n_sizes', C.c_int32),
        ('sizes', C.POINTER(C.c_int32)), ('row_off', C.POINTER(C.c_int32)),
        ('g', C.POINTER(C.c_double)), ('prop', C.POINTER(C.c_double)),
        ('nx', C.c_int32), ('x', C.POINTER(C.c_double)),
        ('nab', C.c_int32), ('abeta', C.POINTER(C.c_double)),
    ]


class BmxRecord(C.Structure):
    """bmx_record: one result row as the multi-GPU gather moves it (16 bytes)."""
    _fields_ = [('clr', C.c_double), ('lin', C.c_int32), ('nsites', C.c_int32)]


RECORD_DTYPE = np.dtype([('clr', '<f8'), ('lin', '<i4'), ('nsites', '<i4')])

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/bmxscan.h declares
PROTOTYPES = {
    'bmx_version': (None, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'bmx_build_id': (C.c_char_p, []),
    'bmx_last_error': (C.c_char_p, []),
    'bmx_device_count': (C.c_int, []),
    'bmx_alpha_cut': (C.c_double, []),
    'bmx_lut_build': (C.c_int, [C.POINTER(BmxModel), _dp, _dp, C.c_int]),
    'bmx_scan': (C.c_int, [C.POINTER(BmxModel), _dp, C.c_int32, C.c_int64, _dp, _ip, C.c_int64, _dp, _lp, _lp,
                           _dp, _ip, _ip, _ip, _ip, C.c_int]),
    'bmx_scan_multi': (C.c_int, [C.POINTER(BmxModel), _dp, C.c_int32, C.c_int64, _dp, _ip, C.c_int64, _dp, _lp, _lp,
                                 _dp, _ip, _ip, _ip, _ip, C.c_int32, _ip]),
    'bmx_ctx_create': (C.c_int, [C.POINTER(_vp), C.c_int]),
    'bmx_ctx_destroy': (None, [_vp]),
    'bmx_ctx_set_model': (C.c_int, [_vp, C.POINTER(BmxModel), _dp, C.c_int32]),
    'bmx_ctx_set_sites': (C.c_int, [_vp, C.c_int64, _dp, _ip]),
    'bmx_ctx_set_tests': (C.c_int, [_vp, C.c_int64, _dp, _lp, _lp]),
    'bmx_ctx_scan': (C.c_int, [_vp]),
    'bmx_ctx_sync': (C.c_int, [_vp]),
    'bmx_ctx_last_scan_ms': (C.c_int, [_vp, _dp]),
    'bmx_ctx_fetch': (C.c_int, [_vp, _dp, _ip, _ip, _ip, _ip]),
    'bmx_ctx_result_ptrs': (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    'bmx_ctx_records': (C.c_int, [_vp, C.POINTER(_vp)]),
    'bmx_ctx_fetch_records': (C.c_int, [_vp, _vp]),
    'bmx_ctx_scan_write': (C.c_int, [_vp, C.c_char_p, _lp, _dp, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                     C.c_char_p, C.c_int, C.c_int64]),
    'bmx_ctx_fetch_lut': (C.c_int, [_vp, _dp, _dp]),
    'bmx_ctx_set_variant': (C.c_int, [_vp, C.c_int]),
    'bmx_ctx_select_slot': (C.c_int, [_vp, C.c_int32]),
    'bmx_ctx_slot_count': (C.c_int, [_vp]),
    'bmx_ctx_pack_records': (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _lp]),
    'bmx_ctx_copy_records': (C.c_int, [_vp, _vp, C.c_int64]),
    'bmx_ctx_plan': (C.c_int, [_vp, _ip, _ip, _ip, _lp]),
    'bmx_ctx_launch_ranges': (C.c_int, [_vp, _lp, C.c_int32, _ip]),
    'bmx_ctx_surface': (C.c_int, [_vp, C.c_double, C.c_int64, C.c_int64, _dp, _ip]),
    'bmx_comm_unique_id': (C.c_int, [C.c_char_p]),
    'bmx_comm_create': (C.c_int, [C.POINTER(_vp), _vp, C.c_char_p, C.c_int32, C.c_int32]),
    'bmx_comm_destroy': (None, [_vp]),
    'bmx_comm_gather_records': (C.c_int, [_vp, _lp, C.c_int32, _vp, C.POINTER(_vp)]),
    'bmx_input_count': (C.c_int, [C.c_char_p, _lp]),
    'bmx_input_parse': (C.c_int, [C.c_char_p, C.c_int64, C.c_int, _lp, _dp, _lp, _lp]),
    'bmx_write_rows': (C.c_int, [C.c_char_p, C.c_int64, _lp, _dp, _dp, _ip, _ip, _ip, _ip, C.c_char_p, C.c_int,
                                 C.c_char_p, C.c_int, C.c_char_p, C.c_int]),
    'bmx_write_records': (C.c_int, [C.c_char_p, C.c_int64, _lp, _dp, C.POINTER(_vp), C.c_int32, C.c_int64, C.c_char_p, C.c_int,
                                    C.c_char_p, C.c_int, C.c_char_p, C.c_int]),
    'bmx_py_repr': (C.c_int, [C.c_double, C.c_char_p]),
}

_lib = None
SOURCES = ('csrc/bmxscan.hip', 'csrc/bmx_io.cpp', 'csrc/bmx_math.h', '../include/bmxscan.h')   # the Makefile's SRCS, in order


def source_id():
    """What bmx_build_id() of a library built from the sources in this tree returns."""
    import hashlib
    h = hashlib.sha256()
    for rel in SOURCES:
        with open(os.path.join(_HERE, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def lib():
    """Load libbmxscan.so once; raise if it is missing (no CPU fallback exists) or was built from other sources
    than the ones next to it (a stale binary must not be tested or benchmarked silently)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('libbmxscan.so not built: %s is missing. Run `make -C %s` '
                              '(needs hipcc); there is no CPU fallback.' % (LIB_PATH, os.path.join(_HERE, 'csrc')))
        L = C.CDLL(LIB_PATH)
        lenient = os.environ.get('BMX_ALLOW_STALE') == '1'       # A/B runs against older builds: entry points they lack stay unbound
        for name, (res, args) in PROTOTYPES.items():
            if lenient and not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        built, want = L.bmx_build_id().decode(), source_id()
        if built != want and os.environ.get('BMX_ALLOW_STALE') != '1':
            raise ImportError('%s is stale: built from sources %s, the tree has %s. Run `make -C %s`.'
                              % (LIB_PATH, built, want, os.path.join(_HERE, 'csrc')))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise BmxError(rc, lib().bmx_last_error().decode('utf-8', 'replace'))


def as_dp(a):
    return a.ctypes.data_as(_dp)


def as_ip(a):
    return a.ctypes.data_as(_ip)


def as_lp(a):
    return a.ctypes.data_as(_lp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def read_input(path, pos_col):
    """(phys i64[N], coord f64[N], k i64[N], n i64[N]) through the native reader."""
    L = lib()
    n = C.c_int64()
    check(L.bmx_input_count(path.encode(), C.byref(n)))
    N = n.value
    phys = np.empty(N, dtype=np.int64)
    coord = np.empty(N, dtype=np.float64)
    k = np.empty(N, dtype=np.int64)
    nn = np.empty(N, dtype=np.int64)
    if N:
        check(L.bmx_input_parse(path.encode(), N, int(pos_col), as_lp(phys), as_dp(coord), as_lp(k), as_lp(nn)))
    return phys, coord, k, nn


def write_rows(path, phys, gen, clr, ix, ia, iA, ns, xs, abs_, As):
    """Append result rows through the native writer (Python-repr-exact floats)."""
    L = lib()
    phys, gen, clr = i64(phys), f64(gen), f64(clr)
    ix, ia, iA, ns = i32(ix), i32(ia), i32(iA), i32(ns)
    pack = lambda v: b'\0'.join(s.encode() for s in v) + b'\0'
    check(L.bmx_write_rows(path.encode(), len(phys), as_lp(phys), as_dp(gen), as_dp(clr), as_ip(ix), as_ip(ia),
                           as_ip(iA), as_ip(ns), pack(xs), len(xs), pack(abs_), len(abs_), pack(As), len(As)))


def write_records(path, phys, gen, per_rank, block, xs, abs_, As):
    """Append the rows of a sharded run from its gathered 16-byte records: per_rank[r] = RECORD array of rank r (its test
    sites in its own order; blocks of `block` test sites dealt round-robin).  One rank: per_rank = [records in order]."""
    L = lib()
    phys, gen = i64(phys), f64(gen)
    keep = [np.ascontiguousarray(a, dtype=RECORD_DTYPE) for a in per_rank]
    ptrs = (_vp * len(keep))(*[a.ctypes.data_as(_vp) for a in keep])
    pack = lambda v: b'\0'.join(s.encode() for s in v) + b'\0'
    check(L.bmx_write_records(path.encode(), len(phys), as_lp(phys), as_dp(gen), ptrs, len(keep), int(block), pack(xs), len(xs),
                              pack(abs_), len(abs_), pack(As), len(As)))
