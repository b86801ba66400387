"""Shared helpers for the tests: fixture paths, oracle loading, model construction."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(HERE, 'golden')
REFT = os.path.join(GOLD, 'ref_test')
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from ballermixplus_amd.hostmodel import Grids, InputData, NeutralSFS  # noqa: E402
from oracle import bmx_oracle as orc  # noqa: E402

STAT_ID = {'B2': 0, 'B2maf': 1, 'B0': 2, 'B0maf': 3, 'B1': 4}
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def c_oracle():
    """The plain-C oracle (oracle/bmx_oracle.c), built on demand with gcc."""
    so = os.path.join(REPO, 'oracle', 'libbmx_oracle.so')
    src = os.path.join(REPO, 'oracle', 'bmx_oracle.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(['make', '-C', os.path.join(REPO, 'oracle')], check=True, stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.orc_lbeta.restype = C.c_double
    L.orc_lbeta.argtypes = [C.c_double, C.c_double]
    L.orc_lgam.restype = C.c_double
    L.orc_lgam.argtypes = [C.c_double]
    L.orc_betabinom_pmf.restype = C.c_double
    L.orc_betabinom_pmf.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
    L.orc_sel_table.argtypes = [C.c_int] * 4 + [_dp, C.c_int, _dp, _dp]
    L.orc_scan.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int64, _dp, _ip, C.c_int64, _dp, _lp, _lp,
                           _dp, _ip, _ip, _ip, _ip]
    return L


def c_sel_table(L, stat, n, min_count, xs, abetas):
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    ab = np.ascontiguousarray(abetas, dtype=np.float64)
    rows = 2 if stat == 'B1' else n + 1
    out = np.zeros((len(xs), len(ab), rows))
    L.orc_sel_table(STAT_ID[stat], n, int(min_count), len(xs), xs.ctypes.data_as(_dp), len(ab), ab.ctypes.data_as(_dp),
                    out.ctypes.data_as(_dp))
    return out


def c_scan(L, R, As, genpos, row, test_gen, lo, hi):
    """R[nx][nab][rows] -> (clr, ix, ia, iA, ns) through the C oracle (OpenMP over test sites)."""
    R = np.ascontiguousarray(R, dtype=np.float64)
    nx, nab, rows = R.shape
    As = np.ascontiguousarray(As, dtype=np.float64)
    g = np.ascontiguousarray(genpos, dtype=np.float64)
    r = np.ascontiguousarray(row, dtype=np.int32)
    t = np.ascontiguousarray(test_gen, dtype=np.float64)
    lo = np.ascontiguousarray(lo, dtype=np.int64)
    hi = np.ascontiguousarray(hi, dtype=np.int64)
    M = len(t)
    clr = np.zeros(M)
    ix, ia, iA, ns = (np.zeros(M, dtype=np.int32) for _ in range(4))
    L.orc_scan(nx, nab, rows, R.ctypes.data_as(_dp), As.ctypes.data_as(_dp), len(As), len(g), g.ctypes.data_as(_dp),
               r.ctypes.data_as(_ip), M, t.ctypes.data_as(_dp), lo.ctypes.data_as(_lp), hi.ctypes.data_as(_lp),
               clr.ctypes.data_as(_dp), ix.ctypes.data_as(_ip), ia.ctypes.data_as(_ip), iA.ctypes.data_as(_ip),
               ns.ctypes.data_as(_ip))
    return clr, ix, ia, iA, ns


def oracle_R(stat, sizes, min_count, spect, props, xs, abetas):
    """R[nx][nab][rows] = P_sel * prop / g - 1 built by the ORACLE alone (oracle/bmx_oracle.py sel_table: scipy's betabinom,
    as the reference calls it, v1:319-433) -- nothing from the GPU's K1, so that a scan compared against c_scan(R=this) is
    checked K1 -> K2 -> finalize end to end.  Rows: one block of n + 1 per sample size, ascending n (2 for B_1)."""
    tabs, g, pr = [], [], []
    for n in sorted(int(v) for v in sizes):
        t = orc.sel_table(stat, n, min_count, list(xs), list(abetas))
        tabs.append(t)
        for k in range(t.shape[2]):
            g.append(spect.get((k, n), np.nan))
            pr.append(props[n])
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.concatenate(tabs, axis=2) * np.array(pr) / np.array(g) - 1.0


def stat_of(nofreq, MAF, nosub):
    if nofreq:
        return 'B1'
    if MAF:
        return 'B0maf' if nosub else 'B2maf'
    return 'B0' if nosub else 'B2'


class Case:
    """One reference-style run: input + helper file + flags, with the oracle model on demand."""

    def __init__(self, infile, spectfile, nofreq=False, MAF=False, nosub=False, phys=False, Rrate=1e-6,
                 x=None, abeta=None, bal=False, pos=False, seqA=None, listA=None):
        self.flags = dict(nofreq=nofreq, MAF=MAF, nosub=nosub)
        self.data = InputData(infile, nofreq, MAF, nosub, 1, phys=phys, Rrate=Rrate)
        self.neut = NeutralSFS(spectfile, nofreq, MAF, nosub)
        self.neut.get_neut_probs(self.data)
        self.grid = Grids(x, abeta, bal, pos, seqA, listA)
        self.stat = stat_of(nofreq, MAF, nosub)
        self.xs, self.abetas, self.As = self.grid.scan_order()
        self._model = None

    def oracle_model(self):
        if self._model is None:
            d = self.data
            self._model = orc.Model(self.stat, d.genPos, d.count, d.total, self.neut.spect, self.neut.sampProps,
                                    d.minCount, self.xs, self.abetas, self.As)
        return self._model


def read_tsv(path):
    """Rows of a reference output file (CR stripped), header dropped."""
    with open(path) as f:
        lines = [l.rstrip('\r\n') for l in f]
    return [l.split('\t') for l in lines[1:] if l]


def rel_close(a, b, rtol=1e-6, atol=1e-9):
    return abs(a - b) <= max(atol, rtol * abs(b))


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)
