"""The oracle is only trustworthy once pinned: every restatement in oracle/ is checked here
against outputs of the reference itself (tests/golden/*, made by tests/golden/make_golden.py)
and against the reference's own example outputs (tests/golden/ref_test/output)."""
import glob
import os

import numpy as np
import pytest
from scipy.special import betaln

import cases
from util import GOLD, c_oracle, c_scan, c_sel_table, orc, read_tsv

from ballermixplus_amd import scan as scanmod

LUTS = sorted(glob.glob(os.path.join(GOLD, 'lut_*.npz')))


def _lut_meta(path):
    name = os.path.basename(path)[4:-4].split('_')
    return name[0], np.load(path)


@pytest.mark.parametrize('path', LUTS, ids=[os.path.basename(p)[4:-4] for p in LUTS])
def test_numpy_oracle_table_is_bit_identical_to_reference(path):
    stat, z = _lut_meta(path)
    xs, ab, minc = z['x'].tolist(), z['abeta'].tolist(), int(z['minCount'])
    for n in sorted(set(z['total'].tolist())):
        mine = orc.sel_table(stat, n, minc, xs, ab)
        sel = z['total'] == n
        assert np.array_equal(z['table'][:, :, sel], mine[:, :, z['count'][sel]])


@pytest.mark.parametrize('path', LUTS, ids=[os.path.basename(p)[4:-4] for p in LUTS])
def test_c_oracle_table_matches_reference(path):
    L = c_oracle()
    stat, z = _lut_meta(path)
    xs, ab, minc = z['x'], z['abeta'], int(z['minCount'])
    for n in sorted(set(z['total'].tolist())):
        mine = c_sel_table(L, stat, n, minc, xs, ab)
        sel = z['total'] == n
        ref = z['table'][:, :, sel]
        got = mine[:, :, z['count'][sel]]
        # differences are libm-vs-numpy exp/log only (<= a few ulp, amplified where normBase -> 0)
        assert np.max(np.abs(ref - got) / np.abs(ref)) < 1e-12


def test_c_cephes_betaln_is_bit_identical_to_scipy():
    """scipy.special.betaln is what scipy.stats.betabinom calls (reference v1:369-371)."""
    L = c_oracle()
    xs = [.05 * i for i in range(1, 11)]
    ab = [0.001, 0.01, 0.05, 0.1, 0.5, 1, 2, 5, 9, 50, 95, 200, 500, 1e3, 1e4, 1e6, 1e9]
    args = []
    for n in (50, 100):
        for a in ab:
            for x0 in xs:
                for x in (x0, 1. - x0):
                    b = a / x - a
                    args.append((float(a), b))
                    args += [(k + a, n - k + b) for k in range(n + 1)]
        args += [(float(n - k + 1), float(k + 1)) for k in range(n + 1)]
    A = np.array(args)
    ref = betaln(A[:, 0], A[:, 1])
    mine = np.array([L.orc_lbeta(p, q) for p, q in args])
    assert np.array_equal(ref, mine)


SURF = sorted(glob.glob(os.path.join(GOLD, 'surface_*.npz')))


@pytest.mark.parametrize('path', SURF, ids=[os.path.basename(p)[8:-4] for p in SURF])
def test_lut_form_reproduces_reference_likelihood_surface(path):
    """T[A,x,a] of the log1p/LUT form == the reference's calcBaller called per grid point."""
    z = np.load(path)
    name = os.path.basename(path)
    key = 'ex1_B2' if 'ex1_B2' in name else 'ex2_B2maf_findBal'
    argv = cases.ALL_CASES[key][0]
    opt, case, ts = cases.host_side(argv)
    assert [float(v) for v in case.xs] == z['x'].tolist()
    assert [float(v) for v in case.abetas] == z['abeta'].tolist()
    assert [float(v) for v in case.As] == z['A'].tolist()
    m = case.oracle_model()
    s = int(z['site'])
    best, Ts, ns = orc.clr_lut(m, 0, m.N - 1, case.data.genPos[s], surface=True)
    ref = z['T']
    pos = ~np.isnan(ref)            # the reference reports a value only where T > 0
    assert np.all((Ts[~pos] <= 0) | np.isnan(Ts[~pos]))
    assert np.max(np.abs(Ts[pos] - ref[pos]) / np.abs(ref[pos])) < 1e-9
    has = pos.any(axis=(1, 2))
    assert np.array_equal(ns[has], z['nsites'][has])
    b = z['best']
    assert abs(best[0] - b[0]) <= 1e-9 * abs(b[0])
    assert (float(case.xs[best[1]]), float(case.abetas[best[2]]), float(case.As[best[3]]), best[4]) == \
        (b[1], b[2], b[3], int(b[4]))


def _faithful_rows(case, ts, idx):
    m = case.oracle_model()
    out = []
    for j in idx:
        T, ix, ia, iA, ns = orc.calc_baller_faithful(m, ts.lo[j], ts.hi[j], ts.test_gen[j])
        out.append((T, ix, ia, iA, ns))
    return out


@pytest.mark.parametrize('name', ['ex1_B2', 'ex2_B2', 'ex2_B2maf', 'ex1_B1', 'ex2_B0maf_1kb', 'ex1_B2_w50_s25',
                                  'ex2_B0_noCenter_2kb'])
def test_faithful_port_reproduces_reference_rows(name):
    """calc_baller_faithful (the timed CPU baseline) on a strided subset of each golden file."""
    argv, gold = cases.ALL_CASES[name]
    opt, case, ts = cases.host_side(argv)
    rows = read_tsv(gold)
    scanned = [i for i in range(len(rows)) if i not in ts.na_rows]
    assert len(scanned) == len(ts)
    pick = list(range(0, len(ts), max(1, len(ts) // 6)))
    class S:  # grids for printing
        grid_x, grid_abeta, grid_A = case.xs, case.abetas, case.As
    for j, (T, ix, ia, iA, ns) in zip(pick, _faithful_rows(case, ts, pick)):
        line = scanmod.format_row(ts.phys[j], ts.gen_label[j], T, ix, ia, iA, ns, S).rstrip('\n').split('\t')
        g = rows[ts.order[j]]
        assert line[:2] == g[:2] and line[3:] == g[3:], (j, line, g)
        assert abs(float(line[2]) - float(g[2])) <= 1e-11 * abs(float(g[2])) + 1e-13


@pytest.mark.parametrize('name', sorted(cases.ALL_CASES))
def test_c_oracle_scan_reproduces_every_golden_file(name):
    """Whole files through the C oracle (log1p/LUT form, numpy-built table): every row, every field."""
    argv, gold = cases.ALL_CASES[name]
    if not os.path.exists(gold):
        pytest.skip('fixture not generated')
    opt, case, ts = cases.host_side(argv)
    m = case.oracle_model()
    L = c_oracle()
    clr, ix, ia, iA, ns = c_scan(L, m.R, case.As, case.data.genPos, m.row, ts.test_gen, ts.lo, ts.hi)
    class S:
        grid_x, grid_abeta, grid_A = case.xs, case.abetas, case.As
    lines = [None] * (len(ts) + len(ts.na_rows))
    for p, l in ts.na_rows.items():
        lines[p] = l
    for j, p in enumerate(ts.order):
        lines[p] = scanmod.format_row(ts.phys[j], ts.gen_label[j], clr[j], int(ix[j]), int(ia[j]), int(iA[j]), ns[j], S)
    worst, ties = cases.compare_rows(lines, gold, rtol=1e-9, case=case, ts=ts)
    assert worst < 1e-9
    # ties within rounding noise only where the table saturates (B_1); none on the B_2/B_0 files
    assert ties == 0 or name.endswith('B1'), ties


def test_alpha_cut_matches_numpy_predicate():
    z = orc.alpha_cut_z()
    assert np.exp(-np.array([z]))[0] >= 1e-8 > np.exp(-np.array([np.nextafter(z, 100.0)]))[0]
    assert z == 18.420680743952364
