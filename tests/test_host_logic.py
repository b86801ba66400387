"""Host-side mirror of the reference (no GPU): grids and their set() iteration order, input
parsing, helper-file generation (byte-exact), window generation for the four Scan modes."""
import filecmp
import os

import numpy as np
import pytest

import cases
from util import GOLD, REFT, load_json, read_tsv

from ballermixplus_amd import _lib, helpers
from ballermixplus_amd.hostmodel import Grids, InputData, NeutralSFS


def _r(v):
    return [repr(x) for x in v]


def _declined(*a, **k):
    raise _lib.BmxError(-1, 'native reader declined (test)')


def test_grid_lists_and_iteration_order_match_reference():
    so = load_json('setorder.json')
    g = Grids(None, None, False, False, None, None)
    assert _r(g.x) == so['default']['x_list'] and _r(g.abeta) == so['default']['abeta_list']
    assert _r(g.A) == so['default']['A_list']
    xs, ab, As = g.scan_order()
    assert (_r(xs), _r(ab), _r(As)) == (so['default']['x'], so['default']['abeta'], so['default']['A'])
    assert len(ab) == 51 and len(As) == 31                      # the duplicated 5 collapses (v1:157)
    g = Grids(None, None, True, False, None, None)
    xs, ab, As = g.scan_order()
    assert (_r(xs), _r(ab), _r(As)) == (so['bal']['x'], so['bal']['abeta'], so['bal']['A'])
    g = Grids(None, None, True, True, None, so['config5']['listA'])   # --findBal wins over --findPos
    xs, ab, As = g.scan_order()
    assert (_r(xs), _r(ab), _r(As)) == (so['config5']['x'], so['config5']['abeta'], so['config5']['A'])
    g = Grids('0.3', 7.0, False, False, None, '250,1e3,77.5')
    xs, ab, As = g.scan_order()
    assert (_r(xs), _r(ab), _r(As)) == (so['fixed']['x'], so['fixed']['abeta'], so['fixed']['A'])


def test_rangeA_implements_the_evident_intent():
    """The reference raises on --rangeA (float range + typo, v1:169-171); ours yields the
    float grid Amin, Amin+step, ..., Amax -- the same values --listA gives (config 5)."""
    g = Grids(None, None, True, True, '100,10000,100', None)
    assert g.A == [float(100 * i) for i in range(1, 101)]


@pytest.mark.parametrize('out,inp,kw', [
    ('spect_ex1_DAF.txt', 'Example1_fullSweep_200kya_DAF.txt', dict(MAF=False, nosub=False)),
    ('spect_ex1_MAFfold.txt', 'Example1_fullSweep_200kya_DAF.txt', dict(MAF=True, nosub=False)),
    ('spect_ex2_MAF.txt', 'Example2_balancing_10MYA_MAF.txt', dict(MAF=True, nosub=False)),
    ('spect_ex2_DAF_nosub.txt', 'Example2_balancing_10MYA_DAF.txt', dict(MAF=False, nosub=True)),
    ('spect_ex2_MAF_nosub.txt', 'Example2_balancing_10MYA_MAF.txt', dict(MAF=True, nosub=True)),
])
def test_getSpect_is_byte_identical(tmp_path, out, inp, kw, monkeypatch):
    dst = tmp_path / out
    helpers.getSpect(os.path.join(REFT, inp), str(dst), kw['MAF'], kw['nosub'])          # columns from the native reader
    assert filecmp.cmp(str(dst), os.path.join(GOLD, 'helpers', out), shallow=False)
    monkeypatch.setattr(_lib, 'read_input', _declined)                                    # ... and from Python's converters
    dst2 = tmp_path / ('py_' + out)
    helpers.getSpect(os.path.join(REFT, inp), str(dst2), kw['MAF'], kw['nosub'])
    assert filecmp.cmp(str(dst2), os.path.join(GOLD, 'helpers', out), shallow=False)


@pytest.mark.parametrize('out,inp', [('config_ex1.txt', 'Example1_fullSweep_200kya_DAF.txt'),
                                     ('config_ex2.txt', 'Example2_balancing_10MYA_DAF.txt')])
def test_getConfig_is_byte_identical(tmp_path, out, inp, monkeypatch):
    dst = tmp_path / out
    helpers.getConfig(os.path.join(REFT, inp), str(dst))
    assert filecmp.cmp(str(dst), os.path.join(GOLD, 'helpers', out), shallow=False)
    monkeypatch.setattr(_lib, 'read_input', _declined)
    dst2 = tmp_path / ('py_' + out)
    helpers.getConfig(os.path.join(REFT, inp), str(dst2))
    assert filecmp.cmp(str(dst2), os.path.join(GOLD, 'helpers', out), shallow=False)


def test_input_parsing_semantics():
    d = InputData(os.path.join(REFT, 'Example1_fullSweep_200kya_DAF.txt'))
    assert d.numSites == 757 and d.minCount == 1 and d.sampSizes == {50}
    assert d.position[0] == 12 and d.genPos[0] == 1.2e-07 and d.count[0] == 50
    m = InputData(os.path.join(REFT, 'Example1_fullSweep_200kya_DAF.txt'), MAF=True)
    assert m.count.max() <= 25 and m.count[0] == 0 and m.minCount == 1      # folded, k=n -> 0
    p = InputData(os.path.join(REFT, 'Example1_fullSweep_200kya_DAF.txt'), phys=True, Rrate=2e-6)
    assert p.genPos[0] == 12.0 * 2e-6
    ns = InputData(os.path.join(REFT, 'Example2_balancing_10MYA_DAF.txt'), nosub=True)
    assert np.all(ns.count != ns.total) and ns.numSites == len(ns.count) == 285
    b1 = InputData(os.path.join(REFT, 'Example1_fullSweep_200kya_DAF.txt'), nofreq=True)
    assert set(b1.count.tolist()) == {0, 1}


def test_neutral_model_semantics():
    n = NeutralSFS(os.path.join(REFT, 'HC_CEU_Neut_DAF_spect_for_B2.txt'), False, False, False)
    assert len(n.spect) == 50 and abs(n.sampProps[50] - 1.0) < 1e-12
    c = NeutralSFS(os.path.join(REFT, 'HC_CEU_Neut_config_for_B1.txt'), True, False, False)
    assert c.spect == {(0, 50): 0.7237520358450141, (1, 50): 0.27624796415498587}
    with pytest.raises(SystemExit):          # (k,n) not covered by the helper file (v1:281-287)
        d = InputData(os.path.join(REFT, 'Example1_fullSweep_200kya_DAF.txt'))
        NeutralSFS(os.path.join(REFT, 'HC_CEU_Neut_DAF-nosub_spect_for_B0.txt'), False, False, False).get_neut_probs(d)


@pytest.mark.parametrize('name', sorted(cases.ALL_CASES))
def test_window_generation_visits_the_reference_test_sites(name):
    """Test positions, NA rows and their order must equal the golden file's first two columns."""
    argv, gold = cases.ALL_CASES[name]
    opt, case, ts = cases.host_side(argv)
    rows = read_tsv(gold)
    assert len(rows) == len(ts) + len(ts.na_rows)
    from ballermixplus_amd import scan as scanmod
    class S:
        grid_x, grid_abeta, grid_A = case.xs, case.abetas, case.As
    for j, p in enumerate(ts.order):
        line = scanmod.format_row(ts.phys[j], ts.gen_label[j], 0.0, 0, 0, 0, 0, S).split('\t')
        assert line[:2] == rows[p][:2]
        assert 0 <= ts.lo[j] and ts.hi[j] <= case.data.numSites - 1
    for p, l in ts.na_rows.items():
        assert l.rstrip('\n').split('\t') == rows[p]


def test_window_bounds_of_each_mode():
    d = InputData(os.path.join(REFT, 'Example1_fullSweep_200kya_DAF.txt'))
    from ballermixplus_amd import scan as scanmod
    ts = scanmod.sites_site_based(d, 50, 25.0)            # -w 50 -s 25: r left, r+1 right (v1:588)
    assert (ts.lo[0], ts.hi[0]) == (0, 51) and (ts.lo[4], ts.hi[4]) == (50, 151)
    assert ts.hi[-1] == d.numSites - 1
    ts = scanmod.sites_alpha(d, 20.0)
    assert len(ts) == 38 and set(ts.lo) == {0} and set(ts.hi) == {756}
    p = InputData(os.path.join(REFT, 'Example2_balancing_10MYA_DAF.txt'), phys=True)
    ts = scanmod.sites_fix_center(p, 5000.0, 40.0)
    for j in range(len(ts)):                               # window = sites within +-2.5 kb, end inclusive
        i = 40 * j
        lo, hi = ts.lo[j], ts.hi[j]
        assert p.position[lo] >= p.position[i] - 2500 and (lo == 0 or p.position[lo - 1] < p.position[i] - 2500)
        assert hi == len(p.position) - 1 or p.position[hi] >= min(p.position[i] + 2500, p.position[-1])


@pytest.mark.parametrize('fname,kw', [
    ('Example1_fullSweep_200kya_DAF.txt', {}),
    ('Example1_fullSweep_200kya_DAF.txt', dict(nofreq=True)),
    ('Example2_balancing_10MYA_MAF_nosub.txt', dict(phys=True, Rrate=1.25e-6)),
    ('Example2_balancing_10MYA_DAF.txt', dict(MAF=True)),
])
def test_native_reader_equals_a_per_line_parse(fname, kw, monkeypatch):
    """libbmxscan's mmap/strtod reader and the column-wise Python converters against a plain per-line parse
    with the conversions the reference applies (v1:103-104, 121-124), written out here."""
    from ballermixplus_amd import hostmodel
    path = os.path.join(REFT, fname)
    pos_col = 1 - int(kw.get('phys', False))
    with open(path) as fh:
        lines = fh.read().splitlines()[1:]
    want = ([], [], [], [])
    for ln in lines:
        c = ln.strip().split('\t')
        want[0].append(int(float(c[0]))); want[1].append(float(c[pos_col])); want[2].append(int(c[2])); want[3].append(int(c[3]))
    nat = hostmodel.read_columns(path, pos_col)
    monkeypatch.setattr(_lib, 'read_input', _declined)
    py = hostmodel.read_columns(path, pos_col)
    for got in (nat, py):
        assert all(np.array_equal(a, np.array(b)) for a, b in zip(got, want))
        assert got[1].dtype == np.float64 and got[0].dtype == np.int64 and got[2].dtype == np.int64


def test_hostmodel_state_equals_the_reference_classes(capsys):
    """NeutralSFS and InputData after construction against what the reference's own classes hold (tests/golden/hostmodel.json, made
    by importing the reference: make_golden.py hostmodel): spectrum, per-size proportions and arrays BITWISE, stdout identical."""
    import hashlib
    gold = load_json('hostmodel.json')
    for c in gold['neutral']:
        capsys.readouterr()
        n = NeutralSFS(os.path.join(REFT, c['file']), *c['args'])
        assert capsys.readouterr().out == c['stdout'], c['file']
        assert [[k, m, repr(v)] for (k, m), v in sorted(n.spect.items())] == c['spect'], (c['file'], c['args'])
        assert [[m, repr(v)] for m, v in sorted(n.sampProps.items())] == c['sampProps'] and sorted(n.sampSizes) == c['sampSizes']
    h = lambda a, dt: hashlib.sha256(np.ascontiguousarray(np.asarray(a), dtype=dt).tobytes()).hexdigest()
    for c in gold['input']:
        capsys.readouterr()
        d = InputData(os.path.join(REFT, c['file']), **c['kw'])
        assert capsys.readouterr().out == c['stdout'], c['file']
        assert (d.numSites, d.minCount, sorted(d.sampSizes)) == (c['numSites'], c['minCount'], c['sampSizes']), (c['file'], c['kw'])
        assert h(d.position, np.int64) == c['position_sha256'] and h(d.genPos, np.float64) == c['genPos_sha256']
        assert h(d.count, np.int64) == c['count_sha256'] and h(d.total, np.int64) == c['total_sha256']


def test_native_reader_edge_cases(tmp_path):
    from ballermixplus_amd import _lib
    f = tmp_path / 'a.txt'
    f.write_text('physPos\tgenPos\tx\tn\n12.0\t1.2e-07\t50\t50\r\n165\t1.65e-06\t3\t50')   # CRLF + no final newline
    ph, co, k, n = _lib.read_input(str(f), 1)
    assert ph.tolist() == [12, 165] and co.tolist() == [1.2e-07, 1.65e-06] and k.tolist() == [50, 3] and n.tolist() == [50, 50]
    f.write_text('physPos\tgenPos\tx\tn\n')
    assert len(_lib.read_input(str(f), 0)[0]) == 0
    f.write_text('physPos\tgenPos\tx\tn\n12\t0.1\tfoo\t50\n')
    with pytest.raises(_lib.BmxError):
        _lib.read_input(str(f), 1)
    with pytest.raises(_lib.BmxError):
        _lib.read_input(str(tmp_path / 'missing.txt'), 1)
    # an empty last column must not take its value from the next line (strtoll skips '\n'); hex floats, inf and
    # fields with leading blanks are left to the Python reader, which reproduces the reference's behaviour
    for body in ('100\t0.1\t5\t\n200\t0.2\t3\t50\n', '100\t0.1\t\t50\n', '0x10\t0.1\t5\t50\n', 'inf\t0.1\t5\t50\n',
                 '100\t 0.1\t5\t50\n', '100\t0.1\t5.0\t50\n', '100\t0.1\t5\t50junk\n'):
        f.write_text('physPos\tgenPos\tx\tn\n' + body)
        with pytest.raises(_lib.BmxError):
            _lib.read_input(str(f), 1)


def test_native_reader_threads_agree_with_one_range(tmp_path):
    """Files beyond 1 MB are cut into one byte range per thread at line boundaries: same arrays as line by line."""
    from ballermixplus_amd import _lib, synth
    phys, gen, k, n = synth.synth_chromosome(120000, 100, 7)
    f = tmp_path / 'big.txt'
    synth.write_input(str(f), phys, gen, k, n)
    assert f.stat().st_size > 2 << 20
    ph, co, kk, nn = _lib.read_input(str(f), 1)
    assert np.array_equal(ph, phys) and np.array_equal(kk, k) and np.array_equal(nn, n)
    assert np.array_equal(co, np.array([float('%.6f' % g) for g in gen]))
    with open(f, 'rb+') as fh:                      # no final newline, and a malformed line deep in the file
        fh.seek(-1, 2)
        fh.truncate()
    assert np.array_equal(_lib.read_input(str(f), 0)[0], phys)
    lines = f.read_text().split('\n')
    lines[90001] = lines[90001].replace('\t', ' ', 1)
    f.write_text('\n'.join(lines))
    with pytest.raises(_lib.BmxError, match='line 90001'):
        _lib.read_input(str(f), 1)


def test_native_float_formatting_equals_python_repr():
    """bmx_py_repr (used by the native row writer) vs repr(): fixed/exponent switch, shortest digits."""
    import ctypes as C
    import random
    import struct
    from ballermixplus_amd import _lib
    L = _lib.lib()
    buf = C.create_string_buffer(64)

    def r(v):
        L.bmx_py_repr(v, buf)
        return buf.value.decode()
    fixed = [0.0, -0.0, 1.0, 100.0, 1e16, 1e15, 9999999999999998.0, 1.2e-07, 1e-4, 0.0001234, 1e-5, 0.1 + 0.2,
             31.21810547602786, 1e22, 1.7976931348623157e308, 5e-324, 2.2250738585072014e-308,
             123456789012345680.0, 0.05, 0.15000000000000002, 1000000000.0, -3.5e-10, 0.006999999999999999]
    for v in fixed:
        assert r(v) == repr(v), v
    rng = random.Random(3)
    for i in range(40000):
        v = struct.unpack('<d', struct.pack('<Q', rng.getrandbits(64)))[0] if i % 2 else rng.random() * 10.0 ** rng.randint(-12, 20)
        if v == v and abs(v) != float('inf'):
            assert r(v) == repr(v), v


def test_native_row_writer_writes_the_same_bytes(tmp_path):
    from ballermixplus_amd import _lib, scan as scanmod
    d = InputData(os.path.join(REFT, 'Example2_balancing_10MYA_DAF.txt'))
    ts = scanmod.sites_alpha(d, 1)
    g = Grids(None, None, False, False, None, None)

    class S:
        grid_x, grid_abeta, grid_A = g.scan_order()
    N = len(ts)
    rg = np.random.default_rng(2)
    res = (rg.random(N) * 10.0 ** rg.integers(-7, 5, N), rg.integers(0, 10, N).astype(np.int32),
           rg.integers(0, 51, N).astype(np.int32), rg.integers(-1, 31, N).astype(np.int32),
           rg.integers(1, 5000, N).astype(np.int32))
    a, b = tmp_path / 'native.txt', tmp_path / 'python.txt'
    scanmod.write_rows(str(a), ts, res, S)
    keep = _lib.write_rows
    try:
        _lib.write_rows = lambda *x, **k: (_ for _ in ()).throw(ImportError())
        scanmod.write_rows(str(b), ts, res, S)
    finally:
        _lib.write_rows = keep
    assert a.read_bytes() == b.read_bytes()
    rows = a.read_text().splitlines()
    assert rows[0] == 'physPos\tgenPos\tCLR\tx_hat\ts_hat\tA_hat\tnSites' and len(rows) == N + 1


def test_native_row_writer_on_several_threads(tmp_path):
    """Beyond 8192 rows the formatter splits a chunk over host threads (and bmx_write_rows works in chunks of 2^18 rows):
    same bytes as Python's own f-string of the same values, all-zero rows included."""
    from ballermixplus_amd import _lib
    N = 300000
    rg = np.random.default_rng(11)
    phys = np.cumsum(rg.integers(1, 200, N)).astype(np.int64)
    gen = phys / 1e6
    clr = rg.random(N) * 10.0 ** rg.integers(-9, 4, N)
    ix, ia = rg.integers(0, 10, N).astype(np.int32), rg.integers(0, 51, N).astype(np.int32)
    iA, ns = rg.integers(-1, 31, N).astype(np.int32), rg.integers(1, 6000, N).astype(np.int32)
    g = Grids(None, None, False, False, None, None)
    xs, ab, As = g.scan_order()
    sx, sa, sA = [f'{v}' for v in xs], [f'{v}' for v in ab], [f'{v}' for v in As]
    out = tmp_path / 'rows.txt'
    _lib.write_rows(str(out), phys, gen, clr, ix, ia, iA, ns, sx, sa, sA)
    want = ''.join(f'{p}\t{float(q)}\t{float(c)}\t{sx[a]}\t{sa[b]}\t{sA[d]}\t{n}\n' if d >= 0 else f'{p}\t{float(q)}\t0.0\t0.0\t0.0\t0.0\t0.0\n'
                   for p, q, c, a, b, d, n in zip(phys.tolist(), gen.tolist(), clr.tolist(), ix.tolist(), ia.tolist(), iA.tolist(), ns.tolist()))
    assert out.read_text() == want
    iA[12345] = 31                                       # a grid index outside the tables is an error, not a crash
    with pytest.raises(_lib.BmxError):
        _lib.write_rows(str(out), phys, gen, clr, ix, ia, iA, ns, sx, sa, sA)


def test_vectorised_window_generation_equals_the_reference_loops(monkeypatch):
    """sites_site_based / sites_fix_center fast paths vs the reference's per-site loops."""
    from ballermixplus_amd import scan as scanmod
    p = InputData(os.path.join(REFT, 'Example2_balancing_10MYA_DAF.txt'), phys=True)

    def loop_site_based(data, r, s):
        out = []
        i = 0
        while i < data.numSites:
            w = np.arange(max(0, i - r), min(data.numSites - 1, i + r + 1) + 1, dtype=int)
            out.append((int(i), int(w[0]), int(w[-1])))
            i += s
        return out

    def loop_fix_center(data, w, s):
        out = []
        pos = data.position
        i = start_i = end_i = 0
        while i < data.numSites:
            start = max(0, pos[i] - w / 2)
            end = min(pos[i] + w / 2, pos[-1])
            while pos[start_i] < start:
                start_i += 1
            while end_i < data.numSites:
                if pos[end_i] < end:
                    end_i += 1
                else:
                    break
            end_i = min(end_i, data.numSites - 1)
            out.append((int(i), int(start_i), int(end_i)))
            i += int(s)
        return out

    for r, s in [(50, 25.0), (3, 1), (400, 7.0), (2000, 100.0)]:
        ts = scanmod.sites_site_based(p, r, s)
        ref = loop_site_based(p, r, s)
        assert [(a, b) for a, b in zip(ts.lo, ts.hi)] == [(b, c) for _, b, c in ref]
        assert ts.test_gen == [float(p.genPos[i]) for i, _, _ in ref]
    for w, s in [(1000.0, 2.0), (5000.0, 40.0), (10.0, 1), (1e7, 13.0)]:
        ts = scanmod.sites_fix_center(p, w, s)
        ref = loop_fix_center(p, w, s)
        assert [(a, b) for a, b in zip(ts.lo, ts.hi)] == [(b, c) for _, b, c in ref]
