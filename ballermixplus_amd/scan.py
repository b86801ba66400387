"""Scan driver: the reference's `class Scan` (BalLeRMix+_v1.py:510-640) with the per-site
calcBaller loop replaced by one batched GPU launch per input file.

The four window modes only differ in which test sites they visit and which
inclusive index window [start_i, end_i] each one gets; those are generated on the
host exactly as the reference's loops do (they are O(N) two-pointer walks), then all
test sites go to the device together.  Output rows are formatted in Python from
the grid's own objects so that repr matches the reference byte for byte (v1:607).
"""
import sys
from datetime import datetime
from math import floor

import numpy as np

from . import engine

HEADER = 'physPos\tgenPos\tCLR\tx_hat\ts_hat\tA_hat\tnSites\n'


class TestSites:
    """Test sites of one scan, in output order.  `na_rows` holds positions in the output where
    the reference prints an 'NA' row instead of scanning (v1:534-536).  The per-site lists
    (phys, gen_label, test_gen, lo, hi) are materialised lazily when the sites came from one
    vectorised add_many (a million-entry Python list costs more than the GPU scan)."""

    _FIELDS = ('phys', 'gen_label', 'test_gen', 'lo', 'hi')

    def __init__(self):
        self._lists = {k: [] for k in self._FIELDS}
        self.na_rows = {}      # output position -> preformatted line
        self._order = []       # output position of each scanned site (exposed as .order)
        self.arrays = None     # (phys i64, gen f64, lo i64, hi i64) when built by one vectorised add_many
        self._n = 0

    def _materialise(self):
        if self.arrays is not None and not self._lists['phys'] and self._n:
            phys, gen, lo, hi = self.arrays
            g = gen.tolist()
            self._lists = {'phys': phys.tolist(), 'gen_label': g, 'test_gen': g, 'lo': lo.tolist(), 'hi': hi.tolist()}
            self._order = list(range(self._n))

    def __getattr__(self, name):
        if name in TestSites._FIELDS:
            self._materialise()
            return self._lists[name]
        if name == 'order':
            self._materialise()
            return self._order
        raise AttributeError(name)

    def add(self, phys, gen_label, test_gen, lo, hi):
        self._materialise()
        self.arrays = None
        self._order.append(self._n + len(self.na_rows))
        self._n += 1
        self._lists['phys'].append(phys)
        self._lists['gen_label'].append(gen_label)
        self._lists['test_gen'].append(float(test_gen))
        self._lists['lo'].append(int(lo))
        self._lists['hi'].append(int(hi))

    def add_na(self, line):
        self._materialise()
        self.na_rows[self._n + len(self.na_rows)] = line

    def add_many(self, phys, gen_label, test_gen, lo, hi):
        """Vectorised add() for modes whose test sites are a plain index stride (gen_label is
        test_gen in those modes)."""
        assert self._n == 0 and not self.na_rows
        self.arrays = (np.asarray(phys, dtype=np.int64), np.asarray(test_gen, dtype=np.float64),
                       np.asarray(lo, dtype=np.int64), np.asarray(hi, dtype=np.int64))
        self._n = len(self.arrays[0])

    def __len__(self):
        return self._n


def sites_alpha(data, s):
    """_alpha, v1:598-610: every int(s)-th site, window = all sites."""
    ts = TestSites()
    N = data.numSites
    idx = np.arange(0, N, int(s))
    g = data.genPos[idx]
    ts.add_many(data.position[idx], g, g, np.zeros(len(idx), np.int64), np.full(len(idx), N - 1, np.int64))
    return ts


def sites_site_based(data, r, s):
    """_siteBased, v1:580-594: r sites to the left, r+1 to the right (end inclusive)."""
    ts = TestSites()
    N = data.numSites
    if float(s) == int(s) and int(s) >= 1 and float(r) == int(r):      # the usual case, vectorised
        idx = np.arange(0, N, int(s))
        g = data.genPos[idx]
        ts.add_many(data.position[idx], g, g, np.maximum(0, idx - int(r)), np.minimum(N - 1, idx + int(r) + 1))
        return ts
    i = 0
    while i < N:
        start_i = max(0, i - r)
        end_i = min(N - 1, i + r + 1)
        w = np.arange(start_i, end_i + 1, dtype=int)      # as the reference builds it
        ts.add(data.position[int(i)], data.genPos[int(i)], data.genPos[int(i)], w[0], w[-1])
        i += s
    return ts


def sites_fix_center(data, w, s):
    """_fixSize_siteCenter, v1:549-577: w-nt window centred on every int(s)-th site.  The
    reference's two monotone pointers are `first index with position >= bound`, i.e. searchsorted."""
    ts = TestSites()
    N = data.numSites
    pos = np.asarray(data.position)
    if N and int(s) >= 1 and np.all(pos[1:] >= pos[:-1]):
        idx = np.arange(0, N, int(s))
        test = pos[idx]
        start = np.maximum(0, test - w / 2)
        end = np.minimum(test + w / 2, pos[-1])
        posf = pos.astype(np.float64)
        start_i = np.searchsorted(posf, start, 'left')
        end_i = np.minimum(np.searchsorted(posf, end, 'left'), N - 1)
        if np.any(end_i < start_i):
            j = int(np.nonzero(end_i < start_i)[0][0])
            print(start[j], start_i[j], end[j], end_i[j])
            sys.exit(1)
        g = data.genPos[idx]
        ts.add_many(test, g, g, start_i, end_i)
        return ts
    i = 0
    start_i = 0
    end_i = 0
    while i < N:
        testSite = pos[i]
        start = max(0, testSite - w / 2)
        end = min(testSite + w / 2, pos[-1])
        while pos[start_i] < start:
            start_i += 1
        while end_i < N:
            if pos[end_i] < end:
                end_i += 1
            else:
                break
        if end_i < start_i:
            print(start, start_i, end, end_i)
            sys.exit(1)
        end_i = min(end_i, N - 1)
        ts.add(testSite, data.genPos[i], data.genPos[i], start_i, end_i)
        i += int(s)
    return ts


def sites_fix_nocenter(data, w, s):
    """_fixSize_noCenter, v1:513-545: test positions every s nt, window one step wide."""
    ts = TestSites()
    N = data.numSites
    pos = data.position
    start = int(floor(2 * float(pos[0]) / w) * (w / 2))
    end = start + s
    midpos = start + s / 2
    start_i = 0
    end_i = 0
    pos_i = 0
    while midpos <= pos[-1]:
        while pos[start_i] < start:
            start_i += 1
        while pos[pos_i] < midpos:
            pos_i += 1
        while (end_i + 1) < N:
            if pos[end_i] < end:
                end_i += 1
            else:
                break
        gen_site = midpos * data.Rrate
        if start_i >= end_i:
            ts.add_na('%d\t%g\t0\tNA\tNA\tNA\t0\n' % (midpos, gen_site))
        else:
            ts.add(midpos, midpos * data.Rrate, gen_site, start_i, end_i)
        start += s
        midpos += s
        end += s
    return ts


def format_row(phys, gen_label, clr, ix, ia, iA, ns, sel):
    """The f-string of v1:540,574,591,607.  All-zero row when nothing beat Tmax = 0 (v1:451)."""
    if isinstance(gen_label, np.floating):
        gen_label = float(gen_label)
    if isinstance(phys, np.floating):
        phys = float(phys)
    if iA < 0:
        return f'{phys}\t{gen_label}\t0.0\t0.0\t0.0\t0.0\t0.0\n'
    return (f'{phys}\t{gen_label}\t{float(clr)}\t{sel.grid_x[ix]}\t{sel.grid_abeta[ia]}\t'
            f'{sel.grid_A[iA]}\t{int(ns)}\n')


def write_rows(outfile, ts, results, sel):
    """All rows of the output file.  Same text as format_row per row, built column-wise
    (a million-row file is otherwise dominated by per-row Python formatting)."""
    clr, ix, ia, iA, ns = results
    total = len(ts) + len(ts.na_rows)
    xs = [f'{v}' for v in sel.grid_x]
    abs_ = [f'{v}' for v in sel.grid_abeta]
    As = [f'{v}' for v in sel.grid_A]
    if not ts.na_rows and len(ts) and (ts.arrays is not None or
                                       all(isinstance(v, (int, np.integer)) for v in (ts.phys[0], ts.phys[-1]))):
        try:          # native writer: same bytes, ~10x faster on million-row files
            from . import _lib
            if ts.arrays is not None:
                phys, genl = ts.arrays[0], ts.arrays[1]
            else:
                phys, genl = np.asarray(ts.phys, dtype=np.int64), np.asarray(ts.gen_label, dtype=np.float64)
            with open(outfile, 'w') as scores:
                scores.write(HEADER)
            _lib.write_rows(outfile, phys, genl, clr, ix, ia, iA, ns, xs, abs_, As)
            return
        except (ImportError, OSError, AttributeError):
            pass
    phys = [float(v) if isinstance(v, np.floating) else v for v in ts.phys]
    gen = [float(v) if isinstance(v, np.floating) else v for v in ts.gen_label]
    body = [f'{p}\t{g}\t{c}\t{xs[a]}\t{abs_[b]}\t{As[d]}\t{n}\n' if d >= 0 else f'{p}\t{g}\t0.0\t0.0\t0.0\t0.0\t0.0\n'
            for p, g, c, a, b, d, n in zip(phys, gen, np.asarray(clr, dtype=np.float64).tolist(), np.asarray(ix).tolist(),
                                           np.asarray(ia).tolist(), np.asarray(iA).tolist(), np.asarray(ns).tolist())]
    if ts.na_rows:
        lines = [None] * total
        for pos, line in ts.na_rows.items():
            lines[pos] = line
        for j, pos in enumerate(ts.order):
            lines[pos] = body[j]
    else:
        lines = body
    with open(outfile, 'w') as scores:
        scores.write(HEADER)
        scores.writelines(lines)


class Scan:
    """v1:510-640: same constructor, same dispatch, same messages."""

    def __init__(self, InputData, NeutralSFS, NormalizedBetaBinom, Grids, outfile, fixSize=False, r=0, s=1,
                 phys=False, noCenter=False, runner=None, verbose=True, keep_results=True, reuse_ctx=None):
        # verbose=False: the ranks of a multi-GPU run that do not write the output stay silent; keep_results=False: the
        # output file is all the caller wants (self.results stays None when the rows were streamed to it);
        # reuse_ctx: a scan context of the same model to keep using (whole-genome runs, engine.NormalizedBetaBinom.bind)
        say = print if verbose else (lambda *a, **k: None)
        if fixSize:
            say('You\'ve chosen to fix the size (in nt) of sliding window for scanning.')
            if r == 0:
                say('Please set a window width in nt with "-w" or "--window" command.')
                sys.exit()
            if not phys:
                say(f'Please make sure to use physical positions as coordinates if fixed-length windows are chosen (--fixSize). Scan will continue with physical positions with a rec rate of {InputData.Rrate} cM/nt.')
                phys = True
            w = float(r)
            if noCenter:
                say(('Computing LR on %.3f kb windows on every %s nt. Using physical positions by default.' % (w / 1e3, s)))
                ts = sites_fix_nocenter(InputData, w, s)
            else:
                say(('Computing LR on %.3f kb windows on every %g informative sites. Using physical positions by default.' % (w / 1e3, s)))
                ts = sites_fix_center(InputData, w, s)
        elif r != 0:
            say(('Computing LR on every %s site/s, with a radius of %g informative sites on either side.' % (s, r)))
            ts = sites_site_based(InputData, r, s)
        else:
            say(('Computing LR on every %s site/s, using informative sites with exp(-A*dist) >= 1e-8.' % (s)))
            ts = sites_alpha(InputData, s)
        say(("writing output to %s" % (outfile)))
        NormalizedBetaBinom.bind(NeutralSFS, reuse=reuse_ctx)
        run = runner or engine.scan_batch
        streamed = False
        if len(ts):
            if runner is None and outfile is not None and ts.arrays is not None and not ts.na_rows:
                # one GPU, plain index-stride test sites (integer physPos): rows are written while the scan runs
                with open(outfile, 'w') as scores:
                    scores.write(HEADER)
                results = engine.scan_stream(NormalizedBetaBinom, ts.arrays[1], ts.arrays[2], ts.arrays[3], outfile,
                                             ts.arrays[0], ts.arrays[1], fetch=keep_results)
                streamed = True
            elif ts.arrays is not None:
                results = run(NormalizedBetaBinom, ts.arrays[1], ts.arrays[2], ts.arrays[3])
            else:
                results = run(NormalizedBetaBinom, ts.test_gen, ts.lo, ts.hi)
        else:
            results = (np.zeros(0), np.zeros(0, int), np.zeros(0, int), np.zeros(0, int), np.zeros(0, int))
        self.test_sites = ts
        if results is not None and hasattr(results, 'per_rank'):
            # a sharded run's records on the writing rank: rows straight from the per-rank arrays (native writer) when the
            # rows are plain (integer physPos, no NA rows); otherwise reassembled and written like any other result
            if outfile is not None and ts.arrays is not None and not ts.na_rows and len(ts):
                sel = NormalizedBetaBinom
                with open(outfile, 'w') as scores:
                    scores.write(HEADER)
                results.write(outfile, ts.arrays[0], ts.arrays[1], [f'{v}' for v in sel.grid_x], [f'{v}' for v in sel.grid_abeta],
                              [f'{v}' for v in sel.grid_A])
                streamed = True
                results = results.unpack() if keep_results else None
            else:
                results = results.unpack()
        self.results = results
        if outfile is not None and results is not None and not streamed:
            write_rows(outfile, ts, results, NormalizedBetaBinom)
        say(f'{datetime.now()}. Scan finished.')
