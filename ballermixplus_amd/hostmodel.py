"""Host-side mirror of the reference's input / neutral-model / grid classes.

These keep the reference's names, argument meaning, messages and exit behaviour
(reference BalLeRMix+_v1.py, cited as v1:LINE) because the CLI built on them must
be a drop-in: same flags, same files in, same files out.  None of this is on the
hot path -- it only produces the arrays the HIP kernels consume.
"""
import sys

import numpy as np


def read_columns(infile, pos_col):
    """The four columns of an input file as arrays: phys = int(float(col0)), coord = float(col[pos_col]), k = int(col2),
    n = int(col3) -- the conversions the reference applies per line (v1:103-104, 121-124).  libbmxscan's mmap/strtod reader
    does this for plain 4-column text (bit-identical, threaded); a file it declines (hex floats, inf, padded fields, more
    columns: anything Python's float()/int() may still accept or must reject with its own error) goes through Python's
    converters, one column at a time."""
    try:
        from . import _lib
        return _lib.read_input(infile, pos_col)
    except Exception:
        pass
    with open(infile, 'r') as fh:
        fh.readline()                                                     # header
        fields = [ln.strip().split('\t') for ln in fh]
    col = lambda j, conv, dt: np.array([conv(f[j]) for f in fields], dtype=dt)
    return (col(0, lambda v: int(float(v)), np.int64), col(pos_col, float, np.float64), col(2, int, np.int64), col(3, int, np.int64))


class InputData:
    """v1:8-131.  Arrays: position (int), genPos (f64), count (k), total (n)."""

    def __init__(self, infile, nofreq=False, MAF=False, nosub=False, minCount=1, phys=False, Rrate=1e-6):
        self.minCount = minCount
        self.Rrate = Rrate
        pos_col = 1 - int(phys)      # column holding the coordinate: 0 physical, 1 genetic (v1:18)
        self.position, coord, count, self.total = read_columns(infile, pos_col)
        # float(col)*(1-pos_type)*Rrate + float(col)*pos_type  (v1:103,124)
        self.genPos = coord * (1 - pos_col) * Rrate + coord * pos_col
        if nofreq:                   # v1:91-100: from the first count that is neither 0 nor 1 on, k becomes (k != n)
            odd = np.flatnonzero((count != 0) & (count != 1))
            if len(odd):
                print('Input includes different variant counts despite choosing not to use allele frequencies (with --noFreq). All sites with counts smaller than substitutions will be considered as polymorphic. All sites with identical counts as sample sizes will be substitutions.')
                count = count.copy()
                count[odd[0]:] = count[odd[0]:] != self.total[odd[0]:]
        self.count = count
        self.numSites = len(count)
        if not nofreq:
            _stat = '%s%s' % (['B_2', 'B_0'][nosub], ['', 'MAF'][MAF])
            if nosub:                                                     # v1:41-50
                if np.sum(self.count == self.total) > 0:
                    print(f'You have chosen to compute {_stat}. Substitutions (x==n) in the input will be ignored.')
                    keep = np.where(self.count != self.total)
                    self.count = self.count[keep]
                    self.total = self.total[keep]
                    self.genPos = self.genPos[keep]
                    self.position = self.position[keep]
                    self.numSites = len(self.count)
            if MAF:                                                       # v1:53-59
                if np.sum(self.count > self.total / 2) > 0:
                    print(f'Input data includes non-MAF site/s (frequency >= 0.5) despite choosing to use {_stat} (with --MAF). These frequencies will be folded for following analyses.')
                    self.count = np.where(self.count > self.total / 2, (self.total - self.count), self.count)
                self.minCount = int(self.count[self.count > 0].min())
            else:                                                         # v1:60-74
                if np.sum(self.count == 0) != 0:
                    print('Please make sure to include derived allele frequency only. Sites with zero derived alleles (x==0) should not be included in your input.')
                    sys.exit()
                self.minCount = int(self.count.min())
        else:
            self.minCount = int(minCount)
        self.sampSizes = set(self.total.tolist())                         # v1:76


class Grids:
    """v1:134-175.  Lists keep the reference's Python objects (ints stay ints, 1e3 stays a
    float) because the output prints them with repr (v1:607)."""

    # the reference's grids, value for value (v1:142, 152-164): part of the contract
    TAIL_ABETA = [300, 500, 1e3, 1e4, 1e6, 1e9]
    WHOLE_ABETA = list(range(1, 10)) + list(range(5, 100, 5)) + list(range(100, 210, 10)) + TAIL_ABETA      # --findBal
    SMALL_ABETA = [0.001, 0.01, 0.05, 0.1, 0.2, 0.5, 0.8]                                                   # --findPos
    DEFAULT_ABETA = SMALL_ABETA + WHOLE_ABETA
    DEFAULT_A = (list(range(100, 1200, 100)) + list(range(1200, 2600, 200)) + list(range(2500, 5000, 500)) +
                 list(range(5000, 11000, 1000)) + [1e6, 1e8])

    def __init__(self, x, abeta, bal, pos, seqA, listA):
        if abeta is None and not bal and pos:        # --findPos alone: x in steps of 0.1, whatever --fixX says (v1:153-155)
            self.x = [.1 * i for i in range(1, 11)]
        else:
            self.x = [float(x)] if x is not None else [.05 * i for i in range(1, 11)]
        self.abeta = list(self.WHOLE_ABETA if bal else self.SMALL_ABETA if pos else self.DEFAULT_ABETA)
        if abeta is not None:
            try:
                self.abeta = [float(abeta)]
            except Exception:
                print(f'The value for "a" provided ({abeta}) is not legitimate. Using the default grid instead.')
                self.abeta = list(self.DEFAULT_ABETA)
        if listA:
            self.A = [float(v) for v in listA.split(',')]
        elif seqA:
            # The reference raises here (float range + 'Atep' typo, v1:169-171; SURVEY 5 defect 1);
            # this is the evident intent: Amin, Amin+Astep, ... up to Amax, as floats.
            first, last, step = (float(v) for v in seqA.split(','))
            self.A = [first + step * i for i in range(int(round((last - first) / step)) + 1)]
        else:
            self.A = list(self.DEFAULT_A)

    def scan_order(self):
        """Iteration order of the grid search: calcBaller loops `for A in set(Grids.A)`,
        `for x in set(Grids.x)`, `for abeta in set(Grids.abeta)` (v1:453,473,474).  CPython's set
        order for ints/floats is deterministic, so list(set(...)) here is that same order and
        the strict '>' argmax (v1:501) resolves ties identically."""
        return list(set(self.x)), list(set(self.abeta)), list(set(self.A))


class NeutralSFS:
    """v1:180-304: neutral spectrum / configuration helper file."""

    def __init__(self, spectfile, nofreq, MAF, nosub):
        self.spect = {}
        self.sampSizes = set()
        self.probs = []
        self.logProbs = []
        self.sampProps = {}
        self.propSizes = []
        if nofreq:
            self.readConfig(spectfile)
        else:
            self.readSpect(spectfile, MAF, nosub)

    @staticmethod
    def _table(path, kinds):
        """The tab-separated columns of a helper file as arrays (one converter per column)."""
        with open(path, 'r') as fh:
            rows = [ln.strip().split('\t') for ln in fh]
        return [np.array([conv(r[j]) for r in rows], dtype=dt) for j, (conv, dt) in enumerate(kinds)]

    @staticmethod
    def _running(keys, vals):
        """{key: sum of vals} with every sum taken in file order (the reference adds line by line, v1:212-214)."""
        uniq, inv = np.unique(keys, return_inverse=True)
        tot = np.zeros(len(uniq))
        np.add.at(tot, inv, vals)                 # unbuffered: element by element, in order
        return uniq, tot

    def readSpect(self, spectfile, MAF, nosub):                           # v1:183-223
        x, n, f = self._table(spectfile, [(int, np.int64), (int, np.int64), (float, np.float64)])
        line = np.arange(len(x))
        fold = np.zeros(len(x), dtype=bool)
        if MAF:
            fold = ~(x < (n / 2 + 1))                                     # polarised entries of a spectrum used folded
            stop = line[(x == 0)] if nosub else line[:0]
            for _ in range(int(fold[:stop[0]].sum()) if len(stop) else int(fold.sum())):
                print('You have indicated to use minor allele frequencies (--MAF) but provided SFS based on polarized allele frequency. This SFS will be folded accordingly.')
            if len(stop):
                print('You have chosen to compute B_0maf. Please do not account for sites with zero counts (x==0) in your input.')
                sys.exit()
        elif nosub and np.any(x == n):
            print('You have chosen to compute B_2maf. Please do not account for substitutions (derived allele count x == n) in your input.')
            sys.exit()
        # an unfolded line SETS its entry, a folded one ADDS to the entry of n - x (v1:195-208): per entry, the value of its last
        # plain line (if any) plus the folded lines that follow it, added in file order
        kx = np.where(fold, n - x, x)
        base = int(n.max()) + 1 if len(n) else 1
        uniq, inv = np.unique(kx * base + n, return_inverse=True)
        last_set = np.full(len(uniq), -1, dtype=np.int64)
        np.maximum.at(last_set, inv[~fold], line[~fold])
        val = np.where(last_set >= 0, f[np.maximum(last_set, 0)], 0.0)
        late = fold & (line > last_set[inv])
        np.add.at(val, inv[late], f[late])
        self.spect = {(int(u) // base, int(u) % base): float(v) for u, v in zip(uniq, val)}
        sizes, props = self._running(n, f)
        self.sampProps = {int(a): float(b) for a, b in zip(sizes, props)}
        checksum = float(np.cumsum(f)[-1]) if len(f) else 0.                # a running sum, like the reference's
        if not np.isclose(checksum, 1.):
            print(f'Fraction of sites do not add up to 1! Sum = {checksum}. Please double-check your inputs.')
            sys.exit()
        self.sampSizes = set(n.tolist())

    def readConfig(self, spectfile):                                      # v1:227-250
        n, sub, poly = self._table(spectfile, [(int, np.int64), (float, np.float64), (float, np.float64)])
        for a, b in zip(sub.tolist(), poly.tolist()):
            print(('Substitutions: %s ; polymorphisms: %s' % (a, b)))
        both = sub + poly
        sizes, props = self._running(n, both)
        self.sampProps = {int(a): float(b) for a, b in zip(sizes, props)}
        checksum = float(np.cumsum(both)[-1]) if len(both) else 0.
        if not checksum == 1.:
            print(f'Fraction of sites do not add up to 1! Sum = {checksum}. Please double-check your inputs.')
            sys.exit()
        # as in the reference, the spectrum holds the LAST line's two entries only (v1:241)
        self.spect = {(0, int(n[-1])): float(sub[-1]), (1, int(n[-1])): float(poly[-1])} if len(n) else {}
        self.sampSizes = set(n.tolist())

    def get_neut_probs(self, data):                                       # v1:278-304
        """Checks that every (k, n) in the input is covered by the helper file.  The per-site
        arrays of the reference are replaced by the (k,n)-indexed table handed to the kernels
        (SelectionTable); probs/logProbs/propSizes are still filled for API compatibility."""
        # the distinct (k, n) of the input, through one integer key per site (a Python set of 40M tuples is what a
        # whole-genome run would otherwise spend its host time on)
        base = int(data.total.max()) + 1 if data.numSites else 1
        key = np.asarray(data.count).astype(np.int64) * base + np.asarray(data.total).astype(np.int64)
        uk, inv = np.unique(key, return_inverse=True)
        combos = [(int(u) // base, int(u) % base) for u in uk]
        if any(c not in self.spect for c in combos):
            print('Input data includes sample counts and sizes not included in the helper file. Please double-check your inputs.')
            sys.exit()
        if len(self.probs) == 0:
            vals = np.array([self.spect[c] for c in combos], dtype=np.float64)
            self.probs = vals[inv] if len(vals) else np.zeros(0)
        assert len(self.probs) == data.numSites
        if len(self.logProbs) == 0:
            self.logProbs = np.log(self.probs)
        if len(self.propSizes) == 0:
            un, inv = np.unique(data.total, return_inverse=True)
            self.propSizes = np.array([self.sampProps[int(n)] for n in un], dtype=np.float64)[inv]
