"""CPU ORACLE -- test infrastructure, NOT product code.

A numpy/scipy restatement of the hot path of the reference
(/root/reference/BalLeRMix+_v1.py, cited below as v1:LINE).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (ballermixplus_amd) never does and fails loudly
when its HIP library is missing.

Pinned: tests/test_oracle_golden.py checks every function here against
fixtures produced by running the reference itself in the build container
(tests/golden/make_golden.py) and against the reference's own example
outputs (tests/golden/ref_test/output/*.txt).

Two restatements of the scan are kept:
  * calc_baller_faithful  -- loop-for-loop what v1:436-507 does (per-A mask over all
    N sites, per-(x,a) gathers, sum(log(mix)) - sum(log(neut)), strict '>' argmax).
    This is the timed CPU baseline ("port").
  * clr_lut               -- the algebraically equal form the HIP kernel computes:
    T = 2*sum_i log1p(alpha_i * R[x,a][row_i]); used to check the restatement itself.
"""
import numpy as np
from scipy.stats import betabinom  # same third-party call the reference makes (v1:308,369)

STATS = ('B2', 'B2maf', 'B0', 'B0maf', 'B1')


def get_b(x, a):
    """v1:315-317"""
    return a / x - a


def raw_probs(stat, n, x, a):
    """Un-normalised, un-mirrored selection pmf for every count k=0..n (v1:375-396).
    For B1 the two entries are [P(substitution), P(polymorphism)] (v1:382)."""
    b = get_b(x, a)
    d = betabinom(n, a, b)
    if stat == 'B1':
        pn = d.pmf(n)
        return np.array([pn, 1. - pn - pn])
    k = np.arange(n + 1)
    if stat in ('B2', 'B0'):
        return d.pmf(k)
    probs = d.pmf(k) + d.pmf(n - k)          # v1:389
    if n % 2 == 0:
        probs = np.where(k == int(n / 2), probs / 2, probs)   # v1:391-392
    return probs


def excluded_counts(stat, n, min_count):
    """Counts removed from the support when normalising (v1:399-433)."""
    m = int(min_count)
    if stat in ('B2', 'B1'):
        return np.arange(m)                                            # v1:401, 417
    if stat == 'B2maf':
        return np.concatenate((np.arange(m), np.arange(n - m + 1, n)))      # v1:409
    if stat == 'B0':
        return np.concatenate((np.arange(m), np.array([n])))               # v1:421
    if stat == 'B0maf':
        return np.concatenate((np.arange(m), np.arange(n - m + 1, n + 1)))  # v1:429
    raise ValueError(stat)


def norm_base(n, x, a, excl):
    """v1:399-433: 1 - sum over excluded counts of the mirrored PLAIN beta-binomial pmf."""
    dx = betabinom(n, a, get_b(x, a))
    dm = betabinom(n, a, get_b(1. - x, a))
    ex = 0.5 * (dx.pmf(excl) + dm.pmf(excl))
    return 1. - np.sum(ex)


def sel_table(stat, n, min_count, xs, abetas):
    """P_sel[ix, ia, k]: NormalizedBetaBinom.normProbs evaluated on every count (v1:319-359).
    Last axis has n+1 entries (2 for B1: k=0 substitution, k=1 polymorphism)."""
    rows = 2 if stat == 'B1' else n + 1
    out = np.zeros((len(xs), len(abetas), rows))
    excl = excluded_counts(stat, n, min_count)
    for ix, x in enumerate(xs):
        for ia, a in enumerate(abetas):
            raw = 0.5 * (raw_probs(stat, n, x, a) + raw_probs(stat, n, 1. - x, a))   # v1:337-351
            out[ix, ia] = raw / norm_base(n, x, a, excl)
    return out


class Model:
    """Everything calcBaller reads, as plain arrays (v1:436: InputData, NeutralSFS,
    NormalizedBetaBinom, Grids).  Grids are given in the reference's ITERATION order,
    i.e. list(set(Grids.x)) etc. (v1:453,473,474)."""

    def __init__(self, stat, genpos, count, total, spect, samp_props, min_count, xs, abetas, As):
        self.stat = stat
        self.genpos = np.asarray(genpos, dtype=np.float64)
        self.count = np.asarray(count, dtype=np.int64)
        self.total = np.asarray(total, dtype=np.int64)
        self.xs, self.abetas, self.As = list(xs), list(abetas), list(As)
        self.N = len(self.genpos)
        # NeutralSFS.get_neut_probs v1:278-304
        self.probs = np.array([spect[(int(k), int(n))] for k, n in zip(self.count, self.total)])
        self.log_probs = np.log(self.probs)
        self.prop_sizes = np.array([samp_props[int(n)] for n in self.total], dtype=np.float64)
        # LUT rows: one block of rows per distinct n, in ascending n
        self.ns = sorted(set(int(n) for n in self.total))
        self.row_off = {}
        off = 0
        tabs = []
        for n in self.ns:
            self.row_off[n] = off
            t = sel_table(stat, n, min_count, self.xs, self.abetas)
            tabs.append(t)
            off += t.shape[2]
        self.psel = np.concatenate(tabs, axis=2)          # [nx, nab, rows]
        self.rows = off
        self.row = np.array([self.row_off[int(n)] + int(k) for k, n in zip(self.count, self.total)],
                            dtype=np.int64)
        g = np.ones(self.rows)
        prop = np.ones(self.rows)
        for n in self.ns:
            nk = 2 if stat == 'B1' else n + 1
            for k in range(nk):
                g[self.row_off[n] + k] = spect.get((k, n), np.nan)
                prop[self.row_off[n] + k] = samp_props[n]
        self.g_row, self.prop_row = g, prop
        with np.errstate(invalid='ignore', divide='ignore'):
            self.R = self.psel * prop / g - 1.0           # [nx, nab, rows]
        self._norm_probs = None

    def norm_probs(self):
        """Per-site arrays exactly as NormalizedBetaBinom.normProbs holds them (v1:359)."""
        if self._norm_probs is None:
            self._norm_probs = {(ix, ia): self.psel[ix, ia][self.row]
                                for ix in range(len(self.xs)) for ia in range(len(self.abetas))}
        return self._norm_probs


def calc_baller_faithful(m, window_lo, window_hi, test_site):
    """v1:436-507 restated loop for loop.  window = indices window_lo..window_hi inclusive.
    Returns (T, ix, ia, iA, nSites) with indices into m.xs/m.abetas/m.As, or
    (0.0, -1, -1, -1, 0) when no grid point has T > 0 (the reference's all-zero row)."""
    norm = m.norm_probs()
    dist = np.abs(m.genpos - test_site)                                  # v1:446
    best = (0.0, -1, -1, -1, 0)
    idx_all = np.arange(m.N)
    in_window = (idx_all >= window_lo) & (idx_all <= window_hi)
    for iA, A in enumerate(m.As):                                        # v1:453
        alphas = np.exp(-A * dist)                                       # v1:454
        sub = np.where((alphas >= 1e-8) & (m.genpos != test_site) & in_window)[0]   # v1:455-457
        if len(sub) == 0:
            continue
        sa = alphas[sub]
        prop = m.prop_sizes[sub]
        for ix in range(len(m.xs)):                                      # v1:473
            for ia in range(len(m.abetas)):                              # v1:474
                neut = m.probs[sub]
                sel = norm[(ix, ia)][sub] * prop                         # v1:479,492
                mix = sa * sel + (1. - sa) * neut                        # v1:494
                with np.errstate(divide='ignore', invalid='ignore'):
                    T = 2 * (np.sum(np.log(mix)) - np.sum(m.log_probs[sub]))   # v1:496-499
                if T > best[0]:                                          # v1:501
                    best = (float(T), ix, ia, iA, len(sub))
    return best


def window_mask(m, A, window_lo, window_hi, test_site):
    dist = np.abs(m.genpos - test_site)
    alphas = np.exp(-A * dist)
    idx = np.arange(m.N)
    keep = (alphas >= 1e-8) & (m.genpos != test_site) & (idx >= window_lo) & (idx <= window_hi)
    return np.where(keep)[0], alphas


def clr_lut(m, window_lo, window_hi, test_site, surface=False):
    """Same statistic through the (k,n)-indexed table: T = 2*sum log1p(alpha_i * R[row_i]).
    With surface=True also returns T[iA, ix, ia] (NaN where the window is empty) and nsites[iA]."""
    nA, nx, nab = len(m.As), len(m.xs), len(m.abetas)
    Ts = np.full((nA, nx, nab), np.nan)
    ns = np.zeros(nA, dtype=np.int64)
    for iA, A in enumerate(m.As):
        sub, alphas = window_mask(m, A, window_lo, window_hi, test_site)
        ns[iA] = len(sub)
        if len(sub) == 0:
            continue
        a = alphas[sub]
        Rw = m.R[:, :, m.row[sub]]                       # [nx, nab, W]
        with np.errstate(divide='ignore', invalid='ignore'):
            Ts[iA] = 2.0 * np.sum(np.log1p(a * Rw), axis=2)
    best = (0.0, -1, -1, -1, 0)
    flat = Ts.reshape(-1)
    for lin in range(flat.size):                          # reference order, strict '>'
        T = flat[lin]
        if T > best[0]:
            iA, rem = divmod(lin, nx * nab)
            ix, ia = divmod(rem, nab)
            best = (float(T), ix, ia, iA, int(ns[iA]))
    if surface:
        return best, Ts, ns
    return best


def alpha_cut_z():
    """Largest double z with numpy's exp(-z) >= 1e-8: the reference's window predicate
    `np.exp(-A*dist) >= 1e-8` (v1:454-455) is then exactly `A*dist <= z`."""
    lo, hi = 18.0, 19.0
    assert np.exp(-np.float64(lo)) >= 1e-8 > np.exp(-np.float64(hi))
    while True:
        mid = 0.5 * (lo + hi)
        if mid == lo or mid == hi:
            break
        if np.exp(-np.array([mid]))[0] >= 1e-8:
            lo = mid
        else:
            hi = mid
    return lo
