"""Python face of the MI355X scan: the two seams of the reference's hot path.

    NormalizedBetaBinom(InputData, Grids, nofreq, MAF, nosub)       reference BalLeRMix+_v1.py:310-433, called at :793
    calcBaller(window_indice, testSite, InputData, NeutralSFS,
               NormalizedBetaBinom, Grids)                          reference BalLeRMix+_v1.py:436-507

keep their names, argument meaning and return values, but run on the GPU through
libbmxscan.so (include/bmxscan.h).  `scan_batch` is the call the CLI uses: all test
sites of a chromosome in one launch.  Nothing here computes likelihoods on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BmxModel

STAT_IDS = {'B2': 0, 'B2maf': 1, 'B0': 2, 'B0maf': 3, 'B1': 4}


def stat_name(nofreq, MAF, nosub):
    """The dispatch of v1:336-352."""
    if nofreq:
        return 'B1'
    if MAF:
        return 'B0maf' if nosub else 'B2maf'
    return 'B0' if nosub else 'B2'


class ModelArrays:
    """The bmx_model struct plus the numpy arrays that back its pointers."""

    def __init__(self, stat, min_count, sizes, spect, samp_props, xs, abetas):
        self.stat = stat
        self.sizes = _lib.i32(sorted(int(n) for n in sizes))
        per = [2 if stat == 'B1' else int(n) + 1 for n in self.sizes]
        self.row_off = _lib.i32(np.concatenate(([0], np.cumsum(per))))
        self.rows = int(self.row_off[-1])
        g = np.full(self.rows, np.nan)
        for j, n in enumerate(self.sizes.tolist()):
            for k in range(per[j]):
                v = spect.get((k, n))
                if v is not None:
                    g[self.row_off[j] + k] = v
        self.g = _lib.f64(g)
        self.prop = _lib.f64([samp_props[int(n)] for n in self.sizes])
        self.x = _lib.f64(xs)
        self.abeta = _lib.f64(abetas)
        self.min_count = int(min_count)
        self._off_of = {int(n): int(o) for n, o in zip(self.sizes, self.row_off[:-1])}
        # two models with equal keys build bit-identical device tables
        self.key = (stat, self.min_count, self.sizes.tobytes(), self.g.tobytes(), self.prop.tobytes(), self.x.tobytes(),
                    self.abeta.tobytes())
        self.c = BmxModel(STAT_IDS[stat], self.min_count, len(self.sizes), _lib.as_ip(self.sizes),
                          _lib.as_ip(self.row_off), _lib.as_dp(self.g), _lib.as_dp(self.prop),
                          len(self.x), _lib.as_dp(self.x), len(self.abeta), _lib.as_dp(self.abeta))

    def rows_of(self, count, total):
        """LUT row of every site: row_off[n] + k."""
        total = np.asarray(total)
        count = np.asarray(count)
        if total.size == 0:
            return _lib.i32(np.zeros(0, dtype=np.int64))
        lo, hi = int(total.min()), int(total.max())
        if lo < 0:
            raise KeyError(lo)
        tab = np.full(hi + 1, -1, dtype=np.int32)         # sample size -> first row; one gather instead of a sort of all sites
        for n, off in self._off_of.items():
            if 0 <= n <= hi:
                tab[n] = off
        rows = tab[total]
        if rows.min() < 0:
            raise KeyError(int(total[int(np.argmin(rows))]))
        rows += count.astype(np.int32, copy=False)
        return _lib.i32(rows)


class Context:
    """One resident scan context (one GPU).  Thin wrapper over the bmx_ctx_* calls."""

    def __init__(self, device=0):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        _lib.check(self._L.bmx_ctx_create(C.byref(self._h), int(device)))
        self.device = int(device)
        self.M = 0
        self.slot = 0
        self._slot_M = {}
        self._keep = []

    def close(self):
        if self._h:
            self._L.bmx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_model(self, model, As):
        A = _lib.f64(As)
        self.model, self.nA = model, len(A)
        self._slot_M = {}
        self.M = 0
        _lib.check(self._L.bmx_ctx_set_model(self._h, C.byref(model.c), _lib.as_dp(A), len(A)))

    def set_sites(self, genpos, rows):
        g, r = _lib.f64(genpos), _lib.i32(rows)
        self.N = len(g)
        self.M = 0                       # the library drops the slot's test sites (they were located in the old array)
        self._slot_M.pop(self.slot, None)
        _lib.check(self._L.bmx_ctx_set_sites(self._h, len(g), _lib.as_dp(g), _lib.as_ip(r)))

    def set_tests(self, test_gen, win_lo=None, win_hi=None):
        """Test positions and their inclusive index windows; no windows = all sites of the chromosome (the reference's default mode)."""
        t = _lib.f64(test_gen)
        self.M = len(t)
        self._slot_M[self.slot] = self.M
        if win_lo is None and win_hi is None:
            _lib.check(self._L.bmx_ctx_set_tests(self._h, len(t), _lib.as_dp(t), None, None))
            return
        lo, hi = _lib.i64(win_lo), _lib.i64(win_hi)
        _lib.check(self._L.bmx_ctx_set_tests(self._h, len(t), _lib.as_dp(t), _lib.as_lp(lo), _lib.as_lp(hi)))

    def set_variant(self, v):
        _lib.check(self._L.bmx_ctx_set_variant(self._h, int(v)))

    def select_slot(self, k):
        """Chromosome slot k of this context (created on first use): its own site arrays, test sites and results,
        the context's one model.  set_sites / set_tests / scan / fetch* act on the selected slot."""
        _lib.check(self._L.bmx_ctx_select_slot(self._h, int(k)))
        self.slot = int(k)
        self.M = self._slot_M.get(self.slot, 0)

    def plan(self):
        """{'J', 'use_lds', 'mode', 'stream_bytes', 'kernel'} of the selected slot's scan (bmx_ctx_plan)."""
        J, ul, mode = C.c_int32(), C.c_int32(), C.c_int32()
        sb = C.c_int64()
        _lib.check(self._L.bmx_ctx_plan(self._h, C.byref(J), C.byref(ul), C.byref(mode), C.byref(sb)))
        lds = 'true' if ul.value else 'false'
        if mode.value == 5:
            name = 'clr_scan_solo_kernel<%s>' % lds
        elif mode.value == 4:
            name = 'clr_scan_prepared_kernel<%d,%s>' % (J.value, lds)
        elif mode.value >= 0:
            name = 'clr_scan_grouped_kernel<%d,%s,%d>' % (J.value, lds, mode.value)
        else:
            name = 'clr_scan_kernel<%s>' % lds
        return {'J': J.value, 'use_lds': bool(ul.value), 'mode': mode.value, 'stream_bytes': sb.value, 'kernel': name}

    def launch_ranges(self):
        """First test-site index of every launch range of the selected slot's scan (bmx_ctx_launch_ranges)."""
        n = C.c_int32()
        _lib.check(self._L.bmx_ctx_launch_ranges(self._h, None, 0, C.byref(n)))
        offs = np.zeros(max(n.value, 1), dtype=np.int64)
        _lib.check(self._L.bmx_ctx_launch_ranges(self._h, _lib.as_lp(offs), len(offs), C.byref(n)))
        return offs[:n.value]

    def pack_records(self, out=None, device_ptr=None, cap=None):
        """Records of every slot with results, slot order.  Host: returns a RECORD array.  device_ptr/cap: packs into
        that device buffer (room for cap records) and returns the count."""
        n = C.c_int64()
        if device_ptr is not None:
            _lib.check(self._L.bmx_ctx_pack_records(self._h, C.c_void_p(int(device_ptr)), int(cap), 1, C.byref(n)))
            return n.value
        total = sum(self._slot_M.values())
        rec = out if out is not None else np.empty(total, dtype=_lib.RECORD_DTYPE)
        _lib.check(self._L.bmx_ctx_pack_records(self._h, rec.ctypes.data_as(C.c_void_p), len(rec), 0, C.byref(n)))
        return rec[:n.value]

    def copy_records(self, device_ptr, cap):
        """The SELECTED slot's records into a device buffer of this GPU (room for cap records); waits for the scan."""
        if cap < self.M:
            raise ValueError('destination holds %d records, the slot has %d' % (cap, self.M))
        _lib.check(self._L.bmx_ctx_copy_records(self._h, C.c_void_p(int(device_ptr)), int(cap)))
        return self.M

    def scan(self):
        _lib.check(self._L.bmx_ctx_scan(self._h))

    def sync(self):
        _lib.check(self._L.bmx_ctx_sync(self._h))

    def last_scan_ms(self):
        ms = C.c_double()
        _lib.check(self._L.bmx_ctx_last_scan_ms(self._h, C.byref(ms)))
        return ms.value

    def fetch(self):
        M = self.M
        clr = np.empty(M, dtype=np.float64)
        ix, ia, iA, ns = (np.empty(M, dtype=np.int32) for _ in range(4))
        _lib.check(self._L.bmx_ctx_fetch(self._h, _lib.as_dp(clr), _lib.as_ip(ix), _lib.as_ip(ia),
                                         _lib.as_ip(iA), _lib.as_ip(ns)))
        return clr, ix, ia, iA, ns

    def result_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self._L.bmx_ctx_result_ptrs(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def records(self):
        """Device address of the M 16-byte result records (bmx_record) of the last scan."""
        a = C.c_void_p()
        _lib.check(self._L.bmx_ctx_records(self._h, C.byref(a)))
        return a.value

    def fetch_records(self):
        """The last scan's results as a structured array (clr f8, lin i4, nsites i4)."""
        rec = np.empty(self.M, dtype=_lib.RECORD_DTYPE)
        _lib.check(self._L.bmx_ctx_fetch_records(self._h, rec.ctypes.data_as(C.c_void_p)))
        return rec

    def scan_write(self, path, phys, gen_label, xs, abs_, As, chunk=0):
        """Scan the test sites set by set_tests and append the reference's rows to `path` while scanning
        (chunks of test sites; a writer thread formats chunk i while chunk i+1 is on the device)."""
        phys, gen_label = _lib.i64(phys), _lib.f64(gen_label)
        if len(phys) != self.M or len(gen_label) != self.M:
            raise ValueError('one (physPos, genPos) label pair per test site is needed')
        pack = lambda v: b'\0'.join(s.encode() for s in v) + b'\0'
        _lib.check(self._L.bmx_ctx_scan_write(self._h, path.encode(), _lib.as_lp(phys), _lib.as_dp(gen_label), pack(xs), len(xs),
                                              pack(abs_), len(abs_), pack(As), len(As), int(chunk)))

    def surface(self, test_gen, win_lo, win_hi):
        """T[nA, nx, nab] (NaN where the window is empty) and nsites[nA] of one test site."""
        m = self.model
        T = np.empty((self.nA, len(m.x), len(m.abeta)), dtype=np.float64)
        ns = np.empty(self.nA, dtype=np.int32)
        _lib.check(self._L.bmx_ctx_surface(self._h, float(test_gen), int(win_lo), int(win_hi), _lib.as_dp(T),
                                           _lib.as_ip(ns)))
        return T, ns

    def fetch_lut(self):
        m = self.model
        shape = (len(m.x), len(m.abeta), m.rows)
        psel, R = np.empty(shape), np.empty(shape)
        _lib.check(self._L.bmx_ctx_fetch_lut(self._h, _lib.as_dp(psel), _lib.as_dp(R)))
        return psel, R


class Comm:
    """The library's own RCCL communicator (bmx_comm_*): one per (context, process group).  `make_id()` on rank 0, the 128 bytes
    to the other ranks by any channel, then Comm(ctx, id, rank, world) on every rank (collective)."""

    @staticmethod
    def make_id():
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().bmx_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, comm_id, rank, world):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        _lib.check(self._L.bmx_comm_create(C.byref(self._h), ctx._h, comm_id, self.rank, self.world))

    def gather_records(self, counts, root=0, out=None):
        """ONE gather of the context's packed records to `root` (ncclSend / ncclRecv inside the library, device to device).
        Root: a RECORD array of sum(counts) records, rank after rank (`out` if given); other ranks: None."""
        cnt = _lib.i64(counts)
        if len(cnt) != self.world:
            raise ValueError('one count per rank is needed')
        rec = None
        if self.rank == root:
            rec = out if out is not None else np.empty(int(cnt.sum()), dtype=_lib.RECORD_DTYPE)
        ptr = rec.ctypes.data_as(C.c_void_p) if rec is not None else None
        _lib.check(self._L.bmx_comm_gather_records(self._h, _lib.as_lp(cnt), int(root), ptr, None))
        return rec

    def close(self):
        if self._h:
            self._L.bmx_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NormalizedBetaBinom:
    """Drop-in for the reference class of the same name (v1:310-433): same constructor
    arguments; `get(x, a)` returns the per-site normalised selection probabilities.  The table
    is built by the device kernel (K1) and stays resident for calcBaller / scan_batch.
    The device table also folds in the neutral spectrum (R = P_sel*prop/g - 1), which the
    reference only supplies at calcBaller time, so the device state is created on first use
    (`bind`), when the NeutralSFS is known."""

    def __init__(self, InputData, Grids, nofreq, MAF, nosub, device=0, ctx=None):
        # ctx: an empty Context created ahead of time (the CLI brings the HIP runtime up while it parses the input)
        self._fresh_ctx = ctx
        self.stat = stat_name(nofreq, MAF, nosub)
        self.grid_x, self.grid_abeta, self.grid_A = Grids.scan_order()
        self._data = InputData
        self.site_gen = InputData.genPos          # (work-balanced sharding estimates window sizes from the site positions)
        self._device = device
        self._bound_to = None
        self.ctx = None
        self._psel = None
        self._key = {}
        for i, x in enumerate(self.grid_x):
            for j, a in enumerate(self.grid_abeta):
                self._key[(x, a)] = (i, j)

    def bind(self, NeutralSFS, reuse=None):
        """Create (once per NeutralSFS) the resident context: K1 table + site arrays in HBM.
        reuse: a Context that already holds this model and A grid (e.g. the previous chromosome's, whole-genome runs):
        its table and buffers are kept and only the site arrays are replaced."""
        if self._bound_to is NeutralSFS and self.ctx is not None:
            return self
        self.prepare(NeutralSFS)
        d = self._data
        if self.ctx is not None and self.ctx is not reuse:
            self.ctx.close()
        akey = tuple(float(a) for a in self.grid_A)
        same_dev = reuse is not None and reuse.device == self._device
        if same_dev and getattr(reuse, 'model', None) is not None and reuse.model.key == self.model.key and \
                getattr(reuse, 'A_key', None) == akey:
            self.ctx = reuse                      # same table: kept, only the site arrays are replaced
            self.ctx.model = self.model
            self.table_reused = True
        else:
            if same_dev:
                self.ctx = reuse                  # same GPU, another model (a chromosome with other sample sizes, another
                                                  # minCount, ...): the context and its buffers are kept, the table is rebuilt
            elif self._fresh_ctx is not None and self._fresh_ctx.device == self._device:
                self.ctx, self._fresh_ctx = self._fresh_ctx, None
            else:
                self.ctx = Context(self._device)
            self.ctx.set_model(self.model, self.grid_A)
            self.ctx.A_key = akey
            self.table_reused = False
        self.ctx.set_sites(d.genPos, self.rows)
        self._bound_to = NeutralSFS
        self._psel = None
        return self

    def prepare(self, NeutralSFS):
        """The host half of bind(): the bmx_model arrays and every site's table row.  No device call, so a whole-genome run
        does it for the next chromosome on a helper thread while the current one is scanned (cli.main_many)."""
        if getattr(self, '_prepared_for', self) is NeutralSFS and getattr(self, 'model', None) is not None:
            return self
        d = self._data
        if NeutralSFS is None:      # get() before any scan: neutral part is irrelevant to P_sel
            spect = {(k, int(n)): 1.0 for n in d.sampSizes for k in range(int(n) + 1)}
            props = {int(n): 1.0 for n in d.sampSizes}
        else:
            spect, props = NeutralSFS.spect, NeutralSFS.sampProps
        self.model = ModelArrays(self.stat, d.minCount, d.sampSizes, spect, props, self.grid_x, self.grid_abeta)
        self.rows = self.model.rows_of(d.count, d.total)
        self._prepared_for = NeutralSFS
        return self

    def get(self, x, a):
        """v1:362-363"""
        if self.ctx is None:
            self.bind(None)
        if self._psel is None:
            self._psel, _ = self.ctx.fetch_lut()
        i, j = self._key[(x, a)]
        return self._psel[i, j][self.rows]


def scan_stream(sel, test_gen, win_lo, win_hi, outfile, phys, gen_label, fetch=True):
    """scan_batch + the output rows appended to `outfile` while the scan runs (the reference writes as it scans,
    v1:599-608).  Returns what scan_batch returns (None with fetch=False: the file is all the caller wants)."""
    sel.ctx.set_tests(test_gen, win_lo, win_hi)
    # chunks of 64k test sites keep the first rows early on small inputs; large inputs take 256k per chunk (fewer kernel tails)
    sel.ctx.scan_write(outfile, phys, gen_label, [f'{v}' for v in sel.grid_x], [f'{v}' for v in sel.grid_abeta],
                       [f'{v}' for v in sel.grid_A], chunk=262144 if len(phys) >= (1 << 20) else 0)
    return sel.ctx.fetch() if fetch else None


def scan_batch(sel, test_gen, win_lo, win_hi):
    """All test sites at once.  Returns (clr f64[M], ix, ia, iA, nsites) with indices into the
    iteration-order grids held by `sel` (iA == -1: the reference's all-zero row)."""
    sel.ctx.set_tests(test_gen, win_lo, win_hi)
    sel.ctx.scan()
    return sel.ctx.fetch()


def calcBaller(window_indice, testSite, InputData, NeutralSFS, NormalizedBetaBinom, Grids):
    """Drop-in for v1:436-507 (one test site).  `window_indice` must be a contiguous index
    range, which is all the reference ever passes (v1:538,572,589,606).  Returns
    [T, x, abeta, A, nSites] with the grid's own Python objects, or the all-zero list."""
    w = np.asarray(window_indice)
    NormalizedBetaBinom.bind(NeutralSFS)
    clr, ix, ia, iA, ns = scan_batch(NormalizedBetaBinom, [float(testSite)], [int(w[0])], [int(w[-1])])
    if iA[0] < 0:
        return [0., 0., 0., 0., 0.]
    s = NormalizedBetaBinom
    return [float(clr[0]), s.grid_x[ix[0]], s.grid_abeta[ia[0]], s.grid_A[iA[0]], int(ns[0])]
