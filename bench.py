#!/usr/bin/env python3
"""bench.py -- CLR windows/s of the B_2 scan (BASELINE.json metric) on N MI355X.

Default workload = BASELINE config 4, the north_star target: a synthetic whole genome, 40M SNPs over 22
chromosomes (GRCh37 proportions), n = 100, default 31x10x51 (A, x, alpha) grid, every SNP a test site.
One step = one pass of the hot path over the whole genome.  ONE scan context per GPU holds the selection table and
every chromosome (one slot each: site arrays + this rank's test sites) resident in HBM before the timed region; a
step launches the 22 scans back to back on the context's stream, waits once, and moves the 16-byte result records of
all test sites with ONE transfer: device -> pinned host memory (N = 1) or ONE gather to rank 0 (N > 1: RCCL over xGMI,
north_star's "only a final RCCL gather"), so the result transfer is INSIDE the timed step.  With N ranks every rank
holds all site arrays (400 MB) and scans every N-th block of 4096 test sites of each chromosome: strong scaling, no
data-path collective, results bitwise independent of N.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 4|3|5] [--total-snps T] [--no-cpu-baseline]

`--gpus N` with N > 1 needs no launcher: a parent that never touches the GPU starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`
as a child and relays its one JSON line; under an existing torch.distributed.run (WORLD_SIZE set) the ranks run as they are.

--config 3: one 1M-SNP chromosome per GPU (weak scaling; the single launch the PMC profiles in profiles/ are taken on);
--config 5: BASELINE config 5, the dense-grid stress: 10M SNPs as 8 contigs of 1.25M, n = 200, A = 100..10000 step 100,
            --findBal --findPos (100x10x44), sharded like config 4.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector FP64: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
MEASURED_FMA_STREAM_TFLOPS = 66.6     # pure FP64 FMA stream, 2 waves per SIMD, all CUs: scripts/ubench_dep.hip on MI355X
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6300 achievable)
SURVEY_FLOP_PER_EVAL = 32        # SURVEY.md 8(d) convention for the reference's FMA + log1p form
CALIBRATION = os.path.join(REPO, 'profiles', 'r04_pmc_calibration.json')


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', type=int, default=4, choices=[3, 4, 5],
                    help='4 (default): BASELINE config 4, whole genome, 22 chromosomes, sharded over the GPUs; 3: one 1M-SNP '
                         'chromosome per GPU; 5: BASELINE config 5, dense-grid stress (10M SNPs in 8 contigs, n=200, 100x10x44 grid)')
    ap.add_argument('--total-snps', type=int, default=None, help='configs 4 / 5: SNPs in the whole workload (default 40M / 10M)')
    ap.add_argument('--snps', type=int, default=1000000, help='config 3: SNPs (= windows) per GPU per step')
    ap.add_argument('--n', type=int, default=100)
    ap.add_argument('--n-spread', type=int, default=0,
                    help='config 3, >0: sample sizes n-spread..n drawn per site (missing data): LUT rows = sum(n_i+1), which '
                         'can exceed LDS and exercises the R-from-L2 path')
    ap.add_argument('--variant', type=int, default=0)
    ap.add_argument('--step', type=int, default=1, help='test site = every step-th SNP (the reference\'s -s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cold-pass', action='store_true', help='no extra (cold) first step before the warm-up steps and no end_to_end_* figures (PMC runs: the '
                                                                'counters must see exactly the timed step)')
    ap.add_argument('--no-parity-sample', action='store_true', help='skip the oracle check of 256 windows of the last step (outside the timed region)')
    ap.add_argument('--cpu-seconds', type=float, default=24.0, help='budget of the faithful CPU port (both legs together)')
    return ap.parse_args(argv)


def load_calibration(config, kernel, build_id):
    """Per-evaluation instruction counts and per-window HBM traffic of the scan kernels, measured with rocprofv3 PMC
    passes (profiles/README.md says how; scripts/pmc_calibrate.py turns the counter CSVs into this file).  An entry
    counts only for the kernel AND the library build it was measured on."""
    try:
        with open(CALIBRATION) as f:
            cal = json.load(f)
    except (OSError, ValueError):
        return None, 'no calibration file'
    ent = cal.get(config)
    if not ent:
        return None, 'no calibration entry %s' % config
    if ent.get('kernel') != kernel:
        return None, 'calibration is for %s, this run used %s' % (ent.get('kernel'), kernel)
    if ent.get('build_id') != build_id:
        return None, 'calibration was measured on library build %s, this is %s: re-run scripts/pmc_collect.sh' % (ent.get('build_id'), build_id)
    return ent, None


def window_work(gen, As, zcut, test_idx, sample=1):
    """(sum over test sites of sum_A W_A(t), sum of W_max(t)): W_A = sites with A*|g_i - t| <= zcut, ties with t excluded
    -- the algorithmic work / bytes of SURVEY.md 8(d).  sample > 1: every sample-th test site, scaled up."""
    import numpy as np
    idx = test_idx[::sample]
    t = gen[idx]
    tot = np.zeros(len(t), dtype=np.int64)
    wmax = np.zeros(len(t), dtype=np.int64)
    ties = np.searchsorted(gen, t, 'right') - np.searchsorted(gen, t, 'left')
    for A in As:
        r = zcut / float(A)
        w = np.searchsorted(gen, t + r, 'right') - np.searchsorted(gen, t - r, 'left') - ties
        w = np.maximum(w, 0)
        tot += w
        wmax = np.maximum(wmax, w)
    scale = len(test_idx) / max(len(idx), 1)
    return float(tot.sum()) * scale, float(wmax.sum()) * scale


# ------------------------------------------------------------------------------------------------ CPU baseline
class _LazyNorm(dict):
    def __init__(self, m):
        super().__init__()
        self.m = m

    def __missing__(self, key):
        ix, ia = key
        return _Gather(self.m.psel[ix, ia], self.m.row)


class _Gather:
    """normProbs[(x,a)][subwindow] == psel[x,a][row[subwindow]]"""

    def __init__(self, tab, row):
        self.tab, self.row = tab, row

    def __getitem__(self, sub):
        return self.tab[self.row[sub]]


def _faithful_model(spec):
    """The oracle's model of one synthetic chromosome, rebuilt from its recipe (worker processes share no arrays)."""
    import numpy as np
    from oracle import bmx_oracle as orc
    from ballermixplus_amd import synth
    from ballermixplus_amd.hostmodel import Grids
    Nc, n, cid, spect, props, min_count, config5 = spec
    phys, gen, k, nn = synth.synth_chromosome(Nc, n, chrom=cid)
    grid = Grids(None, None, True, True, '100,10000,100', None) if config5 else Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    m = orc.Model('B2', gen, k, nn, spect, props, min_count, xs, ab, As)
    # same arithmetic as the reference per window; P_sel gathered from the (k,n) table instead of from 510 materialised
    # N-length arrays (4 GB at N = 1M)
    m._norm_probs = _LazyNorm(m)
    return m, gen


_W = {}


def _worker_init(spec):
    _W['m'], _W['gen'] = _faithful_model(spec)


def _worker_window(i):
    from oracle import bmx_oracle as orc
    t0 = time.time()
    orc.calc_baller_faithful(_W['m'], 0, len(_W['gen']) - 1, _W['gen'][i])
    return time.time() - t0


def _cpu_info():
    model = 'unknown'
    try:
        for l in open('/proc/cpuinfo'):
            if l.startswith('model name'):
                model = l.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return model, avail


def _pool_size(avail):
    """Workers of the multiprocess leg: every CPU this process may run on (nproc-way), within the container's CPU
    quota and with room in memory for one model per worker (~0.4 GB: the arrays of one chromosome + numpy/scipy)."""
    n = max(1, int(avail))
    try:                                   # cgroup v2 CPU quota, if any
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    mem = None
    try:
        v = open('/sys/fs/cgroup/memory.max').read().strip()
        if v != 'max':
            mem = int(v)
    except (OSError, ValueError):
        pass
    try:
        for l in open('/proc/meminfo'):
            if l.startswith('MemAvailable'):
                kb = int(l.split()[1]) * 1024
                mem = kb if mem is None else min(mem, kb)
                break
    except (OSError, ValueError):
        pass
    if mem:
        n = min(n, max(1, int(mem * 0.5 / 0.4e9)))
    return n


def cpu_baseline(specs, sizes, budget_s):
    """Times the ORACLE on this box's host cores (reported baseline, never the product path): the faithful port of
    calcBaller (BalLeRMix+_v1.py:436-507), single process (= what upstream does) and nproc-way multiprocess over windows
    (SURVEY 8d).  `specs`: recipe of every chromosome.  The reference's cost per window grows with the chromosome's length
    (every A touches all N sites, v1:446-457), so the single-process leg takes windows on the longest, a middle and the
    shortest chromosome and reports the SNP-weighted mean of a linear fit t(N) over the workload; the multiprocess leg
    runs on the middle chromosome and is scaled by the same fit."""
    import numpy as np
    import scipy
    import multiprocessing as mp
    sizes = np.asarray(sizes)
    order = np.argsort(sizes)
    picks = sorted(set([int(order[-1]), int(order[len(order) // 2]), int(order[0])]))
    cpu_model, avail = _cpu_info()
    per_window, notes = [], []
    t_all = time.time()
    single_budget = budget_s * 0.5
    mid = picks[len(picks) // 2]
    for ci in picks:
        t0 = time.time()
        m, gen = _faithful_model(specs[ci])
        t_init = time.time() - t0
        N = len(gen)
        from oracle import bmx_oracle as orc
        nwin, t0 = 0, time.time()
        for i in np.linspace(0, N - 1, 64 + 2).astype(int)[1:-1]:
            orc.calc_baller_faithful(m, 0, N - 1, gen[i])
            nwin += 1
            if time.time() - t0 > single_budget / len(picks) and nwin >= 3:
                break
        dt = time.time() - t0
        per_window.append(dt / nwin)
        notes.append('%d windows on %d SNPs: %.2f s/window (table init %.1f s)' % (nwin, N, dt / nwin, t_init))
        del m, gen
    if len(picks) > 1:
        b, a = np.polyfit(sizes[picks].astype(float), np.array(per_window), 1)
        t_of = np.maximum(a + b * sizes, 1e-9)
    else:
        t_of = np.full(len(sizes), per_window[0])
    rate = float(sizes.sum() / (sizes * t_of).sum())
    out = {'value': rate, 'unit': 'windows/s', 'cores': 1, 'kind': 'port',
           'cpu_model': cpu_model, 'cpus_available': avail, 'numpy': np.__version__, 'scipy': scipy.__version__,
           'python': sys.version.split()[0],
           'sample': 'oracle/bmx_oracle.py calc_baller_faithful (numpy, same per-A masks and per-(x,a) sums as '
                     'BalLeRMix+_v1.py:453-505), one process, windows evenly spaced over %s; SNP-weighted mean over the '
                     'workload of the linear fit t(N)' % '; '.join(notes)}
    # nproc-way: the same function, one window per task, over a process pool (workers rebuild the model from the recipe;
    # spawned, not forked: this process has the GPU open)
    try:
        nproc = _pool_size(avail)
        nwin = max(nproc, int(nproc * max(1.0, (budget_s * 0.35) / per_window[picks.index(mid)])))
        N = int(sizes[mid])
        idx = [int(v) for v in np.linspace(0, N - 1, nwin + 2).astype(int)[1:-1]]
        ctx = mp.get_context('spawn')
        t0 = time.time()
        with ctx.Pool(nproc, initializer=_worker_init, initargs=(specs[mid],)) as pool:
            pool.map(_worker_window, idx[:nproc])            # every worker has built its model before the clock starts
            t_start = time.time() - t0
            t1 = time.time()
            times = pool.map(_worker_window, idx, chunksize=1)
            wall = time.time() - t1
        rate_mid = len(idx) / wall
        scale = float(t_of[mid]) * rate      # whole-workload rate / middle-chromosome rate of the single-process fit
        out['multiprocess'] = {'value': rate_mid * scale, 'unit': 'windows/s', 'cores': nproc,
                               'cpu_quota': 'pool = every CPU this process may use: %d hardware threads visible, cgroup cpu.max allows %d' % (avail, nproc),
                               'sample': '%d windows on %d SNPs over a %d-process pool: %.1f windows/s there (%.2f s per window '
                                         'inside a worker, pool start-up %.1f s outside the clock); scaled to the workload by the '
                                         'single-process fit' % (len(idx), N, nproc, rate_mid, float(np.mean(times)), t_start)}
    except Exception as e:      # the baseline must not take the GPU result down with it
        out['multiprocess'] = {'error': repr(e)}
    out['seconds'] = time.time() - t_all
    return out


def parity_sample(chroms, model, spect, props, min_count, xs, ab, As, step, world_size, layout, last, nwin=256):
    """CHECKER, outside the timed region: `nwin` windows of the last step's results, spread over every chromosome, recomputed
    by the plain-C oracle (oracle/bmx_oracle.c: calcBaller, BalLeRMix+_v1.py:436-507) from the ORACLE's own selection table
    (oracle/bmx_oracle.py sel_table: NormalizedBetaBinom, v1:319-433) -- so a fast number from a broken kernel cannot be
    printed unnoticed.  Returns the dict that goes into the JSON line."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from util import c_oracle, c_scan, oracle_R
    from ballermixplus_amd import distributed
    t0 = time.time()
    R = oracle_R('B2', sorted(props), min_count, spect, props, xs, ab)
    L = c_oracle()
    nx, nab = len(xs), len(ab)
    per = max(1, nwin // len(chroms))
    base = np.zeros(world_size, dtype=np.int64)       # where this chromosome's records start in each rank's buffer
    worst, worst_abs, n, bad = 0.0, 0.0, 0, []
    for ci, (gen, k, nn) in enumerate(chroms):
        Nc = len(gen)
        tidx = np.arange(0, Nc, step)
        cnts = layout[ci]
        pick = np.unique(np.linspace(0, len(tidx) - 1, per).astype(np.int64))
        if world_size > 1:
            parts = distributed.assign(len(tidx), world_size)
        rec = np.empty(len(pick), dtype=last[0].dtype)
        for q, j in enumerate(pick):
            if world_size == 1:
                rec[q] = last[0][base[0] + j]
            else:
                r = int((j // distributed.BLOCK) % world_size)
                rec[q] = last[r][base[r] + int(np.searchsorted(parts[r], j))]
        base += np.asarray(cnts, dtype=np.int64)
        t = gen[tidx[pick]]
        oc, ox, oa, oA, on = c_scan(L, R, As, gen, model.rows_of(k, nn), t, np.zeros(len(t), np.int64), np.full(len(t), Nc - 1, np.int64))
        olin = np.where(oA < 0, -1, (oA * nx + ox) * nab + oa)
        err = np.abs(rec['clr'] - oc)
        ok = (rec['lin'] == olin) & (rec['nsites'] == on) & (err <= np.maximum(1e-9, 1e-6 * np.abs(oc)))
        for q in np.where(~ok)[0][:4]:
            bad.append({'chromosome': ci + 1, 'test_site': int(tidx[pick[q]]), 'clr': float(rec['clr'][q]), 'oracle_clr': float(oc[q]),
                        'lin': int(rec['lin'][q]), 'oracle_lin': int(olin[q]), 'nsites': int(rec['nsites'][q]), 'oracle_nsites': int(on[q])})
        worst = max(worst, float(np.max(err / np.maximum(np.abs(oc), 1e-9))))
        worst_abs = max(worst_abs, float(err.max()))
        n += len(t)
    return {'windows': n, 'chromosomes': len(chroms), 'max_rel': worst, 'max_abs': worst_abs, 'mismatches': len(bad), 'first_mismatches': bad[:8],
            'seconds': time.time() - t0,
            'checker': 'oracle/bmx_oracle.c orc_scan on the oracle\'s own table (oracle/bmx_oracle.py sel_table); exact (x, alpha, A, nSites), '
                       'CLR to 1e-6 relative (1e-9 floor); outside the timed region'}


class _QuietStdout:
    """RCCL prints a version banner through C stdio on stdout (flushed at exit, i.e. after
    anything Python printed).  The contract is ONE JSON line on stdout, so fd 1 points at stderr
    while the job runs and is restored, after a C-level flush, just before the line is printed."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def restore(self):
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)

    def __exit__(self, *a):
        return False


def _spawn_ranks(args):
    """--gpus N > 1 without a launcher: this process has made no GPU call (no torch, no HIP); the ranks are fresh child
    processes of torch.distributed.run, and their one JSON line is relayed."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if lines:
        print(lines[-1])
        sys.stdout.flush()
    if r.returncode != 0 or not lines:
        sys.stderr.write('bench.py: the %d-rank run failed (exit code %d)\n' % (args.gpus, r.returncode))
        sys.exit(r.returncode or 1)
    sys.exit(0)


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        _spawn_ranks(args)
    with _QuietStdout() as quiet:
        line = _run(args)
        quiet.restore()
    if line is not None:
        print(line)
        sys.stdout.flush()


def _run(args):
    import numpy as np
    # The process group comes first: nothing below touches the GPU before torch.distributed.run's ranks are up.
    # BMX_DIST_BACKEND=gloo + BMX_SINGLE_DEVICE=1: several ranks sharing ONE GPU with a CPU gather -- only to
    # rehearse the multi-rank control flow (barriers, max-over-ranks, rank-0 output) on a 1-GPU box
    from ballermixplus_amd import distributed
    world = distributed.World.from_env(backend=os.environ.get('BMX_DIST_BACKEND'))
    if world.size != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world.size))
    import torch
    from ballermixplus_amd import _lib, engine, synth
    from ballermixplus_amd.hostmodel import Grids
    rank, dev = world.rank, world.device_index
    torch.cuda.set_device(dev)
    on_gpu = world.backend != 'gloo'
    tdev = torch.device('cuda', dev)

    # ------------------------------------------------------------------ workload
    sharded = args.config in (4, 5)
    n = 200 if args.config == 5 else args.n
    if args.config == 4:
        sizes_c = synth.config4_sizes(args.total_snps or 40000000)
        chrom_ids = list(range(1, len(sizes_c) + 1))
    elif args.config == 5:
        tot = args.total_snps or 10000000
        sizes_c = [tot // 8] * 8
        chrom_ids = list(range(1, 9))
    else:
        sizes_c = [args.snps]
        chrom_ids = [rank + 1]                    # weak scaling: every rank its own chromosome
    chroms = []
    import hashlib
    digest = hashlib.sha256()
    for cid, Nc in zip(chrom_ids, sizes_c):
        phys, gen, k, nn = synth.synth_chromosome(Nc, n, chrom=cid)
        if args.n_spread > 0 and not sharded:      # thin the sample sizes: n_i uniform in [n - spread, n], counts rescaled
            rng = np.random.default_rng(77 + rank)
            n2 = rng.integers(n - args.n_spread, n + 1, Nc)
            k = np.where(k == nn, n2, np.maximum(1, np.minimum(n2 - 1, (k * n2) // nn)))
            nn = n2
        chroms.append((gen, k, nn))
        if rank == 0:
            digest.update(np.ascontiguousarray(gen).tobytes())
            digest.update(np.ascontiguousarray(k, dtype=np.int16).tobytes())
            digest.update(np.ascontiguousarray(nn, dtype=np.int16).tobytes())
    # the helper file of the run: the reference's --getSpect on the concatenation of all chromosomes
    kk = np.concatenate([c[1] for c in chroms])
    nk = np.concatenate([c[2] for c in chroms])
    sizes = sorted(set(nk.tolist()))
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(kk, nk)}
    props = {int(s_): float(sum(f for (a, b), f in spect.items() if b == s_)) for s_ in sizes}
    min_count = int(kk.min())
    del kk, nk
    if args.config == 5:     # --rangeA 100,10000,100 --findBal --findPos  (findBal wins: 10 x, 44 alpha)
        grid = Grids(None, None, True, True, '100,10000,100', None)
    else:
        grid = Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    model = engine.ModelArrays('B2', min_count, sizes, spect, props, xs, ab)
    zcut = _lib.lib().bmx_alpha_cut()

    # ONE resident context: the table, and per chromosome (slot) the site arrays and this rank's test sites, in HBM from here on
    # (t_abi: wall clock inside the calls a user of the C ABI makes once per genome -- context, table build, uploads, test-site location
    # and planning; with the first step below it is the COLD pass the line reports as end_to_end_*)
    phases = {}

    def lap(name, since):
        phases[name] = phases.get(name, 0.0) + time.perf_counter() - since
        return time.perf_counter()

    t_abi = t_c = time.perf_counter()
    ctx = engine.Context(dev)
    t_c = lap('context', t_c)
    ctx.set_variant(args.variant)
    ctx.set_model(model, As)
    lap('table_build', t_c)
    t_abi = time.perf_counter() - t_abi
    slots, counts_by_rank = [], np.zeros(world.size, dtype=np.int64)
    evals_per_step, bytes_per_step, windows_per_step, plan = 0.0, 0.0, 0, None
    layout = []                                   # rank 0: (chromosome, per-rank test-site counts) in slot order
    t_plan = time.time()
    for ci, (gen, k, nn) in enumerate(chroms):
        Nc = len(gen)
        tidx = np.arange(0, Nc, args.step)
        if sharded:
            parts = distributed.assign(len(tidx), world.size)
            mine = tidx[parts[rank]]
            cnts = [len(p) for p in parts]
        else:
            mine, cnts = tidx, [len(tidx)] * world.size
        counts_by_rank += np.asarray(cnts)
        layout.append(cnts)
        t_c = t_c0 = time.perf_counter()
        ctx.select_slot(ci)
        rows_c = model.rows_of(k, nn)
        t_c = lap('host_row_lookup', t_c)
        ctx.set_sites(gen, rows_c)
        t_c = lap('site_uploads', t_c)
        if len(mine):
            tg = gen[::args.step] if len(mine) == len(tidx) else gen[mine]        # (every test site is this rank's: a view, no gather of 40 M positions)
            ctx.set_tests(tg)                 # no window bounds: every window holds all sites (the reference's default mode)
            lap('test_sites_upload_locate_plan', t_c)
        t_abi += time.perf_counter() - t_c0
        if len(mine):
            slots.append(ci)
            if plan is None:
                plan = ctx.plan()
        if rank == 0:            # algorithmic work of the WHOLE job (all ranks), from a 1/64 sample of the test sites
            wsum, wmax = window_work(gen, As, zcut, tidx, sample=64 if len(tidx) > 200000 else 1)
            mult = 1 if sharded else world.size
            evals_per_step += wsum * len(xs) * len(ab) * mult
            bytes_per_step += (wmax * 10.0 + 24.0 * len(tidx)) * mult      # W_max*10 B + 24 B per window
            windows_per_step += len(tidx) * mult
    t_plan = time.time() - t_plan
    stream_bytes = 0
    for ci in slots:
        ctx.select_slot(ci)
        stream_bytes += ctx.plan()['stream_bytes']

    # the step's ONE result transfer: buffers allocated once
    mine_total = int(counts_by_rank[rank])
    pad = int(counts_by_rank.max())
    rec_dt = _lib.RECORD_DTYPE
    if world.distributed:
        gdev = tdev if on_gpu else torch.device('cpu')
        send = torch.zeros((pad, 2), dtype=torch.int64, device=gdev)
        recv = torch.empty((world.size, pad, 2), dtype=torch.int64, device=gdev) if rank == 0 else None
        host = torch.empty((world.size, pad, 2), dtype=torch.int64).pin_memory() if (rank == 0 and on_gpu) else None
        send_np = send.numpy().view(rec_dt).reshape(-1) if not on_gpu else None
    else:
        host = torch.empty((max(mine_total, 1), 2), dtype=torch.int64).pin_memory()
        host_np = host.numpy().view(rec_dt).reshape(-1)

    # BMX_NATIVE_GATHER=1: the gather inside libbmxscan (bmx_comm_gather_records: ncclSend / ncclRecv on the context's stream, no
    # torch in the data path; torch.distributed only carries the 128-byte RCCL id once).  Not the default: with one GPU per box
    # it could only be run with a single rank (tests/test_gpu_round4.py), N > 1 is unmeasured.
    native = {'comm': None}
    if os.environ.get('BMX_NATIVE_GATHER') == '1' and world.distributed and on_gpu:
        native['comm'] = world.native_comm(ctx)
        native['host'] = np.empty(int(counts_by_rank.sum()), dtype=rec_dt) if rank == 0 else None
        native['offs'] = np.concatenate(([0], np.cumsum(counts_by_rank)))

    def collect():
        """All records of this rank's slots -> the host (one process) or rank 0 (ONE gather)."""
        if native['comm'] is not None:
            got = native['comm'].gather_records(counts_by_rank, root=0, out=native['host'])
            if rank != 0:
                return None
            return [got[int(native['offs'][r]):int(native['offs'][r + 1])] for r in range(world.size)]
        if world.distributed:
            if on_gpu:       # device -> device pack, then RCCL
                ctx.pack_records(device_ptr=send.data_ptr(), cap=pad)
            else:            # rehearsal mode: through the host
                ctx.pack_records(out=send_np)
            torch.distributed.gather(send, list(recv.unbind(0)) if rank == 0 else None, dst=0)
            if on_gpu:
                if rank == 0:
                    host.copy_(recv, non_blocking=True)
                torch.cuda.current_stream().synchronize()      # the send buffer is rewritten by the next step's pack
            if rank != 0:
                return None
            got = (host if on_gpu else recv).numpy().view(rec_dt).reshape(world.size, pad)
            return [got[r, :int(counts_by_rank[r])] for r in range(world.size)]
        return [ctx.pack_records(out=host_np)]

    def barrier():
        if world.distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def one_step(kernel_ms=None):
        for ci in slots:                     # back to back on the context's stream; nothing waits in between
            ctx.select_slot(ci)
            ctx.scan()
        got = collect()                      # waits for the stream once
        if kernel_ms is not None:
            for ci in slots:
                ctx.select_slot(ci)
                kernel_ms.append(ctx.last_scan_ms())
        return got

    # ---- the COLD pass: everything above that a caller of the C ABI pays once per genome (t_abi) plus the FIRST step on the fresh
    # context (scans + result transfer, first-touch costs included); untimed, before the warm-up steps
    t_cold, cold = None, None
    if not args.no_cold_pass:
        barrier()
        t0 = time.perf_counter()
        cold = one_step()
        barrier()
        phases['first_step'] = time.perf_counter() - t0
        t_cold = t_abi + phases['first_step']
        if rank == 0:
            cold = [np.array(g, copy=True) for g in cold]       # the transfer buffers are reused by the next step
    for _ in range(args.warmup):
        one_step()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = one_step(kernel_ms)
    barrier()
    dt = time.perf_counter() - t0
    if world.distributed:
        tmax = torch.tensor([dt, sum(kernel_ms)], dtype=torch.float64, device=tdev if on_gpu else torch.device('cpu'))
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt, k_total = float(tmax[0].item()), float(tmax[1].item())
    else:
        k_total = float(sum(kernel_ms))

    if world.distributed and t_cold is not None:
        tc = torch.tensor([t_cold], dtype=torch.float64, device=tdev if on_gpu else torch.device('cpu'))
        torch.distributed.all_reduce(tc, op=torch.distributed.ReduceOp.MAX)
        t_cold = float(tc[0].item())

    if rank == 0:
        checksum = float(sum(float(np.sum(g['clr'])) for g in last))
        cold_same = cold is not None and all(np.array_equal(a, b) for a, b in zip(last, cold))
        nrec = int(sum(len(g) for g in last))
        k_step_s = k_total / args.steps * 1e-3                     # scan kernels of one step (slowest rank)
        windows = float(windows_per_step) * args.steps
        evals_s = evals_per_step / k_step_s
        kname = plan['kernel'] if plan else 'none'
        build_id = _lib.lib().bmx_build_id().decode()
        cal_key = 'config%d' % args.config + ('_step%d' % args.step if args.step != 1 else '') + \
                  ('_nspread%d' % args.n_spread if args.n_spread else '')
        cal, why = load_calibration(cal_key, kname, build_id)
        nchrom = len(chroms)
        workload = {
            4: 'BASELINE config 4: synthetic whole genome, %d SNPs over %d chromosomes (GRCh37 proportions), n=%d, default '
               '31x10x51 (A,x,alpha) grid, B2 scan, every SNP a test site; site arrays of all chromosomes resident on every GPU, '
               'test sites dealt to the GPUs in blocks of 4096' % (sum(sizes_c), nchrom, n),
            3: 'BASELINE config 3: synthetic single chromosome, %d SNPs, n=%d, default 31x10x51 (A,x,alpha) grid, B2 scan, '
               'every SNP a test site; one chromosome per GPU' % (sizes_c[0], n),
            5: 'BASELINE config 5: %d SNPs as %d contigs of %d, n=%d, A=100..10000 step 100, --findBal --findPos '
               '(100x10x44 grid), B2 scan, every SNP a test site; test sites dealt to the GPUs in blocks of 4096' % (sum(sizes_c), nchrom, sizes_c[0], n)}[args.config]
        if args.step != 1:
            workload += '; test site = every %d-th SNP (-s %d)' % (args.step, args.step)
        if args.n_spread:
            workload += '; sample sizes %d..%d drawn per site' % (n - args.n_spread, n)
        res = {
            'metric': 'CLR windows/sec (B2 scan, n=%d)' % n,
            'value': windows / dt,
            'unit': 'windows/s',
            'n_gpus': world.size, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong' if sharded else 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic', 'argv': sys.argv[1:],
            'config': {'workload': workload, 'windows_per_step': windows_per_step, 'launches_per_step': len(slots),
                       'grid_points': len(As) * len(xs) * len(ab),
                       'parallelism': ('test-site sharding, dp%d; one context per GPU holds every chromosome; ONE gather of all 16-B records '
                                       'to rank 0 per step, inside the step' % world.size) if world.distributed else
                                      'one GPU, one context holding every chromosome; the 16-B result records of the whole step are copied to '
                                      'pinned host memory once, inside the step',
                       'kernel_only_windows_per_s': windows_per_step / k_step_s,
                       'kernel_ms_per_step': k_step_s * 1e3,
                       'records_per_step': nrec,
                       'checksum_clr': checksum,
                       'end_to_end_windows_per_s': windows_per_step / t_cold if t_cold else None,
                       'end_to_end_seconds': t_cold,
                       'end_to_end_phases_rank0': {k_: round(v_, 4) for k_, v_ in phases.items()} if t_cold else None,
                       'end_to_end_note': 'the COLD pass of this run, outside the timed steps: from host buffers through the calls a user of the C '
                                          'ABI makes once per genome -- context creation, table build (K1), H2D of all site arrays and test sites, '
                                          'test-site location + planning / counting pass (wall clock inside those calls) -- plus the FIRST step on '
                                          'the fresh context: the scans and the result records to the host (N > 1: the gather); slowest rank; records '
                                          'bitwise equal to the timed steps\': %s' % cold_same,
                       'input_sha256': digest.hexdigest(),
                       'library_build': build_id,
                       'plan_seconds_outside_timed_region': t_plan,
                       'prepared_stream_bytes_per_window': stream_bytes / max(mine_total, 1)},
            'roofline': {
                'bound': 'valu_fp64', 'kernel': kname,
                'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'traffic': None, 'achieved': None, 'frac': None,
                'kernel_ms': k_step_s * 1e3 / max(len(slots), 1), 'launches': len(slots),
                'evals_per_step': evals_per_step, 'evals_per_s': evals_s,
                'survey_convention_tflops': evals_s * SURVEY_FLOP_PER_EVAL / 1e12,
                'note': 'The scan is bound by vector-instruction issue, not by HBM or MFMA (no contraction on this path). '
                        'achieved/frac: the FP64 arithmetic the scan kernels execute (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64: FMA = 2 flop, MUL/ADD = 1 '
                        'per lane) / kernel time against the vector FP64 peak. valu_issue_frac: ALL executed VALU wave-instructions x 128 '
                        'flop-equivalents against the same peak = the share of the chip\'s vector issue slots the kernels fill (every VALU '
                        'instruction, FP64 or not, takes one slot) -- an occupancy figure, not a flop rate. Both use '
                        'per-evaluation instruction counts measured with rocprofv3 PMC passes on THIS library build (calibration) x '
                        'evaluations per second measured here with HIP events; null when the calibration is for another kernel or build. '
                        'kernel_ms: HIP events around each chromosome\'s launches (per-group preparation kernel + scan kernel + finalize) on the '
                        'library\'s stream. Algorithmic work: evals_per_s mixture-likelihood evaluations; under SURVEY 8(d) '
                        '"1 evaluation = 32 flop" that is survey_convention_tflops, above the peak because near sites are multiplied four '
                        'per step in product form and far sites are summed as moments.'},
            'roofline_hbm': {
                'bound': 'hbm', 'kernel': kname,
                'achieved': bytes_per_step / k_step_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': bytes_per_step / k_step_s / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                'algorithmic_bytes_per_step': bytes_per_step},
        }
        if cal:
            r = res['roofline']
            r['fp64_flops_per_eval'] = cal['fp64_flops_per_eval']
            r['achieved'] = evals_s * cal['fp64_flops_per_eval'] / 1e12          # FP64 flop actually executed per second
            r['frac'] = r['achieved'] / FP64_VALU_PEAK_TFLOPS
            r['fp64_tflops'], r['fp64_flops_frac'] = r['achieved'], r['frac']    # (the names round 3's line used)
            r['valu_insts_per_64_evals'] = cal['valu_per_64_evals']
            r['valu_issue_tflops_equiv'] = evals_s * cal['valu_per_64_evals'] * 128 / 64 / 1e12
            r['valu_issue_frac'] = r['valu_issue_tflops_equiv'] / FP64_VALU_PEAK_TFLOPS
            # what a pure v_fma_f64 stream reaches on this chip at two waves per SIMD (scripts/ubench_dep.hip: 4.45 cycles per
            # instruction at a core clock that falls to 2.26 GHz under it; profiles/r04_ubench_dependent_fp64.txt) -- context for
            # `frac`, which stays against the nominal peak
            r['measured_fma_stream_tflops'] = MEASURED_FMA_STREAM_TFLOPS
            r['frac_of_measured_fma_stream'] = r['achieved'] / MEASURED_FMA_STREAM_TFLOPS
            r['calibration'] = cal['source']
            # HBM traffic of one average launch, from the PMC counters per window (FETCH_SIZE doubled as the guide
            # prescribes for gfx950, WRITE_SIZE as read)
            per_launch_windows = windows_per_step / max(len(slots), 1)
            per_window = cal['hbm_read_bytes_per_window'] + cal['hbm_write_bytes_per_window']
            r['traffic'] = res['roofline_hbm']['traffic'] = per_window * per_launch_windows
            res['roofline_hbm']['measured_hbm_gbs'] = per_window * windows_per_step / k_step_s / 1e9
        else:
            res['roofline']['calibration_warning'] = why
        if not args.no_parity_sample and not args.n_spread:
            try:
                ps = parity_sample(chroms, model, spect, props, min_count, xs, ab, As, args.step, world.size, layout, last)
                res['parity_sample'] = ps
                res['parity_sample_max_rel'] = ps['max_rel']
                if ps['mismatches']:
                    res['parity_sample_FAILED'] = True
                    sys.stderr.write('bench.py: PARITY SAMPLE FAILED: %r\n' % (ps['first_mismatches'],))
            except Exception as e:
                res['parity_sample'] = {'error': repr(e)[:500]}
        if not args.no_cpu_baseline and world.size == 1:      # reported baseline: rank 0, N = 1 only
            specs = [(int(Nc), int(n), int(cid), spect, props, min_count, args.config == 5) for cid, Nc in zip(chrom_ids, sizes_c)]
            if args.n_spread:
                res['cpu_baseline'] = {'skipped': 'not defined for --n-spread runs (the recipe of the thinned sample sizes is not shared with the workers)'}
            else:
                try:
                    res['cpu_baseline'] = cpu_baseline(specs, sizes_c, args.cpu_seconds)
                except Exception as e:       # the GPU measurement above stands on its own: say what went wrong with the baseline leg
                    res['cpu_baseline'] = {'error': repr(e)[:500]}
        line = json.dumps(res)
    else:
        line = None
    if native['comm'] is not None:
        native['comm'].close()
    ctx.close()
    world.finish()
    return line


if __name__ == '__main__':
    main()
