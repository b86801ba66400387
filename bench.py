#!/usr/bin/env python3
"""bench.py -- CLR windows/s of the B_2 scan (BASELINE.json metric) on N MI355X.

Default workload = BASELINE config 4, the north_star target: a synthetic whole genome, 40M SNPs over 22
chromosomes (GRCh37 proportions), n = 100, default 31x10x51 (A, x, alpha) grid, every SNP a test site.
One step = one pass of the hot path over the whole genome.  Every chromosome's site arrays and test sites are
resident in HBM before the timed region (one scan context per chromosome); a step launches the 22 scans one after
another and brings the 16-byte result records of every test site back to the host (N = 1) or to rank 0 (N > 1, one
gather per chromosome -- RCCL over xGMI), so the result copy is INSIDE the timed step.  With N ranks every rank holds
all site arrays (400 MB) and scans every N-th block of 4096 test sites of each chromosome: strong scaling, no
data-path collective, results bitwise independent of N.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 4|3|5] [--total-snps 40000000] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

--config 3: one 1M-SNP chromosome per GPU (weak scaling; the launch the PMC profiles in profiles/ are taken on);
--config 5: the dense-grid stress (n = 200, A = 100..10000 step 100, --findBal --findPos) on one chromosome per GPU.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector FP64: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6300 achievable)
SURVEY_FLOP_PER_EVAL = 32        # SURVEY.md 8(d) convention for the reference's FMA + log1p form
CALIBRATION = os.path.join(REPO, 'profiles', 'r02_pmc_calibration.json')


def load_calibration(config):
    """Per-evaluation instruction counts and per-window HBM traffic of the scan kernel, measured with rocprofv3 PMC
    passes (profiles/README.md says how; scripts/pmc_calibrate.py turns the counter CSVs into this file)."""
    try:
        with open(CALIBRATION) as f:
            cal = json.load(f)
    except (OSError, ValueError):
        return None
    return cal.get('config%d' % config) or cal.get('config3')


def window_work(gen, As, zcut, test_idx, sample=1):
    """(sum over test sites of sum_A W_A(t), sum of W_max(t)): W_A = sites with A*|g_i - t| <= zcut, ties with t excluded
    -- the algorithmic work / bytes of SURVEY.md 8(d).  sample > 1: every sample-th test site, scaled up."""
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


def cpu_baseline(chroms, spect, props, grid, budget_s, n_windows_c):
    """Times the ORACLE on this box's host cores (reported baseline, never the product path).  `chroms`: list of
    (gen, k, nn).  The reference's cost per window grows with the chromosome's length (every A touches all N sites,
    v1:446-457), so windows are taken on the longest, a middle and the shortest chromosome and the whole-workload
    rate is the SNP-weighted mean of a linear fit t(N)."""
    from oracle import bmx_oracle as orc
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from util import c_oracle, c_scan
    xs, ab, As = grid.scan_order()
    sizes = np.array([len(c[0]) for c in chroms])
    order = np.argsort(sizes)
    picks = sorted(set([int(order[-1]), int(order[len(order) // 2]), int(order[0])]))
    per_window, notes, models = [], [], {}
    t_all = time.time()
    for ci in picks:
        gen, k, nn = chroms[ci]
        N = len(gen)
        t0 = time.time()
        m = orc.Model('B2', gen, k, nn, spect, props, int(min(c[1].min() for c in chroms)), xs, ab, As)
        # same arithmetic as the reference per window; P_sel gathered from the (k,n) table instead
        # of from 510 materialised N-length arrays (4 GB at N = 1M)
        m._norm_probs = _LazyNorm(m)
        models[ci] = m
        t_init = time.time() - t0
        nwin, t0 = 0, time.time()
        for i in np.linspace(0, N - 1, 64 + 2).astype(int)[1:-1]:
            orc.calc_baller_faithful(m, 0, N - 1, gen[i])
            nwin += 1
            if time.time() - t0 > budget_s / len(picks) and nwin >= 3:
                break
        dt = time.time() - t0
        per_window.append(dt / nwin)
        notes.append('%d windows on %d SNPs: %.2f s/window (table init %.1f s)' % (nwin, N, dt / nwin, t_init))
    if len(picks) > 1:
        b, a = np.polyfit(sizes[picks].astype(float), np.array(per_window), 1)
        t_of = np.maximum(a + b * sizes, 1e-9)
    else:
        t_of = np.full(len(sizes), per_window[0])
    rate = float(sizes.sum() / (sizes * t_of).sum())
    out = {'value': rate, 'unit': 'windows/s', 'cores': 1, 'kind': 'port',
           'sample': 'oracle/bmx_oracle.py calc_baller_faithful (numpy, same per-A masks and per-(x,a) sums as '
                     'BalLeRMix+_v1.py:453-505), windows evenly spaced over %s; SNP-weighted mean over the '
                     'workload of the linear fit t(N); %.0f s in all' % ('; '.join(notes), time.time() - t_all)}
    try:        # optimised C restatement on all cores, for scale
        L = c_oracle()
        cores = os.cpu_count() or 1
        ci = picks[len(picks) // 2]
        gen, k, nn = chroms[ci]
        m = models[ci]
        idx = np.linspace(0, len(gen) - 1, n_windows_c + 2).astype(int)[1:-1]
        t0 = time.time()
        c_scan(L, m.R, As, gen, m.row, gen[idx], np.zeros(len(idx), np.int64), np.full(len(idx), len(gen) - 1, np.int64))
        dt = time.time() - t0
        out['c_port'] = {'value': len(idx) / dt, 'unit': 'windows/s', 'cores': cores,
                         'sample': '%d windows on %d SNPs, oracle/bmx_oracle.c orc_scan (log1p/LUT form, OpenMP), %.1f s' % (len(idx), len(gen), dt)}
    except Exception as e:  # the C leg is optional
        out['c_port'] = {'error': str(e)}
    return out


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


def main():
    with _QuietStdout() as quiet:
        line = _run()
        quiet.restore()
    if line is not None:
        print(line)
        sys.stdout.flush()


def _run():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', type=int, default=4, choices=[3, 4, 5],
                    help='4 (default): BASELINE config 4, whole genome, 22 chromosomes, sharded over the GPUs; 3: one 1M-SNP '
                         'chromosome per GPU; 5: dense-grid stress (n=200, 100x10x44 grid) on one chromosome per GPU')
    ap.add_argument('--total-snps', type=int, default=40000000, help='config 4: SNPs in the whole genome')
    ap.add_argument('--snps', type=int, default=1000000, help='configs 3 and 5: SNPs (= windows) per GPU per step')
    ap.add_argument('--n', type=int, default=100)
    ap.add_argument('--n-spread', type=int, default=0,
                    help='configs 3/5, >0: sample sizes n-spread..n drawn per site (missing data): LUT rows = sum(n_i+1), which '
                         'can exceed LDS and exercises the R-from-L2 path')
    ap.add_argument('--variant', type=int, default=0)
    ap.add_argument('--step', type=int, default=1, help='test site = every step-th SNP (the reference\'s -s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0, help='budget of the faithful CPU port')
    args = ap.parse_args()

    # The process group comes first: nothing below touches the GPU before torch.distributed.run's ranks are up.
    # BMX_DIST_BACKEND=gloo + BMX_SINGLE_DEVICE=1: several ranks sharing ONE GPU with a CPU gather -- only to
    # rehearse the multi-rank control flow (barriers, max-over-ranks, rank-0 output) on a 1-GPU box
    from ballermixplus_amd import distributed
    world = distributed.World.from_env(backend=os.environ.get('BMX_DIST_BACKEND'))
    if world.size != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world.size, args.gpus))
    import torch
    from ballermixplus_amd import _lib, engine, synth
    from ballermixplus_amd.hostmodel import Grids
    rank, dev = world.rank, world.device_index
    torch.cuda.set_device(dev)
    on_gpu = world.backend != 'gloo'

    # ------------------------------------------------------------------ workload
    n = 200 if args.config == 5 else args.n
    if args.config == 4:
        sizes_c = synth.config4_sizes(args.total_snps)
        chrom_ids = list(range(1, len(sizes_c) + 1))
    else:
        sizes_c = [args.snps]
        chrom_ids = [rank + 1]                    # weak scaling: every rank its own chromosome
    chroms = []
    for cid, Nc in zip(chrom_ids, sizes_c):
        phys, gen, k, nn = synth.synth_chromosome(Nc, n, chrom=cid)
        if args.n_spread > 0 and args.config != 4:      # thin the sample sizes: n_i uniform in [n - spread, n], counts rescaled
            rng = np.random.default_rng(77 + rank)
            n2 = rng.integers(n - args.n_spread, n + 1, Nc)
            k = np.where(k == nn, n2, np.maximum(1, np.minimum(n2 - 1, (k * n2) // nn)))
            nn = n2
        chroms.append((gen, k, nn))
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

    # one resident context per chromosome: table, site arrays and this rank's test sites in HBM from here on
    ctxs, shares, evals_per_step, bytes_per_step, windows_per_step = [], [], 0.0, 0.0, 0
    for gen, k, nn in chroms:
        Nc = len(gen)
        tidx = np.arange(0, Nc, args.step)
        if args.config == 4:
            parts = distributed.assign(len(tidx), world.size)
            mine, counts = tidx[parts[rank]], [len(p) for p in parts]
        else:
            parts, mine, counts = None, tidx, None
        ctx = engine.Context(dev)
        ctx.set_variant(args.variant)
        ctx.set_model(model, As)
        ctx.set_sites(gen, model.rows_of(k, nn))
        if len(mine):
            ctx.set_tests(gen[mine], np.zeros(len(mine), np.int64), np.full(len(mine), Nc - 1, np.int64))
        ctxs.append(ctx)
        shares.append((len(mine), counts))
        if rank == 0:            # algorithmic work of the WHOLE job (all ranks), from a 1/64 sample of the test sites
            wsum, wmax = window_work(gen, As, zcut, tidx, sample=64 if len(tidx) > 200000 else 1)
            mult = 1 if args.config == 4 else world.size
            evals_per_step += wsum * len(xs) * len(ab) * mult
            bytes_per_step += (wmax * 10.0 + 24.0 * len(tidx)) * mult      # W_max*10 B + 24 B per window
            windows_per_step += len(tidx) * mult

    rec_dt = _lib.RECORD_DTYPE

    def collect(ctx, share):
        """The step's result copy: records to the host (one process) or to rank 0 (one gather per chromosome)."""
        mine_n, counts = share
        if world.distributed:
            if counts is None:
                counts = [mine_n] * world.size
            if mine_n == 0:
                rec = np.zeros(0, dtype=rec_dt)
            elif on_gpu:      # zero-copy view of the library's device records -> RCCL
                rec = torch.as_tensor(distributed._DevArray(ctx.records(), 2 * mine_n, '<i8'), device=torch.device('cuda', dev))
            else:             # rehearsal mode: through the host
                rec = ctx.fetch_records()
            got = world.gather_records(rec, counts)
            return None if got is None else got
        return ctx.fetch_records() if mine_n else np.zeros(0, dtype=rec_dt)

    def barrier():
        if world.distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def one_step(kernel_ms=None):
        last = None
        for ctx, share in zip(ctxs, shares):
            if share[0]:
                ctx.scan()
                ctx.sync()                       # results must be complete before the copy / gather reads them
                if kernel_ms is not None:
                    kernel_ms.append(ctx.last_scan_ms())
            last = collect(ctx, share)
        return last

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
        tmax = torch.tensor([dt, sum(kernel_ms)], dtype=torch.float64, device=torch.device('cuda', dev) if on_gpu else torch.device('cpu'))
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt, k_total = float(tmax[0].item()), float(tmax[1].item())
    else:
        k_total = float(sum(kernel_ms))

    if rank == 0:
        if isinstance(last, list):
            checksum = float(sum(float(np.sum(g['clr'])) for g in last))
        else:
            checksum = float(np.sum(last['clr']))
        k_step_s = k_total / args.steps * 1e-3                     # scan kernels of one step (slowest rank)
        windows = float(windows_per_step) * args.steps
        evals_s = evals_per_step / k_step_s
        cal = load_calibration(args.config)
        kname = 'clr_scan_grouped_kernel<16,true,3>'
        workload = {
            4: 'BASELINE config 4: synthetic whole genome, %d SNPs over %d chromosomes (GRCh37 proportions), n=%d, default '
               '31x10x51 (A,x,alpha) grid, B2 scan, every SNP a test site; site arrays of all chromosomes resident on every GPU, '
               'test sites dealt to the GPUs in blocks of 4096' % (sum(sizes_c), len(sizes_c), n),
            3: 'BASELINE config 3: synthetic single chromosome, %d SNPs, n=%d, default 31x10x51 (A,x,alpha) grid, B2 scan, '
               'every SNP a test site; one chromosome per GPU' % (sizes_c[0], n),
            5: 'BASELINE config 5 grid on one chromosome per GPU: %d SNPs, n=%d, A=100..10000 step 100, --findBal --findPos '
               '(100x10x44), every SNP a test site' % (sizes_c[0], n)}[args.config]
        res = {
            'metric': 'CLR windows/sec (B2 scan, n=%d)' % n,
            'value': windows / dt,
            'unit': 'windows/s',
            'n_gpus': world.size, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong' if args.config == 4 else 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workload, 'windows_per_step': windows_per_step, 'launches_per_step': len(ctxs),
                       'grid_points': len(As) * len(xs) * len(ab),
                       'parallelism': 'test-site sharding, dp%d; one gather of 16-B records per chromosome to rank 0 inside the step'
                                      % world.size if world.distributed else
                                      'one GPU; the 16-B result records are copied to the host inside the step',
                       'kernel_only_windows_per_s': windows_per_step / k_step_s,
                       'kernel_ms_per_step': k_step_s * 1e3,
                       'checksum_clr_last_chromosome': checksum},
            'roofline': {
                'bound': 'valu_fp64', 'kernel': kname,
                'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'traffic': None,
                'kernel_ms': k_step_s * 1e3 / len(ctxs), 'launches': len(ctxs),
                'evals_per_step': evals_per_step, 'evals_per_s': evals_s,
                'survey_convention_tflops': evals_s * SURVEY_FLOP_PER_EVAL / 1e12,
                'note': 'The scan is bound by vector-instruction issue, not by HBM or MFMA (no contraction on this path). '
                        'achieved/frac: executed VALU wave-instructions x 128 flop / kernel time against the vector FP64 peak = the '
                        'share of the chip\'s vector issue slots the kernel fills (every VALU instruction, FP64 or not, takes one slot). '
                        'fp64_flops_frac: the FP64 arithmetic alone (FMA = 2 flop, MUL/ADD = 1 per lane) against the same peak. Both use '
                        'per-evaluation instruction counts measured with rocprofv3 PMC passes (calibration) x evaluations per second '
                        'measured here with HIP events. Algorithmic work: evals_per_s mixture-likelihood evaluations; under SURVEY 8(d) '
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
            r['valu_insts_per_64_evals'] = cal['valu_per_64_evals']
            r['achieved'] = evals_s * cal['valu_per_64_evals'] * 128 / 64 / 1e12
            r['frac'] = r['achieved'] / FP64_VALU_PEAK_TFLOPS
            r['fp64_flops_per_eval'] = cal['fp64_flops_per_eval']
            r['fp64_tflops'] = evals_s * cal['fp64_flops_per_eval'] / 1e12
            r['fp64_flops_frac'] = r['fp64_tflops'] / FP64_VALU_PEAK_TFLOPS
            r['calibration'] = cal['source']
            # HBM traffic of one average launch, from the PMC counters per window (FETCH_SIZE doubled as the guide
            # prescribes for gfx950, WRITE_SIZE as read)
            per_launch_windows = windows_per_step / len(ctxs)
            r['traffic'] = res['roofline_hbm']['traffic'] = (cal['hbm_read_bytes_per_window'] + cal['hbm_write_bytes_per_window']) * per_launch_windows
            res['roofline_hbm']['measured_hbm_gbs'] = (cal['hbm_read_bytes_per_window'] + cal['hbm_write_bytes_per_window']) * windows_per_step / k_step_s / 1e9
        else:
            res['roofline']['achieved'] = None
            res['roofline']['frac'] = None
        if not args.no_cpu_baseline and world.size == 1:      # reported baseline: rank 0, N = 1 only
            res['cpu_baseline'] = cpu_baseline(chroms, spect, props, grid, args.cpu_seconds, 256)
        line = json.dumps(res)
    else:
        line = None
    for ctx in ctxs:
        ctx.close()
    world.finish()
    return line


if __name__ == '__main__':
    main()
