#!/usr/bin/env python3
"""bench.py -- CLR windows/s of the B_2 scan (BASELINE.json metric) on N MI355X.

One step = one pass of the hot path over one batch: every SNP of a synthetic 1M-SNP, n=100
chromosome is a test site and gets the full default (A, x, alpha_beta) grid search
(BASELINE config 3; at N>1 every rank scans its own chromosome of the same size = weak scaling,
the shape of config 4, followed by the RCCL all-gather of the result records).  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--snps 1000000] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
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
# The scan kernel is bound by vector-instruction issue (every VALU instruction, FP64 or not, takes one
# 4-cycle issue slot of its SIMD).  Its roofline figure prices every executed VALU wave-instruction as one
# FP64 FMA (64 lanes x 2 flop) against the vector FP64 peak, i.e. it is the fraction of the chip's vector
# issue slots the kernel fills.  The instruction count per unit of work is a PMC measurement
# (SQ_INSTS_VALU over one launch, profiles/r01_pmc_scan_kernel_200k_windows.txt): 2.889e10 wave-instructions
# for 1.899e12 evaluations (one site x one grid pair x one test site) = 0.974 per 64 evaluations on config 3.
# (The exact product form alone needs >= 1.53: four sites per step, 4 FMA + 1 MUL per test site; the far field
# -- 3/4 of the sites -- is summed as per-row moments of the log1p series and costs almost nothing per pair.)
VALU_PER_64_EVALS = 0.974
FLOP_PER_EVAL = VALU_PER_64_EVALS * 128 / 64
SURVEY_FLOP_PER_EVAL = 32        # SURVEY.md 8(d) convention for the reference's FMA + log1p form


def window_work(gen, As, zcut, test_idx):
    """sum_A W_A(t) per test site (sites with A*|g_i - t| <= zcut, ties with t excluded) and the
    widest window W_max(t) (A_min) -- the algorithmic work / bytes of SURVEY.md 8(d)."""
    t = gen[test_idx]
    tot = np.zeros(len(t), dtype=np.int64)
    wmax = np.zeros(len(t), dtype=np.int64)
    ties = np.searchsorted(gen, t, 'right') - np.searchsorted(gen, t, 'left')
    for A in As:
        r = zcut / float(A)
        w = np.searchsorted(gen, t + r, 'right') - np.searchsorted(gen, t - r, 'left') - ties
        w = np.maximum(w, 0)
        tot += w
        wmax = np.maximum(wmax, w)
    return tot, wmax


def cpu_baseline(gen, k, nn, spect, props, grid, n_windows_faithful, n_windows_c):
    """Times the ORACLE on this box's host cores (reported baseline, never the product path)."""
    from oracle import bmx_oracle as orc
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from util import c_oracle, c_scan
    xs, ab, As = grid.scan_order()
    N = len(gen)
    t0 = time.time()
    m = orc.Model('B2', gen, k, nn, spect, props, int(k.min()), xs, ab, As)
    # same arithmetic as the reference per window; P_sel gathered from the (k,n) table instead
    # of from 510 materialised N-length arrays (4 GB at N = 1M)
    m._norm_probs = _LazyNorm(m)
    t_init = time.time() - t0
    idx = np.linspace(0, N - 1, n_windows_faithful + 2).astype(int)[1:-1]
    t0 = time.time()
    for i in idx:
        orc.calc_baller_faithful(m, 0, N - 1, gen[i])
    dt = time.time() - t0
    out = {'value': len(idx) / dt, 'unit': 'windows/s', 'cores': 1, 'kind': 'port',
           'sample': '%d windows evenly spaced over the %d-SNP chromosome, oracle/bmx_oracle.py '
                     'calc_baller_faithful (numpy, same per-A masks and per-(x,a) sums as '
                     'BalLeRMix+_v1.py:453-505), %.1f s; table init %.1f s' % (len(idx), N, dt, t_init)}
    # optimised C restatement on all cores, for scale
    try:
        L = c_oracle()
        cores = os.cpu_count() or 1
        idx = np.linspace(0, N - 1, n_windows_c + 2).astype(int)[1:-1]
        t0 = time.time()
        c_scan(L, m.R, As, gen, m.row, gen[idx], np.zeros(len(idx), np.int64), np.full(len(idx), N - 1, np.int64))
        dt = time.time() - t0
        out['c_port'] = {'value': len(idx) / dt, 'unit': 'windows/s', 'cores': cores,
                         'sample': '%d windows, oracle/bmx_oracle.c orc_scan (log1p/LUT form, OpenMP), %.1f s' % (len(idx), dt)}
    except Exception as e:  # the C leg is optional
        out['c_port'] = {'error': str(e)}
    return out


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
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--snps', type=int, default=1000000, help='SNPs (= windows) per GPU per step')
    ap.add_argument('--n', type=int, default=100)
    ap.add_argument('--n-spread', type=int, default=0,
                    help='>0: sample sizes n-spread..n drawn per site (missing data): LUT rows = sum(n_i+1), which '
                         'can exceed LDS and exercises the R-from-L2 path')
    ap.add_argument('--variant', type=int, default=0)
    ap.add_argument('--config', type=int, default=3, choices=[3, 5],
                    help='3 (default): n=100, default 31x10x51 grid; 5: the dense-grid stress of BASELINE config 5 '
                         '(n=200, A=100..10000 step 100, --findBal --findPos grid) on one chromosome per GPU')
    ap.add_argument('--step', type=int, default=1, help='test site = every step-th SNP (the reference\'s -s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-windows', type=int, default=48)
    args = ap.parse_args()

    import torch
    from ballermixplus_amd import distributed, engine, synth
    from ballermixplus_amd.hostmodel import Grids

    # BMX_DIST_BACKEND=gloo + BMX_SINGLE_DEVICE=1: several ranks sharing ONE GPU with a CPU gather -- only to
    # rehearse the multi-rank control flow (barriers, max-over-ranks, rank-0 output) on a 1-GPU box
    world = distributed.World.from_env(backend=os.environ.get('BMX_DIST_BACKEND'))
    if world.size != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world.size, args.gpus))
    rank, dev = world.rank, world.local_rank
    if os.environ.get('BMX_SINGLE_DEVICE') == '1':
        dev = 0
    torch.cuda.set_device(dev)
    on_gpu = world.backend != 'gloo'
    cdev = torch.device('cuda', dev) if on_gpu else torch.device('cpu')

    N, n = args.snps, args.n
    if args.config == 5:
        n = 200
    phys, gen, k, nn = synth.synth_chromosome(N, n, chrom=rank + 1)
    sizes = [n]
    if args.n_spread > 0:      # thin the sample sizes: n_i uniform in [n - spread, n], counts rescaled
        rng = np.random.default_rng(77 + rank)
        n2 = rng.integers(n - args.n_spread, n + 1, N)
        k = np.where(k == nn, n2, np.maximum(1, np.minimum(n2 - 1, (k * n2) // nn)))
        nn = n2
        sizes = sorted(set(nn.tolist()))
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}
    props = {int(s_): float(sum(f for (a, b), f in spect.items() if b == s_)) for s_ in sizes}
    if args.config == 5:     # --rangeA 100,10000,100 --findBal --findPos  (findBal wins: 10 x, 44 alpha)
        grid = Grids(None, None, True, True, '100,10000,100', None)
    else:
        grid = Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    model = engine.ModelArrays('B2', int(k.min()), sizes, spect, props, xs, ab)
    ctx = engine.Context(dev)
    ctx.set_variant(args.variant)
    ctx.set_model(model, As)
    ctx.set_sites(gen, model.rows_of(k, nn))
    tidx = np.arange(0, N, args.step)
    ctx.set_tests(gen[tidx], np.zeros(len(tidx), np.int64), np.full(len(tidx), N - 1, np.int64))   # resident in HBM from here on

    from ballermixplus_amd import _lib
    zcut = _lib.lib().bmx_alpha_cut()
    wsum, wmax = window_work(gen, As, zcut, tidx)
    evals_per_step = float(wsum.sum()) * len(xs) * len(ab)
    M = len(tidx)
    bytes_per_step = float(wmax.sum()) * 10.0 + 24.0 * M      # W_max*10 B + 24 B per window

    def gather():
        if world.distributed:
            if on_gpu:      # zero-copy views of the library's device buffers -> RCCL
                pc, pl, pn = ctx.result_ptrs()
                clr = torch.as_tensor(distributed._DevArray(pc, M, '<f8'), device=cdev)
                lin = torch.as_tensor(distributed._DevArray(pl, M, '<i4'), device=cdev)
                ns = torch.as_tensor(distributed._DevArray(pn, M, '<i4'), device=cdev)
            else:           # rehearsal mode: through the host
                c_, ix_, ia_, iA_, n_ = ctx.fetch()
                clr, lin, ns = torch.from_numpy(c_), torch.from_numpy(iA_.copy()), torch.from_numpy(n_)
            outs = []
            for t in (clr, lin, ns):
                buf = torch.empty(M * world.size, dtype=t.dtype, device=cdev)
                torch.distributed.all_gather_into_tensor(buf, t)
                outs.append(buf)
            return outs
        return None

    def barrier():
        if world.distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.scan()
        ctx.sync()
        gather()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.scan()
        ctx.sync()                       # results must be complete before the gather reads them
        kernel_ms.append(ctx.last_scan_ms())
        gather()
    barrier()
    dt = time.perf_counter() - t0
    if world.distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    clr, ix, ia, iA, ns = ctx.fetch()
    checksum = float(np.sum(clr))

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        windows = float(M) * world.size * args.steps
        evals_s = evals_per_step / (k_ms * 1e-3)
        res = {
            'metric': 'CLR windows/sec (B2 scan, n=%d)' % n,
            'value': windows / dt,
            'unit': 'windows/s',
            'n_gpus': world.size, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('BASELINE config 3: synthetic single chromosome, %d SNPs, n=%d, default '
                                    '31x10x51 (A,x,alpha) grid, B2 scan, every SNP a test site; one chromosome '
                                    'per GPU' % (N, n)) if args.config == 3 else
                                   ('BASELINE config 5 grid on one chromosome per GPU: %d SNPs, n=%d, A=100..10000 step 100, '
                                    '--findBal --findPos (100x10x44), every SNP a test site' % (N, n)),
                       'windows_per_step_per_gpu': M, 'grid_points': len(As) * len(xs) * len(ab),
                       'parallelism': 'test-site sharding, dp%d, RCCL all_gather of 16-B records per step' % world.size,
                       'checksum_clr_rank0': checksum},
            'roofline': {
                'bound': 'valu_fp64', 'kernel': 'clr_scan_grouped_kernel<16,true,3>',
                'achieved': evals_s * FLOP_PER_EVAL / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': evals_s * FLOP_PER_EVAL / 1e12 / FP64_VALU_PEAK_TFLOPS, 'traffic': None,
                'kernel_ms': k_ms, 'evals_per_launch': evals_per_step, 'evals_per_s': evals_s,
                'valu_insts_per_64_evals': VALU_PER_64_EVALS,
                'valu_calibration': 'SQ_INSTS_VALU, config 3, profiles/r01_pmc_scan_kernel_200k_windows.txt'
                                    + ('' if args.config == 3 else ' (this workload is not calibrated separately)'),
                'survey_convention_tflops': evals_s * SURVEY_FLOP_PER_EVAL / 1e12,
                'note': 'achieved = executed VALU wave-instructions x 128 flop / kernel time: the share of the vector '
                        'issue slots the kernel fills (peak = vector FP64; there is no contraction in this path, so '
                        'not MFMA).  Algorithmic work: evals_per_s mixture-likelihood evaluations; under SURVEY 8(d) '
                        '"1 evaluation = 32 flop" that is survey_convention_tflops, above the peak because near '
                        'sites are multiplied four per step in product form and far sites are summed as moments'},
            'roofline_hbm': {
                'bound': 'hbm', 'kernel': 'clr_scan_grouped_kernel<16,true,3>',
                'achieved': bytes_per_step / (k_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': bytes_per_step / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                'algorithmic_bytes_per_launch': bytes_per_step},
        }
        if not args.no_cpu_baseline and world.size == 1:      # reported baseline: rank 0, N = 1 only
            res['cpu_baseline'] = cpu_baseline(gen, k, nn, spect, props, grid, args.cpu_windows, 256)
        line = json.dumps(res)
    else:
        line = None
    ctx.close()
    world.finish()
    return line


if __name__ == '__main__':
    main()
