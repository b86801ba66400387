"""GPU tests added in round 2: the one-shot C ABI driven with raw ctypes exactly as INTEGRATION.md binds it, the
streaming writer, the 16-byte records, the result-state guard, the multi-rank CLI on one GPU, and BASELINE
configs 4 and 5 at their full per-chromosome sizes (whole-genome run on one context, reference windows on the
chromosomes the reference could afford)."""
import ctypes as C
import functools
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
from util import GOLD, REFT, REPO, read_tsv

pytestmark = pytest.mark.gpu

LIB = os.path.join(REPO, 'ballermixplus_amd', 'libbmxscan.so')


# ------------------------------------------------------------------------------------------------ one-shot ABI
class _Model(C.Structure):          # include/bmxscan.h: typedef struct bmx_model
    _fields_ = [('stat', C.c_int32), ('min_count', C.c_int32), ('n_sizes', C.c_int32),
                ('sizes', C.POINTER(C.c_int32)), ('row_off', C.POINTER(C.c_int32)),
                ('g', C.POINTER(C.c_double)), ('prop', C.POINTER(C.c_double)),
                ('nx', C.c_int32), ('x', C.POINTER(C.c_double)), ('nab', C.c_int32), ('abeta', C.POINTER(C.c_double))]


def _raw_lib():
    """The binding of INTEGRATION.md, written out: nothing from ballermixplus_amd.engine / _lib."""
    L = C.CDLL(LIB)
    dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.bmx_lut_build.restype = C.c_int
    L.bmx_lut_build.argtypes = [C.POINTER(_Model), dp, dp, C.c_int]
    L.bmx_scan.restype = C.c_int
    L.bmx_scan.argtypes = [C.POINTER(_Model), dp, C.c_int32, C.c_int64, dp, ip, C.c_int64, dp, lp, lp, dp, ip, ip, ip, ip, C.c_int]
    L.bmx_last_error.restype = C.c_char_p
    return L


def _arr(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    ct = {np.float64: C.c_double, np.int32: C.c_int32, np.int64: C.c_int64}[dt]
    return a, a.ctypes.data_as(C.POINTER(ct))


def _model_by_hand(stat_id, min_count, sizes, g_of, prop_of, xs, ab):
    sizes_a, sizes_p = _arr(sorted(sizes), np.int32)
    off = np.concatenate(([0], np.cumsum([n + 1 for n in sizes_a])))
    off_a, off_p = _arr(off, np.int32)
    g = np.full(int(off[-1]), np.nan)
    for j, n in enumerate(sizes_a.tolist()):
        for k in range(n + 1):
            v = g_of(k, n)
            if v is not None:
                g[off[j] + k] = v
    g_a, g_p = _arr(g, np.float64)
    pr_a, pr_p = _arr([prop_of(int(n)) for n in sizes_a], np.float64)
    x_a, x_p = _arr(xs, np.float64)
    ab_a, ab_p = _arr(ab, np.float64)
    m = _Model(stat_id, min_count, len(sizes_a), sizes_p, off_p, g_p, pr_p, len(x_a), x_p, len(ab_a), ab_p)
    m._keep = (sizes_a, off_a, g_a, pr_a, x_a, ab_a)
    return m, off


def test_one_shot_lut_build_against_reference_table():
    """bmx_lut_build (the binding of NormalizedBetaBinom(...), BalLeRMix+_v1.py:793) with a hand-filled bmx_model,
    against the reference's normProbs for B_2, n = 50, default grid (tests/golden/lut_B2_n50_default_min1.npz)."""
    L = _raw_lib()
    z = np.load(os.path.join(GOLD, 'lut_B2_n50_default_min1.npz'))
    sizes = sorted(set(z['total'].tolist()))
    m, off = _model_by_hand(0, int(z['minCount']), sizes, lambda k, n: 1.0, lambda n: 1.0, z['x'], z['abeta'])
    rows = int(off[-1])
    psel = np.empty((len(z['x']), len(z['abeta']), rows))
    R = np.empty_like(psel)
    rc = L.bmx_lut_build(C.byref(m), psel.ctypes.data_as(C.POINTER(C.c_double)), R.ctypes.data_as(C.POINTER(C.c_double)), 0)
    assert rc == 0, L.bmx_last_error()
    row_of = {n: int(o) for n, o in zip(sizes, off[:-1])}
    r = np.array([row_of[int(n)] + int(k) for k, n in zip(z['count'], z['total'])])
    rel = np.abs(psel[:, :, r] - z['table']) / np.abs(z['table'])
    assert np.nanmax(rel) < 1e-12
    assert np.allclose(R[:, :, r], psel[:, :, r] - 1.0, rtol=0, atol=1e-13)       # g = prop = 1: R = P_sel - 1
    # psel_out / R_out are optional
    assert L.bmx_lut_build(C.byref(m), None, R.ctypes.data_as(C.POINTER(C.c_double)), 0) == 0
    assert L.bmx_lut_build(C.byref(m), None, None, 99) == -2 and b'device' in L.bmx_last_error()     # BMX_E_NODEVICE


def test_one_shot_scan_against_reference_output():
    """bmx_scan (the binding of the calcBaller loop, BalLeRMix+_v1.py:606) with host buffers prepared by hand from the
    reference's Example 1 input and helper file, against test/output/Example1_B2.txt (every 4th row); then the ABI's
    error codes."""
    L = _raw_lib()
    rows_in = [l.split('\t') for l in open(os.path.join(REFT, 'Example1_fullSweep_200kya_DAF.txt')).read().splitlines()[1:] if l]
    gen = np.array([float(r[1]) for r in rows_in])
    k = np.array([int(r[2]) for r in rows_in])
    nn = np.array([int(r[3]) for r in rows_in])
    spect = {}
    for l in open(os.path.join(REFT, 'HC_CEU_Neut_DAF_spect_for_B2.txt')).read().splitlines():
        f = l.split('\t')
        if len(f) == 3 and f[0].strip().isdigit():
            spect[(int(f[0]), int(f[1]))] = float(f[2])
    sizes = sorted(set(nn.tolist()))
    prop = {n: sum(v for (kk, n2), v in spect.items() if n2 == n) for n in sizes}
    from util import load_json
    order = load_json('setorder.json')['default']          # list(set(Grids.*)) as the reference iterates them
    sx, sab, sA = order['x'], order['abeta'], order['A']       # the grids' printed forms, in iteration order
    xs, ab, As = [float(v) for v in sx], [float(v) for v in sab], [float(v) for v in sA]
    m, off = _model_by_hand(0, int(k.min()), sizes, lambda kk, n: spect.get((kk, n)), lambda n: prop[n], xs, ab)
    row_of = {n: int(o) for n, o in zip(sizes, off[:-1])}
    row_a, row_p = _arr([row_of[int(n)] + int(kk) for kk, n in zip(k, nn)], np.int32)
    gen_a, gen_p = _arr(gen, np.float64)
    A_a, A_p = _arr(As, np.float64)
    N = len(gen)
    idx = np.arange(0, N, 4)
    t_a, t_p = _arr(gen[idx], np.float64)
    lo_a, lo_p = _arr(np.zeros(len(idx)), np.int64)
    hi_a, hi_p = _arr(np.full(len(idx), N - 1), np.int64)
    M = len(idx)
    clr = np.empty(M)
    ix, ia, iA, ns = (np.empty(M, np.int32) for _ in range(4))
    ptr = lambda a, ct: a.ctypes.data_as(C.POINTER(ct))
    call = lambda g_p_, t_p_, dev: L.bmx_scan(C.byref(m), A_p, len(A_a), N, g_p_, row_p, M, t_p_, lo_p, hi_p, ptr(clr, C.c_double),
                                              ptr(ix, C.c_int32), ptr(ia, C.c_int32), ptr(iA, C.c_int32), ptr(ns, C.c_int32), dev)
    assert call(gen_p, t_p, 0) == 0, L.bmx_last_error()
    gold = read_tsv(os.path.join(REFT, 'output', 'Example1_B2.txt'))[::4]
    for j, r in enumerate(gold):
        assert abs(clr[j] - float(r[2])) <= 1e-6 * abs(float(r[2])), (j, clr[j], r)
        assert (sx[ix[j]], sab[ia[j]], sA[iA[j]], str(ns[j])) == tuple(r[3:7]), (j, r)
    # error codes: unsorted positions -> BMX_E_INVALID; device out of range -> BMX_E_NODEVICE; NULL model -> BMX_E_INVALID
    bad_a, bad_p = _arr(gen[::-1], np.float64)
    assert call(bad_p, t_p, 0) == -1 and b'non-decreasing' in L.bmx_last_error()
    assert call(gen_p, t_p, 99) == -2
    assert L.bmx_scan(None, A_p, len(A_a), N, gen_p, row_p, M, t_p, lo_p, hi_p, ptr(clr, C.c_double), ptr(ix, C.c_int32),
                      ptr(ia, C.c_int32), ptr(iA, C.c_int32), ptr(ns, C.c_int32), 0) == -1


# ------------------------------------------------------------------------------------------------ streaming, records, state
def _config3_ctx(N=300000, n=100, chrom=3):
    from ballermixplus_amd import engine as eng, synth
    from ballermixplus_amd.hostmodel import Grids
    phys, gen, k, nn = synth.synth_chromosome(N, n, chrom)
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}
    grid = Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    model = eng.ModelArrays('B2', int(k.min()), [n], spect, {n: 1.0}, xs, ab)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    ctx.set_sites(gen, model.rows_of(k, nn))
    return ctx, phys, gen, (xs, ab, As)


@pytest.mark.parametrize('M,chunk', [(70001, 0), (70001, 1000), (5000, 4096), (130, 64)])
def test_streamed_rows_equal_batch_rows(tmp_path, M, chunk):
    """bmx_ctx_scan_write (chunks scanned while a host thread writes the previous chunk's rows) produces the bytes of
    scan + fetch + bmx_write_rows, and leaves the same results in the context (bitwise: chunks end where workgroups do)."""
    from ballermixplus_amd import _lib
    ctx, phys, gen, (xs, ab, As) = _config3_ctx()
    N = len(gen)
    idx = 100000 + np.arange(M)
    lo, hi = np.zeros(M, np.int64), np.full(M, N - 1, np.int64)
    ctx.set_tests(gen[idx], lo, hi)
    ctx.scan()
    base = ctx.fetch()
    sx, sa, sA = [f'{v}' for v in xs], [f'{v}' for v in ab], [f'{v}' for v in As]
    f1, f2 = str(tmp_path / 'batch.txt'), str(tmp_path / 'stream.txt')
    _lib.write_rows(f1, phys[idx], gen[idx], *base, sx, sa, sA)
    ctx.set_tests(gen[idx], lo, hi)
    ctx.scan_write(f2, phys[idx], gen[idx], sx, sa, sA, chunk=chunk)
    assert open(f1, 'rb').read() == open(f2, 'rb').read()
    again = ctx.fetch()
    assert all(np.array_equal(a, b) for a, b in zip(base, again))
    rec = ctx.fetch_records()                       # the 16-byte records hold the same three values
    npairs = len(xs) * len(ab)
    lin = np.where(base[3] < 0, -1, base[3] * npairs + base[1] * len(ab) + base[2])
    assert np.array_equal(rec['clr'], base[0]) and np.array_equal(rec['lin'], lin) and np.array_equal(rec['nsites'], base[4])
    ctx.close()


def test_results_belong_to_the_test_sites_they_were_computed_for():
    """After set_tests (or set_sites / set_model) the previous scan's results are gone: fetch, result_ptrs, records and
    last_scan_ms refuse with BMX_E_STATE until a new scan has been launched."""
    from ballermixplus_amd import _lib
    ctx, phys, gen, _ = _config3_ctx(N=50000)
    N = len(gen)
    idx = np.arange(1000, 1256)
    ctx.set_tests(gen[idx], np.zeros(256, np.int64), np.full(256, N - 1, np.int64))
    for call in (ctx.fetch, ctx.result_ptrs, ctx.records, ctx.fetch_records, ctx.last_scan_ms):
        with pytest.raises(_lib.BmxError) as e:
            call()
        assert e.value.code == -5
    ctx.scan()
    first = ctx.fetch()
    ctx.set_tests(gen[idx + 7], np.zeros(256, np.int64), np.full(256, N - 1, np.int64))
    with pytest.raises(_lib.BmxError) as e:
        ctx.fetch()
    assert e.value.code == -5
    ctx.scan()
    assert not np.array_equal(ctx.fetch()[0], first[0])
    ctx.close()


def test_context_reuse_across_chromosomes_is_bitwise_neutral():
    """One context fed chromosome after chromosome (buffers only grow, never reallocated smaller) returns what a fresh
    context returns -- larger, smaller, larger again."""
    from ballermixplus_amd import engine as eng, synth
    from ballermixplus_amd.hostmodel import Grids
    grid = Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    data = [synth.synth_chromosome(N, 100, c) for c, N in ((5, 90000), (6, 30000), (7, 120000))]
    kk, nk = np.concatenate([d[2] for d in data]), np.concatenate([d[3] for d in data])
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(kk, nk)}
    model = eng.ModelArrays('B2', int(kk.min()), [100], spect, {100: 1.0}, xs, ab)

    def run(ctx, d, M):
        phys, gen, k, nn = d
        ctx.set_sites(gen, model.rows_of(k, nn))
        idx = np.arange(len(gen) // 3, len(gen) // 3 + M)
        ctx.set_tests(gen[idx], np.zeros(M, np.int64), np.full(M, len(gen) - 1, np.int64))
        ctx.scan()
        return ctx.fetch()

    shared = eng.Context(0)
    shared.set_model(model, As)
    for d, M in zip(data, (20000, 3000, 33000)):
        got = run(shared, d, M)
        fresh = eng.Context(0)
        fresh.set_model(model, As)
        want = run(fresh, d, M)
        fresh.close()
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    shared.close()


# ------------------------------------------------------------------------------------------------ multi-rank CLI on one GPU
def _cli(args, env=None, nproc=0, port=29641):
    exe = [sys.executable]
    if nproc:
        exe += ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
                '--master-port', str(port)]
    r = subprocess.run(exe + [os.path.join(REPO, 'BalLeRMixPlus_amd.py')] + args, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r


def test_cli_two_ranks_on_one_gpu_write_the_single_rank_file(tmp_path):
    """The drop-in CLI under torch.distributed.run with two ranks that both compute on this box's one GPU and gather
    their 16-byte records on rank 0 through gloo: the output file is byte-identical to the single-process file --
    BASELINE config 2 (Example 2, --MAF --findBal; shards of 64 test sites so that both ranks scan) and a 64k-site
    synthetic chromosome (blocks of 4096).  Only rank 0 talks."""
    from ballermixplus_amd import helpers, synth
    argv, gold = cases.ALL_CASES['ex2_B2maf_findBal']
    one, two = tmp_path / 'one.txt', tmp_path / 'two.txt'
    _cli(argv + ['-o', str(one)])
    env = {'BMX_DIST_BACKEND': 'gloo', 'BMX_SINGLE_DEVICE': '1', 'BMX_SHARD_BLOCK': '64'}
    r = _cli(argv + ['-o', str(two)], env=env, nproc=2)
    assert one.read_bytes() == two.read_bytes()
    assert r.stdout.count('Pipeline finished.') == 1 and r.stdout.count('writing output to') == 1
    worst, ties = 0.0, 0
    got = two.read_text().splitlines()[1:]
    opt, case, ts = cases.host_side(argv)
    worst, ties = cases.compare_rows(got, gold, rtol=1e-6, case=case, ts=ts)       # ... and it is the reference's file
    assert worst < 1e-6 and ties == 0
    # 64k synthetic sites, default block size: 16 blocks dealt 8/8
    phys, gen, k, nn = synth.synth_chromosome(65536, 100, 9)
    inp, sp = tmp_path / 'in.txt', tmp_path / 'spect.txt'
    synth.write_input(str(inp), phys, gen, k, nn)
    helpers.getSpect(str(inp), str(sp), False, False)
    a, b = tmp_path / 's1.txt', tmp_path / 's2.txt'
    _cli(['-i', str(inp), '--spect', str(sp), '-o', str(a)])
    _cli(['-i', str(inp), '--spect', str(sp), '-o', str(b)], env={'BMX_DIST_BACKEND': 'gloo', 'BMX_SINGLE_DEVICE': '1'}, nproc=2, port=29643)
    assert a.read_bytes() == b.read_bytes() and a.stat().st_size > 3 << 20
    # --device is refused under a multi-process launch
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29645', os.path.join(REPO, 'BalLeRMixPlus_amd.py'), '-i', str(inp), '--spect', str(sp),
                        '-o', str(tmp_path / 'x.txt'), '--device', '0'], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, BMX_DIST_BACKEND='gloo', BMX_SINGLE_DEVICE='1'))
    assert r.returncode != 0 and '--device cannot be combined' in r.stdout + r.stderr


def test_bench_contract_two_ranks_on_one_gpu():
    """PLAIN `python bench.py --gpus 2 ...` -- no launcher: the parent, which never touches the GPU, starts the two ranks itself
    (torch.distributed.run as a child) and relays their one JSON line.  Default workload (config 4, here a 1/100 genome), two
    ranks sharing this box's one GPU (gloo gather through the host instead of RCCL): exit code 0, exactly ONE JSON line on
    stdout, whole-job value over both ranks, strong scaling, ONE gather inside the step; same checksum and same number of
    records as the one-rank run, step time within 25 % of it (both ranks' kernels share the one GPU)."""
    import json
    env = dict(os.environ, BMX_DIST_BACKEND='gloo', BMX_SINGLE_DEVICE='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    common = ['--steps', '2', '--warmup', '1', '--total-snps', '400000', '--no-cpu-baseline']
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2'] + common,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'strong'
    assert d['unit'] == 'windows/s' and d['higher_is_better'] is True and d['vs_baseline'] is None
    W = d['config']['windows_per_step']
    assert 399990 <= W <= 400010 and d['config']['launches_per_step'] == 22 and d['config']['records_per_step'] == W
    assert abs(d['value'] - W / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert 'ONE gather' in d['config']['parallelism']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in d['roofline']
    assert 'cpu_baseline' not in d            # reported at N = 1 only
    r1 = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + common, capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.strip()][-1])
    assert d1['n_gpus'] == 1 and d1['config']['records_per_step'] == W
    assert d1['config']['checksum_clr'] == pytest.approx(d['config']['checksum_clr'], rel=1e-12)    # same rows, summed in another order
    assert d1['config']['input_sha256'] == d['config']['input_sha256']
    print('bench step: 2 ranks on one GPU %.1f ms, 1 rank %.1f ms (ratio %.3f)' % (d['ms_per_step'], d1['ms_per_step'], d['ms_per_step'] / d1['ms_per_step']))
    assert d['ms_per_step'] < 1.25 * d1['ms_per_step']


# ------------------------------------------------------------------------------------------------ configs 4 and 5 at size
@functools.lru_cache(maxsize=1)
def _config4_data():
    from ballermixplus_amd import synth
    sizes = synth.config4_sizes(40_000_000)
    data = [synth.synth_chromosome(N, 100, c + 1) for c, N in enumerate(sizes)]
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(np.concatenate([d[2] for d in data]), np.concatenate([d[3] for d in data]))}
    return sizes, data, spect


def _check_reference_rows(path, got, phys_of_idx, xs, ab, As):
    """rows of a reference run made with -s: exact (x, alpha, A, nSites), CLR to 1e-6"""
    gold = read_tsv(path)
    clr, ix, ia, iA, ns = got
    assert len(gold) == len(clr)
    worst = 0.0
    for j, r in enumerate(gold):
        assert int(r[0]) == int(phys_of_idx[j])
        if r[3:] == ['0.0'] * 4 or iA[j] < 0:
            assert iA[j] < 0 and r[3:] == ['0.0'] * 4, (j, r)
            continue
        assert (repr(xs[ix[j]]), repr(ab[ia[j]]), repr(As[iA[j]]), str(ns[j])) == tuple(r[3:7]), (j, r, clr[j])
        d = abs(clr[j] - float(r[2])) / max(abs(float(r[2])), 1e-9)
        assert abs(clr[j] - float(r[2])) <= max(1e-9, 1e-6 * abs(float(r[2]))), (j, r, clr[j])
        worst = max(worst, d)
    return worst


def test_config4_whole_genome_on_one_context():
    """BASELINE config 4 in full on one GPU: 40M SNPs, 22 chromosomes (up to 3.46M sites), n = 100, default grid, every
    SNP a test site, ONE context fed chromosome after chromosome (set_sites -> set_tests -> scan -> fetch).  Checked:
    (i) chromosomes 1, 12 and 22 equal a fresh context's run bitwise; (ii) every chromosome: all windows scanned, CLR
    finite and non-negative, nSites within the window bound, a checksum over (CLR, argmax) that a second pass reproduces
    for the three chromosomes of (i); (iii) the reference's own rows for the chromosomes it could afford
    (tests/golden/synth/config4_chr{21,22}_step50000.tsv, made with the helper file of the whole genome); (iv) chromosome 22 in
    full through the round-2 kernel as well."""
    from ballermixplus_amd import engine as eng
    from ballermixplus_amd.hostmodel import Grids
    sizes, data, spect = _config4_data()
    grid = Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    model = eng.ModelArrays('B2', 1, [100], spect, {100: 1.0}, xs, ab)
    shared = eng.Context(0)
    shared.set_model(model, As)
    total, kernel_ms = 0, 0.0
    keep = {}
    for c, (phys, gen, k, nn) in enumerate(data, start=1):
        N = len(gen)
        shared.set_sites(gen, model.rows_of(k, nn))
        shared.set_tests(gen, np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
        shared.scan()
        clr, ix, ia, iA, ns = shared.fetch()
        kernel_ms += shared.last_scan_ms()
        total += N
        assert len(clr) == N and np.all(np.isfinite(clr)) and np.all(clr >= 0) and np.all((iA >= 0) == (clr > 0))
        assert ns.max() <= 2 * 2700 and np.all(ns[iA >= 0] > 0)      # A = 100: +-18.42/100 genetic units at ~13.8k sites per unit
        if c in (1, 12, 22):
            keep[c] = (clr.copy(), ix.copy(), ia.copy(), iA.copy(), ns.copy())
        for cc in (21, 22):
            path = os.path.join(GOLD, 'synth', 'config4_chr%d_step50000.tsv' % cc)
            if c == cc and os.path.exists(path):
                idx = np.arange(0, N, 50000)
                worst = _check_reference_rows(path, tuple(a[idx] for a in (clr, ix, ia, iA, ns)), phys[idx], xs, ab, As)
                assert worst < 1e-6
    assert total == 39999999
    print('config 4 on one context: %d windows, scan kernels %.2f s = %.3f M windows/s' % (total, kernel_ms / 1e3, total / kernel_ms / 1e3))
    shared.close()
    for c, want in keep.items():
        phys, gen, k, nn = data[c - 1]
        N = len(gen)
        fresh = eng.Context(0)
        fresh.set_model(model, As)
        fresh.set_sites(gen, model.rows_of(k, nn))
        fresh.set_tests(gen, np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
        fresh.scan()
        got = fresh.fetch()
        if c == 22:
            # (iv) the whole chromosome through the round-2 single-kernel form (variant 12): the two pipelines agree on the
            # argmax and nSites of every one of its 712 415 windows, CLR to 1e-9 (+ 1e-13 absolute)
            fresh.set_variant(12)
            fresh.set_tests(gen, np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
            fresh.scan()
            old = fresh.fetch()
            assert all(np.array_equal(a, b) for a, b in zip(got[1:], old[1:]))
            # (|dCLR| <= 1e-9 |CLR| + 1e-13: where the best CLR of a window is ~1e-7 -- two sites at A = 1e6 -- one ulp of the sum is
            # already 6e-10 of it)
            assert np.all(np.abs(got[0] - old[0]) <= 1e-9 * np.abs(old[0]) + 1e-13)
        fresh.close()
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), c
    for cc in (21, 22):
        assert os.path.exists(os.path.join(GOLD, 'synth', 'config4_chr%d_step50000.tsv' % cc)), 'reference fixture missing'


def test_config5_in_full():
    """BASELINE config 5 in full: 10M SNPs as 8 contigs of 1.25M, n = 200, A = 100..10000 step 100, --findBal --findPos
    (100 x 10 x 44), every SNP a test site, helper file from the concatenation of all 8 contigs -- ONE context, one slot per
    contig, the eight scans launched back to back, one pack of all 10M records.  Reference rows
    (tests/golden/synth/config5_contig{1,2}_step125000.tsv) exact in (x, alpha, A, nSites), CLR to 1e-6; a block of 4096
    consecutive windows on contigs 1 and 2 equals the per-site kernel's (variant 2) argmax and nSites; contigs 5 and 8 are
    bitwise equal to a fresh context's."""
    from ballermixplus_amd import engine as eng, synth
    from ballermixplus_amd.hostmodel import Grids
    data = [synth.synth_chromosome(1250000, 200, c + 1) for c in range(8)]
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(np.concatenate([d[2] for d in data]), np.concatenate([d[3] for d in data]))}
    grid = Grids(None, None, True, True, '100,10000,100', None)
    xs, ab, As = grid.scan_order()
    assert (len(As), len(xs), len(ab)) == (100, 10, 44)
    model = eng.ModelArrays('B2', 1, [200], spect, {200: 1.0}, xs, ab)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    for c in range(8):
        phys, gen, k, nn = data[c]
        N = len(gen)
        ctx.select_slot(c)
        ctx.set_sites(gen, model.rows_of(k, nn))
        ctx.set_tests(gen, np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
    for c in range(8):
        ctx.select_slot(c)
        ctx.scan()
    rec = ctx.pack_records()
    assert len(rec) == 10000000 and np.all(np.isfinite(rec['clr'])) and np.all(rec['clr'] >= 0)
    print('config 5 in full: %d windows scanned on one context (8 slots)' % len(rec))
    npairs = len(xs) * len(ab)
    found = 0
    for c in (1, 2):
        phys, gen, k, nn = data[c - 1]
        N = len(gen)
        ctx.select_slot(c - 1)
        clr, ix, ia, iA, ns = ctx.fetch()
        r = rec[(c - 1) * N:c * N]
        lin = np.where(iA < 0, -1, iA * npairs + ix * len(ab) + ia)
        assert np.array_equal(r['clr'], clr) and np.array_equal(r['lin'], lin) and np.array_equal(r['nsites'], ns)
        path = os.path.join(GOLD, 'synth', 'config5_contig%d_step125000.tsv' % c)
        if os.path.exists(path):
            found += 1
            idx = np.arange(0, N, 125000)
            assert _check_reference_rows(path, tuple(a[idx] for a in (clr, ix, ia, iA, ns)), phys[idx], xs, ab, As) < 1e-6
        blk = np.arange(600000, 600000 + 4096)
        ctx.set_variant(2)
        ctx.set_tests(gen[blk], np.zeros(4096, np.int64), np.full(4096, N - 1, np.int64))
        ctx.scan()
        c2, ix2, ia2, iA2, ns2 = ctx.fetch()
        ctx.set_variant(0)
        assert np.array_equal(ix2, ix[blk]) and np.array_equal(ia2, ia[blk]) and np.array_equal(iA2, iA[blk]) and np.array_equal(ns2, ns[blk])
        assert np.max(np.abs(c2 - clr[blk]) / np.maximum(clr[blk], 1e-9)) < 1e-9
    for c in (5, 8):
        phys, gen, k, nn = data[c - 1]
        N = len(gen)
        fresh = eng.Context(0)
        fresh.set_model(model, As)
        fresh.set_sites(gen, model.rows_of(k, nn))
        fresh.set_tests(gen, np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
        fresh.scan()
        want = fresh.fetch_records()
        fresh.close()
        assert np.array_equal(rec[(c - 1) * N:c * N], want), c
    ctx.close()
    assert found == 2, 'reference fixtures missing'


def test_more_test_sites_than_one_launch_covers():
    """A scan is launched in ranges of at most 4 194 304 test sites (512 MB of per-slice winners at 8 slices): 4.4M windows on
    one chromosome take two launches, and the windows on either side of the cut equal a separate scan of just those test sites
    bitwise (ranges end where workgroups do); the streaming writer's chunks obey the same rule."""
    from ballermixplus_amd import engine as eng, synth
    from ballermixplus_amd.hostmodel import Grids
    N = 4400000
    phys, gen, k, nn = synth.synth_chromosome(N, 100, 13)
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}
    grid = Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    model = eng.ModelArrays('B2', int(k.min()), [100], spect, {100: 1.0}, xs, ab)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    ctx.set_sites(gen, model.rows_of(k, nn))
    ctx.set_tests(gen, np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
    ctx.scan()
    full = ctx.fetch()
    assert np.all(np.isfinite(full[0])) and np.all(full[0] >= 0)
    for lo in (4194304 - 2048, N - 4096 - (N - 4096) % 16, 0):
        idx = np.arange(lo, lo + 4096)
        ctx.set_tests(gen[idx], np.zeros(4096, np.int64), np.full(4096, N - 1, np.int64))
        ctx.scan()
        part = ctx.fetch()
        assert all(np.array_equal(a[idx], b) for a, b in zip(full, part)), lo
    ctx.close()
