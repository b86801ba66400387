"""GPU tests added in round 3: the prepared pipeline (prep_kernel + clr_scan_prepared_kernel) against the round-2 grouped
kernel, the per-site kernel and the reference's rows; chromosome slots of one context; the multi-file CLI form; the per-site
kernel on grids of more than 8191 A values; the plan query."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
from util import GOLD, REFT, REPO, c_oracle, c_scan, read_tsv

pytestmark = pytest.mark.gpu


def _synth_model(n=100, chroms=((1, 200000),), bal=False):
    from ballermixplus_amd import engine as eng, synth
    from ballermixplus_amd.hostmodel import Grids
    data = [synth.synth_chromosome(N, n, c) for c, N in chroms]
    kk, nk = np.concatenate([d[2] for d in data]), np.concatenate([d[3] for d in data])
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(kk, nk)}
    grid = Grids(None, None, True, True, '100,10000,100', None) if bal else Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    model = eng.ModelArrays('B2', int(kk.min()), [n], spect, {n: 1.0}, xs, ab)
    return eng, data, model, (xs, ab, As)


def test_plan_names_the_kernel_that_runs():
    """bmx_ctx_plan: dense test sites -> prepared pipeline, J = 16, table in LDS; every 8th or 10th SNP -> J = 8; every 11th (round 4: the
    measured crossover moved from a gap of 12 to 10), 13th or 200th -> one test site per wave, prepared (solo); variant 2 -> the
    round-2 per-site kernel; variant 12 -> the round-2 grouped kernel."""
    eng, data, model, (xs, ab, As) = _synth_model()
    phys, gen, k, nn = data[0]
    N = len(gen)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    ctx.set_sites(gen, model.rows_of(k, nn))
    want = {1: ('clr_scan_prepared_kernel<16,true>', 4), 3: ('clr_scan_prepared_kernel<16,true>', 4), 8: ('clr_scan_prepared_kernel<8,true>', 4),
            10: ('clr_scan_prepared_kernel<8,true>', 4), 11: ('clr_scan_solo_kernel<true>', 5), 13: ('clr_scan_solo_kernel<true>', 5),
            200: ('clr_scan_solo_kernel<true>', 5)}
    for step, (name, mode) in want.items():
        idx = np.arange(0, N, step)
        ctx.set_tests(gen[idx], np.zeros(len(idx), np.int64), np.full(len(idx), N - 1, np.int64))
        pl = ctx.plan()
        assert pl['kernel'] == name and pl['mode'] == mode, (step, pl)
        assert pl['stream_bytes'] > 0
    ctx.set_variant(2)
    ctx.set_tests(gen[idx], np.zeros(len(idx), np.int64), np.full(len(idx), N - 1, np.int64))
    pl = ctx.plan()
    assert pl['kernel'] == 'clr_scan_kernel<true>' and pl['mode'] == -1 and pl['stream_bytes'] == 0
    ctx.set_variant(12)
    idx = np.arange(0, N)
    ctx.set_tests(gen[idx], np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
    assert ctx.plan()['kernel'] == 'clr_scan_grouped_kernel<16,true,3>'
    ctx.close()


@pytest.mark.parametrize('step,J,v0,v12', [(1, 16, 0, 12), (2, 16, 0, 12), (5, 8, 0, 12), (30, 4, 15, 12)])
def test_prepared_pipeline_equals_round2_kernel_and_oracle(step, J, v0, v12):
    """The prepared pipeline (variant 0) against the round-2 grouped kernel (variant 12: same group size) on 32k config-3
    windows at several test-site strides: identical argmax and nSites, CLR to 1e-10 (the two differ only in which sites
    count as far: one threshold per row instead of one per row and slice); and against the C oracle on a sample."""
    eng, data, model, (xs, ab, As) = _synth_model(chroms=((1, 400000),))
    phys, gen, k, nn = data[0]
    N = len(gen)
    rows = model.rows_of(k, nn)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    ctx.set_sites(gen, rows)
    M = 32768 // max(1, step // 2)
    idx = 50000 + step * np.arange(M)
    idx = idx[idx < N - 1000]
    M = len(idx)
    lo, hi = np.zeros(M, np.int64), np.full(M, N - 1, np.int64)
    out = {}
    for key, v in ((0, v0), (12, v12)):      # (J = 4 is no longer what the default plan picks at any stride: variant 15 asks for it)
        ctx.set_variant(v)
        ctx.set_tests(gen[idx], lo, hi)
        pl = ctx.plan()
        assert pl['J'] == J and pl['mode'] == (4 if key == 0 else 3)
        ctx.scan()
        out[key] = ctx.fetch()
    for a, b in zip(out[0][1:], out[12][1:]):
        assert np.array_equal(a, b)
    assert np.max(np.abs(out[0][0] - out[12][0]) / np.maximum(np.abs(out[12][0]), 1e-9)) < 1e-10
    _, R = ctx.fetch_lut()
    samp = np.arange(0, M, max(1, M // 48))
    ref = c_scan(c_oracle(), np.where(np.isfinite(R), R, 0.0), As, gen, rows, gen[idx[samp]], lo[samp], hi[samp])
    for q in (1, 2, 3, 4):
        assert np.array_equal(out[0][q][samp], ref[q])
    assert np.allclose(out[0][0][samp], ref[0], rtol=1e-9, atol=1e-12)
    ctx.close()


def test_prepared_pipeline_windows_and_partial_groups():
    """Window modes through the prepared pipeline: -w index windows, windows that end inside the near field, a partial last group,
    and two scans of the same test sites are bitwise equal (the stream is rebuilt per scan)."""
    eng, data, model, (xs, ab, As) = _synth_model(chroms=((3, 120000),))
    phys, gen, k, nn = data[0]
    N = len(gen)
    rows = model.rows_of(k, nn)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    ctx.set_sites(gen, rows)
    _, R = ctx.fetch_lut()
    Rf = np.where(np.isfinite(R), R, 0.0)
    L = c_oracle()
    for r in (3, 40, 700, 30000):
        idx = np.arange(20000, 20000 + 1003)                 # 1003 = 62 groups of 16 + 11
        lo = np.maximum(idx - r, 0).astype(np.int64)
        hi = np.minimum(idx + r + 1, N - 1).astype(np.int64)
        ctx.set_tests(gen[idx], lo, hi)
        assert ctx.plan()['mode'] == 4
        ctx.scan()
        got = ctx.fetch()
        ctx.scan()
        again = ctx.fetch()
        assert all(np.array_equal(a, b) for a, b in zip(got, again))
        ref = c_scan(L, Rf, As, gen, rows, gen[idx], lo, hi)
        for q in (1, 2, 3, 4):
            assert np.array_equal(got[q], ref[q]), (r, q)
        assert np.allclose(got[0], ref[0], rtol=1e-9, atol=1e-12)
    ctx.close()


def test_slots_scanned_back_to_back_equal_separate_contexts():
    """One context, three chromosome slots (different lengths), scans launched back to back without waiting, ONE pack of all
    records -- bitwise what three separate contexts return; re-setting a slot's test sites drops only that slot's results; a new
    model drops every slot."""
    from ballermixplus_amd import _lib
    eng, data, model, (xs, ab, As) = _synth_model(chroms=((5, 90000), (6, 30000), (7, 120000)))
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    Ms = (40000, 30000, 50001)
    for c, (d, M) in enumerate(zip(data, Ms)):
        phys, gen, k, nn = d
        ctx.select_slot(c)
        ctx.set_sites(gen, model.rows_of(k, nn))
        ctx.set_tests(gen[:M], np.zeros(M, np.int64), np.full(M, len(gen) - 1, np.int64))
    for c in range(3):
        ctx.select_slot(c)
        ctx.scan()
    rec = ctx.pack_records()
    assert len(rec) == sum(Ms)
    at = 0
    for c, (d, M) in enumerate(zip(data, Ms)):
        phys, gen, k, nn = d
        one = eng.Context(0)
        one.set_model(model, As)
        one.set_sites(gen, model.rows_of(k, nn))
        one.set_tests(gen[:M], np.zeros(M, np.int64), np.full(M, len(gen) - 1, np.int64))
        one.scan()
        want = one.fetch_records()
        one.close()
        assert np.array_equal(rec[at:at + M], want), c
        ctx.select_slot(c)
        assert np.array_equal(ctx.fetch_records(), want)
        at += M
    # new test sites for slot 1: its results are gone, the others stay
    ctx.select_slot(1)
    ctx.set_tests(data[1][1][:100], np.zeros(100, np.int64), np.full(100, len(data[1][1]) - 1, np.int64))
    with pytest.raises(_lib.BmxError) as e:
        ctx.fetch_records()
    assert e.value.code == -5
    assert len(ctx.pack_records()) == Ms[0] + Ms[2]
    ctx.set_model(model, As)                  # a new model: every slot's sites and tests are dropped
    for c in range(3):
        ctx.select_slot(c)
        with pytest.raises(_lib.BmxError) as e:
            ctx.set_tests(data[c][1][:10], np.zeros(10, np.int64), np.full(10, 5, np.int64))
        assert e.value.code == -5
    ctx.close()


def test_per_site_kernel_with_more_than_8191_A_values():
    """nA = 9000 (e.g. --rangeA 1,9000,1): the grouped kernels pack the winning A index into 13 bits and are not used; the
    per-site kernel keeps it in a register of its own.  One grid pair (--fixX --fixAlpha), the winner must come from beyond
    index 8191 on some test sites, against the C oracle."""
    from ballermixplus_amd import engine as eng
    rng = np.random.default_rng(5)
    N, n = 600, 50
    gen = np.cumsum(rng.geometric(0.2, N)).astype(np.float64) * 1e-6
    k = rng.integers(1, n + 1, N)
    nn = np.full(N, n)
    cnt = {}
    for a in k.tolist():
        cnt[(a, n)] = cnt.get((a, n), 0) + 1
    spect = {key: v / N for key, v in cnt.items()}
    # 8300 values at which no window holds a site (nothing can win there), then 700 useful ones: every winner has an index > 8191
    As = [1e9 + 1e3 * i for i in range(8300)] + [float(v) for v in np.linspace(2e5, 10.0, 700)]
    model = eng.ModelArrays('B2', int(k.min()), [n], spect, {n: 1.0}, [0.3], [10.0])
    rows = model.rows_of(k, nn)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    ctx.set_sites(gen, rows)
    idx = np.arange(0, N, 7)
    lo, hi = np.zeros(len(idx), np.int64), np.full(len(idx), N - 1, np.int64)
    _, R = ctx.fetch_lut()
    ref = c_scan(c_oracle(), np.where(np.isfinite(R), R, 0.0), As, gen, rows, gen[idx], lo, hi)
    for variant, kernel in ((0, 'clr_scan_solo_kernel'), (2, 'clr_scan_kernel')):      # both one-test-site-per-wave kernels
        ctx.set_variant(variant)
        ctx.set_tests(gen[idx], lo, hi)
        assert ctx.plan()['kernel'].startswith(kernel + '<')
        ctx.scan()
        got = ctx.fetch()
        for q in (1, 2, 3, 4):
            assert np.array_equal(got[q], ref[q]), (variant, q)
        assert np.allclose(got[0], ref[0], rtol=1e-9, atol=1e-12)
        assert np.sum(got[3] >= 0) > 20 and np.min(got[3][got[3] >= 0]) > 8191
    ctx.close()


def _cli(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(REPO, 'BalLeRMixPlus_amd.py')] + args, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r


def test_cli_several_input_files_in_one_process(tmp_path):
    """The multi-file form (-i a,b,c / --inputs list, -o directory or pattern): three chromosomes in one process on one
    context -- byte-identical to three single-file runs; the table is built once while the model stays the same and rebuilt
    for a file with another sample size."""
    from ballermixplus_amd import helpers, synth
    files, cat = [], tmp_path / 'all.txt'
    with open(cat, 'w') as fc:
        fc.write('physPos\tgenPos\tx\tn\n')
        for c, (N, n) in enumerate(((30000, 100), (12000, 100), (20000, 100), (9000, 60)), start=1):
            phys, gen, k, nn = synth.synth_chromosome(N, n, 20 + c)
            p = tmp_path / ('chr%d.txt' % c)
            synth.write_input(str(p), phys, gen, k, nn)
            fc.write(''.join(open(p).readlines()[1:]))
            files.append(str(p))
    sp = tmp_path / 'spect.txt'
    helpers.getSpect(str(cat), str(sp), False, False)
    single = []
    for f in files:
        o = f + '.single'
        _cli(['-i', f, '--spect', str(sp), '-o', o])
        single.append(open(o, 'rb').read())
    outdir = tmp_path / 'out'
    r = _cli(['-i', ','.join(files), '--spect', str(sp), '-o', str(outdir)])
    assert 'selection table built 2 time(s)' in r.stdout            # chr1-3 share n = 100; chr4 has n = 60
    for f, want in zip(files, single):
        assert open(outdir / (os.path.basename(f) + '.out.txt'), 'rb').read() == want
    lst = tmp_path / 'list.txt'
    lst.write_text('\n'.join(files[:3]) + '\n')
    r = _cli(['--inputs', str(lst), '--spect', str(sp), '-o', str(tmp_path / 'pat_{}.tsv'), '-s', '3'])
    assert 'selection table built 1 time(s)' in r.stdout
    for f in files[:3]:
        o = f + '.s3'
        _cli(['-i', f, '--spect', str(sp), '-o', o, '-s', '3'])
        stem = os.path.splitext(os.path.basename(f))[0]
        assert open(tmp_path / ('pat_%s.tsv' % stem), 'rb').read() == open(o, 'rb').read()
    # the one-file form is untouched, and a reference command line still works as it is
    assert subprocess.run([sys.executable, os.path.join(REPO, 'BalLeRMixPlus_amd.py'), '-i', files[0] + ',' + files[1], '--spect', str(sp)],
                          capture_output=True, text=True).returncode != 0      # several files need -o


def test_library_level_multi_gpu_one_shot():
    """bmx_scan_multi (threads + one context per device inside libbmxscan.so, test sites dealt in blocks of 4096): with one
    worker, and with two and three workers that all use this box's one GPU, bitwise what bmx_scan returns; a bad device index
    is reported from the worker that hit it."""
    import ctypes as C
    from ballermixplus_amd import _lib
    eng, data, model, (xs, ab, As) = _synth_model(chroms=((4, 60000),))
    phys, gen, k, nn = data[0]
    N = len(gen)
    rows = model.rows_of(k, nn)
    M = 3 * 4096 + 777
    idx = 10000 + np.arange(M)
    tg = _lib.f64(gen[idx])
    lo, hi = np.zeros(M, np.int64), np.full(M, N - 1, np.int64)
    A = _lib.f64(As)
    L = _lib.lib()

    def call(fn, *tail):
        clr = np.empty(M)
        ix, ia, iA, ns = (np.empty(M, np.int32) for _ in range(4))
        rc = fn(C.byref(model.c), _lib.as_dp(A), len(A), N, _lib.as_dp(_lib.f64(gen)), _lib.as_ip(rows), M, _lib.as_dp(tg),
                _lib.as_lp(lo), _lib.as_lp(hi), _lib.as_dp(clr), _lib.as_ip(ix), _lib.as_ip(ia), _lib.as_ip(iA), _lib.as_ip(ns), *tail)
        return rc, (clr, ix, ia, iA, ns)

    rc, want = call(L.bmx_scan, 0)
    assert rc == 0
    for devs in ([0], [0, 0], [0, 0, 0]):
        d = _lib.i32(devs)
        rc, got = call(L.bmx_scan_multi, len(devs), _lib.as_ip(d))
        assert rc == 0, L.bmx_last_error()
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), devs
    rc, got = call(L.bmx_scan_multi, 1, None)          # devices = NULL: GPU 0
    assert rc == 0 and all(np.array_equal(a, b) for a, b in zip(got, want))
    d = _lib.i32([0, 99])
    rc, _ = call(L.bmx_scan_multi, 2, _lib.as_ip(d))
    assert rc == -2 and b'worker 1' in L.bmx_last_error()


@pytest.mark.parametrize('step', [14, 64, 200, 1000])
def test_solo_pipeline_on_sparse_test_sets(step):
    """Sparse test sets (-s 60 ... 1000) go through the one-test-site-per-wave prepared pipeline (prep_solo_kernel +
    clr_scan_solo_kernel): identical argmax and nSites, CLR to 1e-10, against the round-2 per-site kernel (variant 2) on every
    test site and against the C oracle on a sample; windows limited by index bounds too."""
    eng, data, model, (xs, ab, As) = _synth_model(chroms=((2, 600000),))
    phys, gen, k, nn = data[0]
    N = len(gen)
    rows = model.rows_of(k, nn)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    ctx.set_sites(gen, rows)
    idx = np.arange(1000, N - 1000, step)[:6000]
    M = len(idx)
    for lo, hi in ((np.zeros(M, np.int64), np.full(M, N - 1, np.int64)),
                   (np.maximum(idx - 300, 0).astype(np.int64), np.minimum(idx + 1500, N - 1).astype(np.int64))):
        out = {}
        for v in (0, 2):
            ctx.set_variant(v)
            ctx.set_tests(gen[idx], lo, hi)
            assert ctx.plan()['mode'] == (5 if v == 0 else -1)
            ctx.scan()
            out[v] = ctx.fetch()
        for a, b in zip(out[0][1:], out[2][1:]):
            assert np.array_equal(a, b)
        assert np.max(np.abs(out[0][0] - out[2][0]) / np.maximum(np.abs(out[2][0]), 1e-9)) < 1e-10
        _, R = ctx.fetch_lut()
        samp = np.arange(0, M, max(1, M // 32))
        ref = c_scan(c_oracle(), np.where(np.isfinite(R), R, 0.0), As, gen, rows, gen[idx[samp]], lo[samp], hi[samp])
        for q in (1, 2, 3, 4):
            assert np.array_equal(out[0][q][samp], ref[q])
        assert np.allclose(out[0][0][samp], ref[0], rtol=1e-9, atol=1e-12)
    ctx.close()


def test_bench_rccl_path_with_one_rank():
    """bench.py with the process group forced on for a single rank (BMX_FORCE_DIST=1, backend nccl = RCCL): the step packs the
    records on the device, gathers them with RCCL to rank 0 and copies them to pinned memory -- the code path of an N-GPU run,
    on the one GPU this box has; same checksum as the plain one-process run."""
    import json
    env = dict(os.environ, BMX_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29671')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BMX_DIST_BACKEND', 'BMX_SINGLE_DEVICE'):
        env.pop(k, None)
    common = [sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '2', '--warmup', '1', '--total-snps', '400000', '--no-cpu-baseline']
    r = subprocess.run(common, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and 'ONE gather' in d['config']['parallelism'] and d['config']['records_per_step'] == d['config']['windows_per_step']
    env.pop('BMX_FORCE_DIST')
    r1 = subprocess.run(common, capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith('{')][-1])
    assert 'pinned host memory' in d1['config']['parallelism']
    assert d1['config']['checksum_clr'] == pytest.approx(d['config']['checksum_clr'], rel=1e-12)


@pytest.mark.parametrize('step', [1, 5])
def test_prepared_pipeline_with_a_table_too_large_for_lds(step):
    """41 sample sizes per file (3321 table rows: too large for LDS, read from global memory / L2) through the prepared
    pipeline at J = 16 and J = 8: against the round-2 grouped kernel (same table path) on every test site and the C oracle
    on a sample; the plan says R comes from global memory."""
    from ballermixplus_amd import engine as eng, synth
    from ballermixplus_amd.hostmodel import Grids
    N, n, spread = 200000, 100, 40
    phys, gen, k, nn = synth.synth_chromosome(N, n, 6)
    rng = np.random.default_rng(11)
    n2 = rng.integers(n - spread, n + 1, N)
    k = np.where(k == nn, n2, np.maximum(1, np.minimum(n2 - 1, (k * n2) // nn)))
    nn = n2
    sizes = sorted(set(nn.tolist()))
    cnt = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}
    props = {int(s_): float(sum(f for (a, b), f in cnt.items() if b == s_)) for s_ in sizes}
    grid = Grids(None, None, False, False, None, None)
    xs, ab, As = grid.scan_order()
    model = eng.ModelArrays('B2', int(k.min()), sizes, cnt, props, xs, ab)
    rows = model.rows_of(k, nn)
    ctx = eng.Context(0)
    ctx.set_model(model, As)
    ctx.set_sites(gen, rows)
    M = 16384 // step
    idx = 40000 + step * np.arange(M)
    lo, hi = np.zeros(M, np.int64), np.full(M, N - 1, np.int64)
    out = {}
    for v in (0, 12):
        ctx.set_variant(v)
        ctx.set_tests(gen[idx], lo, hi)
        pl = ctx.plan()
        assert not pl['use_lds'] and pl['mode'] == (4 if v == 0 else 3)
        ctx.scan()
        out[v] = ctx.fetch()
    for a, b in zip(out[0][1:], out[12][1:]):
        assert np.array_equal(a, b)
    assert np.max(np.abs(out[0][0] - out[12][0]) / np.maximum(np.abs(out[12][0]), 1e-9)) < 1e-10
    _, R = ctx.fetch_lut()
    samp = np.arange(0, M, max(1, M // 40))
    ref = c_scan(c_oracle(), np.where(np.isfinite(R), R, 0.0), As, gen, rows, gen[idx[samp]], lo[samp], hi[samp])
    for q in (1, 2, 3, 4):
        assert np.array_equal(out[0][q][samp], ref[q])
    assert np.allclose(out[0][0][samp], ref[0], rtol=1e-9, atol=1e-12)
    ctx.close()
