#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container, where the reference checkout is mounted at
/root/reference (it never travels to the GPU box).  The reference module is
imported by path and driven through its own classes/functions; nothing of its
source is copied.  Each sub-command writes small data files (inputs + expected
outputs) next to this script.

    python tests/golden/make_golden.py lut        # NormalizedBetaBinom tables
    python tests/golden/make_golden.py setorder   # list(set(grid)) iteration orders
    python tests/golden/make_golden.py surface    # per-site T[A,x,a] surfaces
    python tests/golden/make_golden.py helpers    # --getSpect / --getConfig outputs
    python tests/golden/make_golden.py hostmodel  # NeutralSFS / InputData state after construction (bitwise) + stdout
    python tests/golden/make_golden.py e2e NAME   # whole-CLI runs (slow, minutes..)
    python tests/golden/make_golden.py synth      # synthetic 20k / 1M strided windows
    python tests/golden/make_golden.py config4 21 22   # BASELINE config 4: chromosomes of the 40M-SNP genome, -s 50000
    python tests/golden/make_golden.py config5 1 2     # BASELINE config 5: contigs of 1.25M SNPs, n=200, 100x10x44 grid
"""
import importlib.util
import json
import os
import subprocess
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
REF_PY = os.path.join(REF, 'BalLeRMix+_v1.py')
REF_TEST = os.path.join(REF, 'test')
sys.path.insert(0, REPO)


def load_ref():
    spec = importlib.util.spec_from_file_location('ballermix_ref', REF_PY)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def tiny_input(path, rows):
    with open(path, 'w') as f:
        f.write('physPos\tgenPos\tx\tn\n')
        for i, (k, n) in enumerate(rows):
            f.write('%d\t%.6f\t%d\t%d\n' % (100 * (i + 1), 1e-4 * (i + 1), k, n))


def grid_of(ref, kind):
    if kind == 'default':
        return ref.Grids(None, None, False, False, None, None)
    if kind == 'bal':
        return ref.Grids(None, None, True, False, None, None)
    raise ValueError(kind)


# ----------------------------------------------------------------------------- lut
def cmd_lut():
    """normProbs[(x,a)] for every admissible k, per statistic / n / minCount."""
    ref = load_ref()
    tmp = '/tmp/bmx_golden_lut.txt'
    cases = []
    for n in (50, 100):
        cases += [('B2', n, 'default', 1), ('B2maf', n, 'default', 1),
                  ('B0', n, 'default', 1), ('B0maf', n, 'default', 1)]
    cases += [('B2', 50, 'default', 2), ('B2maf', 50, 'bal', 2), ('B2', 200, 'bal', 1),
              ('B0maf', 50, 'default', 3), ('B1', 50, 'default', 1), ('B2', 7, 'default', 1),
              ('B2maf', 9, 'default', 1), ('B2maf', 8, 'bal', 1)]
    for stat, n, gkind, minc in cases:
        maf = stat.endswith('maf')
        nosub = stat.startswith('B0')
        nofreq = stat == 'B1'
        if nofreq:
            ks = [0, 1]
        elif maf:
            lo = 0 if not nosub else minc
            ks = [k for k in range(lo, n // 2 + 1) if k == 0 or k >= minc]
        else:
            hi = n - 1 if nosub else n
            ks = list(range(minc, hi + 1))
        rows = [(k, n) for k in ks]
        # two sample sizes in one file for one case, to pin the per-n handling
        if stat == 'B2' and n == 50 and minc == 1:
            rows += [(k, 40) for k in range(1, 41)]
        tiny_input(tmp, rows)
        data = ref.InputData(tmp, nofreq, maf, nosub, minc)
        grid = grid_of(ref, gkind)
        t0 = time.time()
        nb = ref.NormalizedBetaBinom(data, grid, nofreq, maf, nosub)
        xs = list(grid.x)
        abs_ = list(grid.abeta)
        tab = np.stack([np.stack([nb.get(x, a) for a in abs_]) for x in xs])
        name = 'lut_%s_n%d_%s_min%d.npz' % (stat, n, gkind, minc)
        np.savez_compressed(os.path.join(HERE, name), table=tab,
                            x=np.array(xs, dtype=np.float64), abeta=np.array(abs_, dtype=np.float64),
                            count=np.asarray(data.count), total=np.asarray(data.total),
                            minCount=np.int64(data.minCount))
        print(name, tab.shape, 'minCount', data.minCount, '%.1fs' % (time.time() - t0))


# ------------------------------------------------------------------------ setorder
def cmd_setorder():
    ref = load_ref()
    out = {}
    for kind in ('default', 'bal'):
        g = grid_of(ref, kind)
        out[kind] = {
            'x': [repr(v) for v in list(set(g.x))],
            'abeta': [repr(v) for v in list(set(g.abeta))],
            'A': [repr(v) for v in list(set(g.A))],
            'x_list': [repr(v) for v in g.x],
            'abeta_list': [repr(v) for v in g.abeta],
            'A_list': [repr(v) for v in g.A],
        }
    listA = ','.join(str(100 * i) for i in range(1, 101))
    g = ref.Grids(None, None, True, True, None, listA)
    out['config5'] = {'x': [repr(v) for v in list(set(g.x))],
                      'abeta': [repr(v) for v in list(set(g.abeta))],
                      'A': [repr(v) for v in list(set(g.A))],
                      'listA': listA}
    g = ref.Grids('0.3', 7.0, False, False, None, '250,1e3,77.5')
    out['fixed'] = {'x': [repr(v) for v in list(set(g.x))],
                    'abeta': [repr(v) for v in list(set(g.abeta))],
                    'A': [repr(v) for v in list(set(g.A))]}
    with open(os.path.join(HERE, 'setorder.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('setorder.json written')


# ------------------------------------------------------------------------- surface
def _surface(ref, data, neut, nb, grid, site_i, window=None):
    xs, abs_, As = list(set(grid.x)), list(set(grid.abeta)), list(set(grid.A))
    T = np.full((len(As), len(xs), len(abs_)), np.nan)
    ns = np.zeros(len(As), dtype=np.int64)
    if window is None:
        window = np.arange(data.numSites, dtype=int)
    test = data.genPos[site_i]
    for iA, A in enumerate(As):
        for ix, x in enumerate(xs):
            for ia, a in enumerate(abs_):
                g1 = types.SimpleNamespace(x=[x], abeta=[a], A=[A])
                r = ref.calcBaller(window, test, data, neut, nb, g1)
                if r[0] > 0:
                    T[iA, ix, ia] = r[0]
                    ns[iA] = r[4]
    return T, ns, xs, abs_, As


def cmd_surface():
    """T[A,x,a] (NaN where the reference reports 'not > 0') at a few sites."""
    ref = load_ref()
    jobs = [
        ('ex1_B2', 'Example1_fullSweep_200kya_DAF.txt', 'HC_CEU_Neut_DAF_spect_for_B2.txt',
         dict(nofreq=False, MAF=False, nosub=False), 'default', [0, 378, 756]),
        ('ex2_B2maf_bal', 'Example2_balancing_10MYA_MAF.txt', 'HC_CEU_Neut_MAF_spect_for_B2maf.txt',
         dict(nofreq=False, MAF=True, nosub=False), 'bal', [592, 1183]),
    ]
    for name, inp, spect, fl, gkind, sites in jobs:
        data = ref.InputData(os.path.join(REF_TEST, inp), fl['nofreq'], fl['MAF'], fl['nosub'], 1)
        neut = ref.NeutralSFS(os.path.join(REF_TEST, spect), fl['nofreq'], fl['MAF'], fl['nosub'])
        neut.get_neut_probs(data)
        grid = grid_of(ref, gkind)
        nb = ref.NormalizedBetaBinom(data, grid, fl['nofreq'], fl['MAF'], fl['nosub'])
        for s in sites:
            t0 = time.time()
            T, ns, xs, abs_, As = _surface(ref, data, neut, nb, grid, s)
            full = ref.calcBaller(np.arange(data.numSites, dtype=int), data.genPos[s], data, neut, nb, grid)
            fn = 'surface_%s_site%d.npz' % (name, s)
            np.savez_compressed(os.path.join(HERE, fn), T=T, nsites=ns,
                                x=np.array(xs, float), abeta=np.array(abs_, float), A=np.array(As, float),
                                site=np.int64(s), best=np.array([float(v) for v in full]))
            print(fn, 'best', full, '%.1fs' % (time.time() - t0))


# ------------------------------------------------------------------------- helpers
def cmd_helpers():
    """--getSpect / --getConfig byte-exact outputs on the example inputs."""
    outdir = os.path.join(HERE, 'helpers')
    os.makedirs(outdir, exist_ok=True)
    runs = [
        ('spect_ex1_DAF.txt', 'Example1_fullSweep_200kya_DAF.txt', ['--getSpect']),
        ('spect_ex1_MAFfold.txt', 'Example1_fullSweep_200kya_DAF.txt', ['--getSpect', '--MAF']),
        ('spect_ex2_MAF.txt', 'Example2_balancing_10MYA_MAF.txt', ['--getSpect', '--MAF']),
        ('spect_ex2_DAF_nosub.txt', 'Example2_balancing_10MYA_DAF.txt', ['--getSpect', '--noSub']),
        ('spect_ex2_MAF_nosub.txt', 'Example2_balancing_10MYA_MAF.txt', ['--getSpect', '--MAF', '--noSub']),
        ('config_ex1.txt', 'Example1_fullSweep_200kya_DAF.txt', ['--getConfig']),
        ('config_ex2.txt', 'Example2_balancing_10MYA_DAF.txt', ['--getConfig']),
    ]
    for out, inp, flags in runs:
        cmd = [sys.executable, REF_PY, '-i', os.path.join(REF_TEST, inp), '--spect',
               os.path.join(outdir, out)] + flags
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        print(out)


# ----------------------------------------------------------------------------- e2e
E2E = {
    # BASELINE config 2 (no upstream golden exists for --findBal)
    'ex2_B2maf_findBal': (['-i', REF_TEST + '/Example2_balancing_10MYA_MAF.txt', '--spect',
                           REF_TEST + '/HC_CEU_Neut_MAF_spect_for_B2maf.txt', '--MAF', '--findBal'], None),
    # window modes (strided so they finish in minutes)
    'ex1_B2_w50_s25': (['-i', REF_TEST + '/Example1_fullSweep_200kya_DAF.txt', '--spect',
                        REF_TEST + '/HC_CEU_Neut_DAF_spect_for_B2.txt', '-w', '50', '-s', '25'], None),
    'ex2_B2_fix5kb_s40': (['-i', REF_TEST + '/Example2_balancing_10MYA_DAF.txt', '--spect',
                           REF_TEST + '/HC_CEU_Neut_DAF_spect_for_B2.txt', '--fixWinSize', '-w', '5000',
                           '-s', '40', '--usePhysPos'], None),
    'ex2_B0_noCenter_2kb': (['-i', REF_TEST + '/Example2_balancing_10MYA_DAF_nosub.txt', '--spect',
                             REF_TEST + '/HC_CEU_Neut_DAF-nosub_spect_for_B0.txt', '--noSub', '--fixWinSize',
                             '-w', '2000', '-s', '2000', '--noCenter', '--usePhysPos'], None),
    'ex1_B2_fixX_fixAlpha_listA': (['-i', REF_TEST + '/Example1_fullSweep_200kya_DAF.txt', '--spect',
                                    REF_TEST + '/HC_CEU_Neut_DAF_spect_for_B2.txt', '--fixX', '0.3',
                                    '--fixAlpha', '7', '--listA', '250,1e3,77.5', '-s', '5'], None),
    'ex1_B2maf_s20': (['-i', REF_TEST + '/Example1_fullSweep_200kya_MAF.txt', '--spect',
                       REF_TEST + '/HC_CEU_Neut_MAF_spect_for_B2maf.txt', '--MAF', '-s', '20'], None),
}


def cmd_e2e(names):
    outdir = os.path.join(HERE, 'e2e')
    os.makedirs(outdir, exist_ok=True)
    for name in names:
        args, _ = E2E[name]
        out = os.path.join(outdir, name + '.tsv')
        t0 = time.time()
        subprocess.run([sys.executable, REF_PY] + args + ['-o', out], check=True, stdout=subprocess.DEVNULL)
        with open(os.path.join(outdir, name + '.args.json'), 'w') as f:
            json.dump([a.replace(REF_TEST + '/', '') for a in args], f)
        print(name, '%.0fs' % (time.time() - t0))


# --------------------------------------------------------------------------- synth
def cmd_synth(which):
    """Strided reference runs on the synthetic chromosomes of SURVEY 8(d)."""
    from ballermixplus_amd import synth
    ref = load_ref()
    outdir = os.path.join(HERE, 'synth')
    os.makedirs(outdir, exist_ok=True)
    cfgs = {'20k': (20000, 100, 1, 200), '1M': (1000000, 100, 1, 100000), '1M_dense': (1000000, 100, 1, 4000), '1M_n200_bal': (1000000, 200, 2, 40000),
            '20k_n200_bal': (20000, 200, 2, 400)}
    for key in which:
        N, n, chrom, step = cfgs[key]
        phys, gen, k, nn = synth.synth_chromosome(N, n, chrom)
        inp = '/tmp/bmx_synth_%s.txt' % key
        spectf = '/tmp/bmx_synth_%s_spect.txt' % key
        synth.write_input(inp, phys, gen, k, nn)
        ref.getSpect(inp, spectf, False, False)
        bal = key.endswith('bal')
        data = ref.InputData(inp, False, False, False, 1)
        neut = ref.NeutralSFS(spectf, False, False, False)
        neut.get_neut_probs(data)
        if bal:
            listA = ','.join(str(100 * i) for i in range(1, 101))
            grid = ref.Grids(None, None, True, True, None, listA)
        else:
            grid = grid_of(ref, 'default')
        t0 = time.time()
        nb = ref.NormalizedBetaBinom(data, grid, False, False, False)
        print(key, 'NormalizedBetaBinom %.0fs' % (time.time() - t0), flush=True)
        rows = []
        allidx = np.arange(data.numSites, dtype=int)
        for i in range(0, data.numSites, step):
            t0 = time.time()
            r = ref.calcBaller(allidx, data.genPos[i], data, neut, nb, grid)
            rows.append(f'{data.position[i]}\t{data.genPos[i]}\t{r[0]}\t{r[1]}\t{r[2]}\t{r[3]}\t{r[4]}')
            print(key, i, rows[-1], '%.1fs' % (time.time() - t0), flush=True)
        with open(os.path.join(outdir, 'synth_%s_step%d.tsv' % (key.replace('_dense', ''), step)), 'w') as f:
            f.write('physPos\tgenPos\tCLR\tx_hat\ts_hat\tA_hat\tnSites\n')
            f.write('\n'.join(rows) + '\n')


def _strided_reference_run(ref, tag, phys, gen, k, nn, spect_rows, grid, step, out_path):
    """One chromosome through the reference's own classes, every step-th site, with the helper file of the whole
    configuration (spect_rows: the --getSpect table of the concatenated input)."""
    inp = '/tmp/bmx_%s.txt' % tag
    spectf = '/tmp/bmx_%s_spect.txt' % tag
    from ballermixplus_amd import synth
    synth.write_input(inp, phys, gen, k, nn)
    with open(spectf, 'w') as f:                       # the format getSpect writes (v1:699-708): k n fraction
        for a, b, fr in spect_rows:
            f.write('%s\t%s\t%s\n' % (a, b, fr))
    data = ref.InputData(inp, False, False, False, 1)
    neut = ref.NeutralSFS(spectf, False, False, False)
    neut.get_neut_probs(data)
    t0 = time.time()
    nb = ref.NormalizedBetaBinom(data, grid, False, False, False)
    print(tag, 'NormalizedBetaBinom %.0fs' % (time.time() - t0), flush=True)
    rows = []
    allidx = np.arange(data.numSites, dtype=int)
    for i in range(0, data.numSites, step):
        t0 = time.time()
        r = ref.calcBaller(allidx, data.genPos[i], data, neut, nb, grid)
        rows.append(f'{data.position[i]}\t{data.genPos[i]}\t{r[0]}\t{r[1]}\t{r[2]}\t{r[3]}\t{r[4]}')
        print(tag, i, rows[-1], '%.1fs' % (time.time() - t0), flush=True)
    with open(out_path, 'w') as f:
        f.write('physPos\tgenPos\tCLR\tx_hat\ts_hat\tA_hat\tnSites\n')
        f.write('\n'.join(rows) + '\n')
    os.remove(inp)


def cmd_config4(chroms, step=50000):
    """BASELINE config 4: 22 chromosomes, 40M SNPs, n=100; the helper file is --getSpect of the concatenation
    of all 22.  The reference scans the requested chromosomes with -s 50000 (it needs ~5 min and ~8 GB per chromosome of
    ~700k SNPs; chr1's 3.46M are out of its reach)."""
    from ballermixplus_amd import synth
    ref = load_ref()
    sizes = synth.config4_sizes(40_000_000)
    data = [synth.synth_chromosome(N, 100, c + 1) for c, N in enumerate(sizes)]
    spect_rows = synth.spect_from_counts(np.concatenate([d[2] for d in data]), np.concatenate([d[3] for d in data]))
    outdir = os.path.join(HERE, 'synth')
    for c in chroms:
        phys, gen, k, nn = data[c - 1]
        _strided_reference_run(ref, 'cfg4_chr%d' % c, phys, gen, k, nn, spect_rows, grid_of(ref, 'default'), step,
                               os.path.join(outdir, 'config4_chr%d_step%d.tsv' % (c, step)))


def cmd_config5(contigs, step=125000):
    """BASELINE config 5: 8 contigs of 1.25M SNPs, n=200, A = 100..10000 step 100 (--listA: the reference's --rangeA
    raises), --findBal --findPos; helper file from the concatenation of the 8 contigs."""
    from ballermixplus_amd import synth
    ref = load_ref()
    data = [synth.synth_chromosome(1250000, 200, c + 1) for c in range(8)]
    spect_rows = synth.spect_from_counts(np.concatenate([d[2] for d in data]), np.concatenate([d[3] for d in data]))
    listA = ','.join(str(100 * i) for i in range(1, 101))
    grid = ref.Grids(None, None, True, True, None, listA)
    outdir = os.path.join(HERE, 'synth')
    for c in contigs:
        phys, gen, k, nn = data[c - 1]
        _strided_reference_run(ref, 'cfg5_contig%d' % c, phys, gen, k, nn, spect_rows, grid, step,
                               os.path.join(outdir, 'config5_contig%d_step%d.tsv' % (c, step)))


# ----------------------------------------------------------------------------- neutral model + input classes
NEUTRAL_CASES = [('HC_CEU_Neut_DAF_spect_for_B2.txt', (False, False, False)), ('HC_CEU_Neut_MAF_spect_for_B2maf.txt', (False, True, False)),
                 ('HC_CEU_Neut_DAF_spect_for_B2.txt', (False, True, False)),          # a polarised spectrum used folded (v1:195-203)
                 ('HC_CEU_Neut_MAF-noSub_spect_for_B0maf.txt', (False, True, True)), ('HC_CEU_Neut_DAF-nosub_spect_for_B0.txt', (False, False, True)),
                 ('HC_CEU_Neut_config_for_B1.txt', (True, False, False))]
INPUT_CASES = [('Example1_fullSweep_200kya_DAF.txt', {}), ('Example1_fullSweep_200kya_DAF.txt', dict(nofreq=True)),
               ('Example1_fullSweep_200kya_DAF.txt', dict(MAF=True)), ('Example2_balancing_10MYA_MAF_nosub.txt', dict(phys=True, Rrate=1.25e-6)),
               ('Example2_balancing_10MYA_DAF.txt', dict(nosub=True)), ('Example2_balancing_10MYA_DAF.txt', dict(MAF=True, nosub=True))]


def cmd_hostmodel():
    """What the reference's NeutralSFS / InputData hold after construction (floats as repr strings: bitwise), plus stdout."""
    import contextlib
    import hashlib
    import io
    ref = load_ref()
    out = {'neutral': [], 'input': []}
    for fname, args in NEUTRAL_CASES:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            n = ref.NeutralSFS(os.path.join(REF_TEST, fname), *args)
        out['neutral'].append({'file': fname, 'args': list(args), 'stdout': buf.getvalue(),
                               'spect': [[int(k), int(m), repr(float(v))] for (k, m), v in sorted(n.spect.items())],
                               'sampProps': [[int(m), repr(float(v))] for m, v in sorted(n.sampProps.items())],
                               'sampSizes': sorted(int(v) for v in n.sampSizes)})
    for fname, kw in INPUT_CASES:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            d = ref.InputData(os.path.join(REF_TEST, fname), kw.get('nofreq', False), kw.get('MAF', False), kw.get('nosub', False), 1,
                              phys=kw.get('phys', False), Rrate=kw.get('Rrate', 1e-6))
        h = lambda a, dt: hashlib.sha256(np.ascontiguousarray(np.asarray(a), dtype=dt).tobytes()).hexdigest()
        out['input'].append({'file': fname, 'kw': kw, 'stdout': buf.getvalue(), 'numSites': int(d.numSites), 'minCount': int(d.minCount),
                             'sampSizes': sorted(int(v) for v in d.sampSizes), 'position_sha256': h(d.position, np.int64),
                             'genPos_sha256': h(d.genPos, np.float64), 'count_sha256': h(d.count, np.int64), 'total_sha256': h(d.total, np.int64)})
    with open(os.path.join(HERE, 'hostmodel.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('hostmodel.json: %d neutral, %d input cases' % (len(out['neutral']), len(out['input'])))


if __name__ == '__main__':
    cmd = sys.argv[1]
    if cmd == 'hostmodel':
        cmd_hostmodel()
        sys.exit(0)
    if cmd == 'lut':
        cmd_lut()
    elif cmd == 'setorder':
        cmd_setorder()
    elif cmd == 'surface':
        cmd_surface()
    elif cmd == 'helpers':
        cmd_helpers()
    elif cmd == 'e2e':
        cmd_e2e(sys.argv[2:] or list(E2E))
    elif cmd == 'synth':
        cmd_synth(sys.argv[2:] or ['20k'])
    elif cmd == 'config4':
        cmd_config4([int(v) for v in sys.argv[2:]] or [21, 22])
    elif cmd == 'config5':
        cmd_config5([int(v) for v in sys.argv[2:]] or [1, 2])
    else:
        raise SystemExit(__doc__)
