"""Catalogue of reference-style command lines with golden outputs.

Goldens come from (a) the reference's own example outputs (tests/golden/ref_test/output, the
README commands of /root/reference/README.md:122-171) and (b) runs of the reference made in the
build container by tests/golden/make_golden.py (tests/golden/e2e, tests/golden/synth)."""
import json
import os

import numpy as np

from util import GOLD, REFT, Case, read_tsv

from ballermixplus_amd import scan as scanmod
from ballermixplus_amd.cli import build_parser

T = REFT + '/'
REF_CASES = {
    # name: (argv without -o, golden file)
    'ex1_B2': (['-i', T + 'Example1_fullSweep_200kya_DAF.txt', '--spect', T + 'HC_CEU_Neut_DAF_spect_for_B2.txt'],
               T + 'output/Example1_B2.txt'),
    'ex1_B2maf': (['-i', T + 'Example1_fullSweep_200kya_DAF.txt', '--spect', T + 'HC_CEU_Neut_MAF_spect_for_B2maf.txt',
                   '--MAF'], T + 'output/Example1_B2maf.txt'),
    'ex1_B1': (['-i', T + 'Example1_fullSweep_200kya_DAF.txt', '--spect', T + 'HC_CEU_Neut_config_for_B1.txt',
                '--noFreq'], T + 'output/Example1_B1.txt'),
    'ex2_B2': (['-i', T + 'Example2_balancing_10MYA_DAF.txt', '--spect', T + 'HC_CEU_Neut_DAF_spect_for_B2.txt'],
               T + 'output/Example2_B2.txt'),
    'ex2_B2maf': (['-i', T + 'Example2_balancing_10MYA_DAF.txt', '--spect', T + 'HC_CEU_Neut_MAF_spect_for_B2maf.txt',
                   '--MAF'], T + 'output/Example2_B2maf.txt'),
    'ex2_B1': (['-i', T + 'Example2_balancing_10MYA_DAF.txt', '--spect', T + 'HC_CEU_Neut_config_for_B1.txt',
                '--noFreq'], T + 'output/Example2_B1.txt'),
    'ex2_B0maf_1kb': (['-i', T + 'Example2_balancing_10MYA_MAF_nosub.txt', '--spect',
                       T + 'HC_CEU_Neut_MAF-noSub_spect_for_B0maf.txt', '--noSub', '--MAF', '--usePhysPos',
                       '--fixWinSize', '-w', '1000', '--step', '2'], T + 'output/Example2_B0maf_1kb-2site.txt'),
}


def e2e_cases():
    out = {}
    d = os.path.join(GOLD, 'e2e')
    for f in sorted(os.listdir(d)):
        if f.endswith('.args.json'):
            name = f[:-len('.args.json')]
            with open(os.path.join(d, f)) as fh:
                args = json.load(fh)
            args = [T + a if a.endswith('.txt') else a for a in args]
            out[name] = (args, os.path.join(d, name + '.tsv'))
    return out


ALL_CASES = dict(REF_CASES)
ALL_CASES.update(e2e_cases())


def parse(argv):
    return build_parser().parse_args(argv + ['-o', '/dev/null'])


def host_side(argv):
    """Everything the CLI does before touching the GPU: Case (data, neutral, grids) + test sites."""
    opt = parse(argv)
    case = Case(opt.infile, opt.spectfile, nofreq=opt.nofreq, MAF=opt.MAF, nosub=opt.nosub,
                phys=opt.phys or opt.size, Rrate=opt.Rrate, x=opt.x, abeta=opt.abeta, bal=opt.bal, pos=opt.pos,
                seqA=opt.seqA, listA=opt.listA)
    d = case.data
    if opt.size:
        w = float(opt.w)
        ts = scanmod.sites_fix_nocenter(d, w, opt.step) if opt.noCenter else scanmod.sites_fix_center(d, w, opt.step)
    elif opt.w != 0:
        ts = scanmod.sites_site_based(d, opt.w, opt.step)
    else:
        ts = scanmod.sites_alpha(d, opt.step)
    return opt, case, ts


def T_at_grid_point(case, ts, j, x_s, a_s, A_s):
    """T of test site j at the grid point printed as (x_s, a_s, A_s), by the oracle's LUT form."""
    from util import orc
    m = case.oracle_model()
    ix = [repr(v) if not isinstance(v, str) else v for v in case.xs].index(x_s)
    ia = [repr(v) for v in case.abetas].index(a_s)
    iA = [repr(v) for v in case.As].index(A_s)
    sub, alphas = orc.window_mask(m, case.As[iA], ts.lo[j], ts.hi[j], ts.test_gen[j])
    if len(sub) == 0:
        return float('nan'), 0
    with np.errstate(divide='ignore', invalid='ignore'):
        return float(2.0 * np.sum(np.log1p(alphas[sub] * m.R[ix, ia, m.row[sub]]))), len(sub)


def compare_rows(got_lines, gold_path, rtol=1e-6, atol=1e-9, case=None, ts=None, tie_rtol=1e-9):
    """got_lines: output lines (header excluded).  All fields but CLR must be string-identical;
    CLR within rtol relative (atol floor).  A different argmax is accepted only as a TIE WITHIN
    ROUNDING NOISE: the oracle's T at the golden's grid point is within tie_rtol of ours (the
    reference itself resolves such ties by the rounding of its own sums; they occur where the
    selection table saturates, e.g. B_1's two-row table).  Returns (worst relative CLR
    difference, number of ties)."""
    gold = read_tsv(gold_path)
    got = [l.rstrip('\n').split('\t') for l in got_lines]
    assert len(got) == len(gold), (len(got), len(gold))
    worst = 0.0
    ties = 0
    pos_of = {p: j for j, p in enumerate(ts.order)} if ts is not None else {}
    for i, (a, b) in enumerate(zip(got, gold)):
        assert a[:2] == b[:2], (i, a, b)
        if a[2] != b[2]:
            x, y = float(a[2]), float(b[2])
            assert abs(x - y) <= max(atol, rtol * abs(y)), (i, a, b)
            if y != 0:
                worst = max(worst, abs(x - y) / abs(y))
        if a[3:] != b[3:]:
            assert case is not None and b[5] != 'NA', (i, a, b)
            if b[3:] == ['0.0'] * 4:      # golden says "nothing beat 0": our T must be ~0 too
                assert abs(float(a[2])) <= atol, (i, a, b)
            else:
                T, ns = T_at_grid_point(case, ts, pos_of[i], b[3], b[4], b[5])
                assert abs(T - float(a[2])) <= max(atol, tie_rtol * abs(float(a[2]))), (i, a, b, T)
                assert a[6] == b[6] or str(ns) == b[6], (i, a, b)
            ties += 1
    return worst, ties
