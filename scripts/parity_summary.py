"""Observed parity of the GPU path against every golden file: rows, exact-field matches, max |dCLR|/CLR."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import cases
from util import GOLD, read_tsv
from ballermixplus_amd import engine, scan as scanmod

print('%-32s %6s %6s %12s' % ('golden file', 'rows', 'ties', 'max rel dCLR'))
for name in sorted(cases.ALL_CASES):
    argv, gold = cases.ALL_CASES[name]
    if not os.path.exists(gold):
        continue
    opt, case, ts = cases.host_side(argv)
    sel = engine.NormalizedBetaBinom(case.data, case.grid, opt.nofreq, opt.MAF, opt.nosub).bind(case.neut)
    res = engine.scan_batch(sel, ts.test_gen, ts.lo, ts.hi)
    out = '/tmp/parity_%s.tsv' % name
    scanmod.write_rows(out, ts, res, sel)
    lines = open(out).readlines()[1:]
    worst, ties = cases.compare_rows(lines, gold, rtol=1e-6, case=case, ts=ts)
    print('%-32s %6d %6d %12.2e' % (name, len(lines), ties, worst))
# synthetic, full scale
from test_gpu_parity import _synth_case
for key, N, n, step, bal in [('20k', 20000, 100, 200, False), ('20k_n200_bal', 20000, 200, 400, True),
                             ('1M', 1000000, 100, 100000, False), ('1M', 1000000, 100, 4000, False),
                             ('1M_n200_bal', 1000000, 200, 40000, True)]:
    path = os.path.join(GOLD, 'synth', 'synth_%s_step%d.tsv' % (key, step))
    if not os.path.exists(path):
        continue
    listA = ','.join(str(100 * i) for i in range(1, 101)) if bal else None
    phys, gen, k, nn, spect, props, grid = _synth_case(N, n, 2 if bal else 1, bal, listA)
    xs, ab, As = grid.scan_order()
    model = engine.ModelArrays('B2', int(k.min()), [n], spect, props, xs, ab)
    ctx = engine.Context(0); ctx.set_model(model, As); ctx.set_sites(gen, model.rows_of(k, nn))
    ref_idx = np.arange(0, N, step)
    idx = np.unique(np.clip(ref_idx[:, None] + np.arange(-5, 11)[None, :], 0, N - 1).reshape(-1))   # grouped kernel
    ctx.set_tests(gen[idx], np.zeros(len(idx), np.int64), np.full(len(idx), N - 1, np.int64))
    ctx.scan(); clr, ix, ia, iA, ns = ctx.fetch()
    rows = read_tsv(path)
    worst = 0.0; exact = 0
    for j, r in zip(np.searchsorted(idx, ref_idx), rows):
        if r[3:6] == ['0.0', '0.0', '0.0']:
            exact += int(iA[j] < 0); continue
        exact += int((repr(xs[ix[j]]), repr(ab[ia[j]]), repr(As[iA[j]]), str(ns[j])) == (r[3], r[4], r[5], r[6]))
        worst = max(worst, abs(clr[j] - float(r[2])) / abs(float(r[2])))
    print('%-32s %6d %6s %12.2e   (argmax+nSites identical on %d rows)' % ('synth_%s_step%d' % (key, step), len(rows), '-', worst, exact))
    ctx.close()
