"""Far-field power sums (the default kernel form): CLR difference against the exact product form
(variant 10) and time per scan, for several truncation thresholds (BMX_FAR_EPS).
Config-3 chromosome, a block of consecutive test sites."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ballermixplus_amd import engine as eng, synth
from ballermixplus_amd.hostmodel import Grids

N, n = 1000000, 100
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
phys, gen, k, nn = synth.synth_chromosome(N, n, 1)
spect = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}
grid = Grids(None, None, False, False, None, None)
xs, ab, As = grid.scan_order()
model = eng.ModelArrays('B2', int(k.min()), [n], spect, {n: 1.0}, xs, ab)
ctx = eng.Context(0)
ctx.set_model(model, As)
ctx.set_sites(gen, model.rows_of(k, nn))
idx = np.arange(300000, 300000 + M)
lo, hi = np.zeros(M, np.int64), np.full(M, N - 1, np.int64)

def run(variant, eps=None):
    if eps is None:
        os.environ.pop('BMX_FAR_EPS', None)
    else:
        os.environ['BMX_FAR_EPS'] = repr(eps)
    ctx.set_variant(variant)
    ctx.set_tests(gen[idx], lo, hi)
    ctx.scan(); ctx.sync()
    first = [a.copy() for a in ctx.fetch()]
    ctx.scan(); ctx.sync()
    ms = ctx.last_scan_ms()
    out = [a.copy() for a in ctx.fetch()]
    if not all(np.array_equal(a, b) for a, b in zip(first, out)):
        print('  !! two scans of the same input differ bitwise (variant %d)' % variant)
    return out, ms

base, ms0 = run(10)
print('variant 10 (exact products): %.1f ms' % ms0)
for variant, epss in ((0, (0.0, 2e-2, 3e-2, 4e-2, 5e-2)), (3, (None,)), (4, (None,))):
    for eps in epss:
        got, ms = run(variant, eps)
        d = np.abs(got[0] - base[0])
        rel = d / np.maximum(np.abs(base[0]), 1e-300)
        mism = int(np.sum((got[1] != base[1]) | (got[2] != base[2]) | (got[3] != base[3])))
        print('variant %d eps %s: %.1f ms (x%.3f)  max|dCLR| %.2e  max rel %.2e  argmax mismatches %d  nSites equal %s'
              % (variant, eps, ms, ms0 / ms, d.max(), rel.max(), mism, np.array_equal(got[4], base[4])))
