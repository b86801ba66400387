"""Where the scan time goes, by the range of A: config-3 chromosome, 65536 consecutive test sites,
the default x / alpha grids, sub-lists of the default A grid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ballermixplus_amd import engine as eng, synth
from ballermixplus_amd.hostmodel import Grids

N, n, M = 1000000, 100, 65536
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
phys, gen, k, nn = synth.synth_chromosome(N, n, 1)
spect = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}
grid = Grids(None, None, False, False, None, None)
xs, ab, As = grid.scan_order()
model = eng.ModelArrays('B2', int(k.min()), [n], spect, {n: 1.0}, xs, ab)
idx = np.arange(300000, 300000 + M)
lo, hi = np.zeros(M, np.int64), np.full(M, N - 1, np.int64)
rows = model.rows_of(k, nn)
As_sorted = sorted(As)
print('A grid:', As_sorted)
zcut = 18.420680743952367
groups = [('all', As), ('A<1e3', [a for a in As if a < 1e3]), ('1e3<=A<1e4', [a for a in As if 1e3 <= a < 1e4]),
          ('1e4<=A<1e5', [a for a in As if 1e4 <= a < 1e5]), ('A>=1e5', [a for a in As if a >= 1e5])]
if len(sys.argv) > 2 and sys.argv[2] == 'each':
    groups = [('all', As)] + [('A=%g' % a, [a]) for a in As_sorted]
for name, sub in groups:
    ctx = eng.Context(0)
    ctx.set_model(model, sub)
    ctx.set_sites(gen, rows)
    ctx.set_variant(variant)
    ctx.set_tests(gen[idx], lo, hi)
    ctx.scan(); ctx.sync(); ctx.scan(); ctx.sync()
    sites = sum(float(np.mean(np.searchsorted(gen, gen[idx] + zcut / a, 'right') - np.searchsorted(gen, gen[idx] - zcut / a, 'left'))) for a in sub)
    print('%-12s %2d values: %8.2f ms   %9.1f sites per window (summed over the A values)' % (name, len(sub), ctx.last_scan_ms(), sites))
    ctx.close()
