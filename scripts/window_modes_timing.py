import sys, time, numpy as np
sys.path.insert(0, '.')
from ballermixplus_amd import engine, synth
from ballermixplus_amd.hostmodel import Grids
N, n = 400000, 100
phys, gen, k, nn = synth.synth_chromosome(N, n, 1)
sp = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}
g = Grids(None, None, False, False, None, None); xs, ab, As = g.scan_order()
m = engine.ModelArrays('B2', 1, [n], sp, {n: 1.0}, xs, ab)
c = engine.Context(0); c.set_model(m, As); c.set_sites(gen, m.rows_of(k, nn))
idx = np.arange(N)
for r in (5, 50, 500, 5000):
    lo = np.maximum(0, idx - r); hi = np.minimum(N - 1, idx + r + 1)
    c.set_tests(gen, lo, hi); c.scan(); c.sync(); c.scan(); c.sync()
    print('-w', r, 'kernel ms', round(c.last_scan_ms(), 1), 'windows/s', round(N / c.last_scan_ms() * 1e3))
# fixed physical windows, non centred-ish: window entirely to the right of the test site
lo = np.minimum(idx + 10, N - 1); hi = np.minimum(N - 1, idx + 200)
c.set_tests(gen, lo, hi); c.scan(); c.sync()
print('offset windows kernel ms', round(c.last_scan_ms(), 1))
