"""Kernel time of index-bounded windows (-w r: r sites to either side; and windows that do not contain the test site) at scale,
through the default plan (prepared pipeline) and the round-2 grouped kernel (variant 12), with a parity check between the two.
    python scripts/window_modes_timing.py"""
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
cases = [('-w %d' % r, np.maximum(0, idx - r), np.minimum(N - 1, idx + r + 1)) for r in (5, 50, 500, 5000)]
# windows entirely to the right of the test site
cases.append(('offset windows [i+10, i+200]', np.minimum(idx + 10, N - 1), np.minimum(N - 1, idx + 200)))
for name, lo, hi in cases:
    out = {}
    for v in (12, 0):
        c.set_variant(v)
        c.set_tests(gen, lo, hi); c.scan(); c.sync(); c.scan(); c.sync()
        out[v] = (c.last_scan_ms(), c.fetch(), c.plan()['kernel'])
    same = all(np.array_equal(a, b) for a, b in zip(out[0][1][1:], out[12][1][1:]))
    rel = np.max(np.abs(out[0][1][0] - out[12][1][0]) / np.maximum(np.abs(out[12][1][0]), 1e-9))
    print('%-30s round-2 kernel %7.1f ms (%.2f M windows/s) | %s %7.1f ms (%.2f M windows/s) | argmax/nSites identical: %s, max rel dCLR %.1e'
          % (name, out[12][0], N / out[12][0] / 1e3, out[0][2], out[0][0], N / out[0][0] / 1e3, same, rel))
