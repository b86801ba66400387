"""Round-3 bring-up check of the prepared pipeline (variant 0) against the round-2 grouped kernel (variant 12) and the
per-site kernel (variant 2) on a block of config-3 windows: identical argmax / nSites, CLR to 1e-9, and the kernel times.
    python scripts/prep_check.py [--windows 65536] [--config 3|5] [--step 1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ballermixplus_amd import engine as eng, synth
from ballermixplus_amd.hostmodel import Grids

ap = argparse.ArgumentParser()
ap.add_argument('--snps', type=int, default=1000000)
ap.add_argument('--windows', type=int, default=65536)
ap.add_argument('--config', type=int, default=3)
ap.add_argument('--step', type=int, default=1)
ap.add_argument('--variants', default='12,0,2')
ap.add_argument('--reps', type=int, default=2)
ap.add_argument('--n-spread', type=int, default=0, help='sample sizes n - spread .. n drawn per site (bench.py --n-spread)')
a = ap.parse_args()
N, n = a.snps, (200 if a.config == 5 else 100)
phys, gen, k, nn = synth.synth_chromosome(N, n, 1)
sizes, props = [n], {n: 1.0}
if a.n_spread > 0:      # as bench.py: n_i uniform in [n - spread, n], counts rescaled
    rng = np.random.default_rng(77)
    n2 = rng.integers(n - a.n_spread, n + 1, N)
    k = np.where(k == nn, n2, np.maximum(1, np.minimum(n2 - 1, (k * n2) // nn)))
    nn = n2
spect = {(x, y): f for x, y, f in synth.spect_from_counts(k, nn)}
if a.n_spread > 0:
    sizes = sorted(set(nn.tolist()))
    props = {int(s_): float(sum(f for (x, y), f in spect.items() if y == s_)) for s_ in sizes}
grid = Grids(None, None, True, True, '100,10000,100', None) if a.config == 5 else Grids(None, None, False, False, None, None)
xs, ab, As = grid.scan_order()
model = eng.ModelArrays('B2', int(k.min()), sizes, spect, props, xs, ab)
ctx = eng.Context(0)
ctx.set_model(model, As)
ctx.set_sites(gen, model.rows_of(k, nn))
M = min(a.windows, (N - 300000) // a.step)
idx = 300000 + a.step * np.arange(M)
out = {}
for v in [int(x) for x in a.variants.split(',')]:
    ctx.set_variant(v)
    t0 = time.time()
    ctx.set_tests(gen[idx], np.zeros(M, np.int64), np.full(M, N - 1, np.int64))
    t_set = time.time() - t0
    pl = ctx.plan()
    ms = []
    for r in range(a.reps):
        ctx.scan(); ctx.sync()
        ms.append(ctx.last_scan_ms())
    out[v] = ctx.fetch()
    print('variant %2d %-40s set_tests %.3f s  scan %s ms  -> %.3f M windows/s  stream %.1f MB (%.0f B/window)'
          % (v, pl['kernel'], t_set, ' '.join('%.2f' % x for x in ms), M / min(ms) / 1e3, pl['stream_bytes'] / 1e6, pl['stream_bytes'] / M))
    sys.stdout.flush()
ref = out[list(out)[0]]
for v, got in out.items():
    same = all(np.array_equal(got[i], ref[i]) for i in (1, 2, 3, 4))
    rel = np.max(np.abs(got[0] - ref[0]) / np.maximum(np.abs(ref[0]), 1e-12))
    nd = int(np.sum((got[1] != ref[1]) | (got[2] != ref[2]) | (got[3] != ref[3]) | (got[4] != ref[4])))
    print('variant %2d vs %d: argmax/nSites identical: %s (%d differ), max rel dCLR %.3e, max abs %.3e' % (v, list(out)[0], same, nd, rel, np.max(np.abs(got[0] - ref[0]))))
