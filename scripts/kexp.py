"""Kernel experiment runner: one config-3 style scan of a block of consecutive test sites, kernel time and a checksum.
    BMX_LIB_NAME=libbmx_prof.so python scripts/kexp.py [--windows 131072] [--variant 0] [--step 1] [--config 3|5] [--reps 3]
Used to compare library builds (BMX_LIB_NAME) and variants on the GPU box; prints one line per run."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ballermixplus_amd import engine as eng, synth
from ballermixplus_amd.hostmodel import Grids

ap = argparse.ArgumentParser()
ap.add_argument('--snps', type=int, default=1000000)
ap.add_argument('--windows', type=int, default=131072)
ap.add_argument('--variant', type=int, default=0)
ap.add_argument('--step', type=int, default=1)
ap.add_argument('--config', type=int, default=3)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--tag', default='')
a = ap.parse_args()
N, n = a.snps, (200 if a.config == 5 else 100)
phys, gen, k, nn = synth.synth_chromosome(N, n, 1)
spect = {(x, y): f for x, y, f in synth.spect_from_counts(k, nn)}
grid = Grids(None, None, True, True, '100,10000,100', None) if a.config == 5 else Grids(None, None, False, False, None, None)
xs, ab, As = grid.scan_order()
model = eng.ModelArrays('B2', int(k.min()), [n], spect, {n: 1.0}, xs, ab)
ctx = eng.Context(0)
ctx.set_variant(a.variant)
ctx.set_model(model, As)
ctx.set_sites(gen, model.rows_of(k, nn))
M = min(a.windows, (N - 300000) // a.step)
idx = 300000 + a.step * np.arange(M)
ctx.set_tests(gen[idx], np.zeros(M, np.int64), np.full(M, N - 1, np.int64))
ms = []
for r in range(a.reps):
    ctx.scan(); ctx.sync()
    ms.append(ctx.last_scan_ms())
clr, ix, ia, iA, ns = ctx.fetch()
print('%s lib=%s variant=%d windows=%d step=%d: best %.2f ms (%.3f M windows/s), all %s; sum CLR %.12g, sum lin %d, sum ns %d'
      % (a.tag, os.environ.get('BMX_LIB_NAME', 'libbmxscan.so'), a.variant, M, a.step, min(ms), M / min(ms) / 1e3,
         ' '.join('%.2f' % v for v in ms), float(np.sum(clr)), int(np.sum(ix.astype(np.int64) * 1000 + ia + iA * 1000000)), int(np.sum(ns))))
sys.stdout.flush()
