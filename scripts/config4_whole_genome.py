"""BASELINE config 4 on the GPUs of this process group: synthetic whole genome, 40M SNPs over 22
chromosomes (GRCh37 proportions), n = 100, default grid, every SNP a test site.  Chromosomes are
dealt to ranks longest-first (each rank scans whole chromosomes; no data-path collective), results
are all-gathered per chromosome.  Prints total windows/s.  Single GPU: `python scripts/config4_whole_genome.py`;
N GPUs: `python -m torch.distributed.run --nproc-per-node N scripts/config4_whole_genome.py`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ballermixplus_amd import distributed, engine, synth
from ballermixplus_amd.hostmodel import Grids

total = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
world = distributed.World.from_env()
sizes = synth.config4_sizes(total)
order = np.argsort(sizes)[::-1]                      # longest processing time first
load = [0] * world.size
mine = []
for c in order:
    r = int(np.argmin(load))
    load[r] += sizes[c]
    if r == world.rank:
        mine.append(int(c))
grid = Grids(None, None, False, False, None, None)
xs, ab, As = grid.scan_order()
ctx = engine.Context(world.local_rank)
t_gen = t_scan = 0.0
windows = 0
chk = 0.0
for c in mine:
    t0 = time.time()
    phys, gen, k, nn = synth.synth_chromosome(sizes[c], 100, c + 1)
    spect = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}     # per-chromosome spectrum (synthetic)
    model = engine.ModelArrays('B2', int(k.min()), [100], spect, {100: 1.0}, xs, ab)
    t_gen += time.time() - t0
    t0 = time.time()
    ctx.set_model(model, As)
    ctx.set_sites(gen, model.rows_of(k, nn))
    N = sizes[c]
    ctx.set_tests(gen, np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
    ctx.scan()
    clr, ix, ia, iA, ns = ctx.fetch()
    t_scan += time.time() - t0
    windows += N
    chk += float(clr.sum())
    print('rank %d chr%-2d %8d SNPs  scan kernel %.0f ms' % (world.rank, c + 1, N, ctx.last_scan_ms()), flush=True)
print('rank %d: %d windows in %.2f s (H2D + K1 + scan + D2H; synthetic data generation %.1f s not counted) = %.0f windows/s; checksum %.6f'
      % (world.rank, windows, t_scan, t_gen, windows / t_scan, chk))
world.finish()
