"""Where a cold whole-genome pass spends its time (BASELINE config 4 on one GPU): wall clock per phase around the calls bench.py's
cold pass makes -- context + table build (K1), host-side row lookup, site uploads, test-site arrays + uploads + location + planning,
the scans, the result transfer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ballermixplus_amd import engine, synth, _lib
from ballermixplus_amd.hostmodel import Grids

total = int(sys.argv[1]) if len(sys.argv) > 1 else 40000000
n = 100
sizes_c = synth.config4_sizes(total)
chroms = []
for cid, Nc in enumerate(sizes_c, 1):
    phys, gen, k, nn = synth.synth_chromosome(Nc, n, chrom=cid)
    chroms.append((gen, k, nn))
kk = np.concatenate([c[1] for c in chroms]); nk = np.concatenate([c[2] for c in chroms])
spect = {(a, b): f for a, b, f in synth.spect_from_counts(kk, nk)}
xs, ab, As = Grids(None, None, False, False, None, None).scan_order()
model = engine.ModelArrays('B2', int(kk.min()), [n], spect, {n: 1.0}, xs, ab)
M = sum(len(c[0]) for c in chroms)
host = np.empty(M, dtype=_lib.RECORD_DTYPE)

for rep in range(2):
    ph = {}
    def lap(name, t0):
        ph[name] = ph.get(name, 0.0) + time.perf_counter() - t0
    T0 = time.perf_counter()
    t = time.perf_counter(); ctx = engine.Context(0); lap('context', t)
    t = time.perf_counter(); ctx.set_model(model, As); lap('set_model (K1 table)', t)
    for ci, (gen, k, nn) in enumerate(chroms):
        Nc = len(gen)
        ctx.select_slot(ci)
        t = time.perf_counter(); rows = model.rows_of(k, nn); lap('host: rows_of', t)
        t = time.perf_counter(); ctx.set_sites(gen, rows); lap('set_sites', t)
        t = time.perf_counter(); lo = np.zeros(Nc, np.int64); hi = np.full(Nc, Nc - 1, np.int64); lap('host: window arrays', t)
        t = time.perf_counter(); ctx.set_tests(gen, lo, hi); lap('set_tests (upload, locate, plan)', t)
    t = time.perf_counter()
    for ci in range(len(chroms)):
        ctx.select_slot(ci); ctx.scan()
    lap('scan launches', t)
    t = time.perf_counter(); ctx.pack_records(out=host); lap('wait + records to host', t)
    tot = time.perf_counter() - T0
    print('pass %d: %.3f s = %.3f M windows/s' % (rep, tot, M / tot / 1e6))
    for k_, v in ph.items():
        print('   %-36s %7.3f s' % (k_, v))
    ctx.close()
