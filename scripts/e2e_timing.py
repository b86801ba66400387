"""End-to-end timing of the drop-in CLI on the synthetic 1M-SNP chromosome (config 3) and the
PCIe-inclusive rate of the one-shot host-buffer entry point."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ballermixplus_amd import synth, engine, helpers
from ballermixplus_amd.hostmodel import Grids

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
phys, gen, k, nn = synth.synth_chromosome(N, 100, 1)
inp, spect, out = '/tmp/bmx_e2e_in.txt', '/tmp/bmx_e2e_spect.txt', '/tmp/bmx_e2e_out.txt'
t = time.time(); synth.write_input(inp, phys, gen, k, nn); print('write input     %.2f s' % (time.time() - t))
t = time.time(); helpers.getSpect(inp, spect, False, False); print('--getSpect      %.2f s' % (time.time() - t))
t = time.time()
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'BalLeRMixPlus_amd.py'),
                '-i', inp, '--spect', spect, '-o', out], check=True, stdout=subprocess.DEVNULL)
dt = time.time() - t
print('CLI end to end  %.2f s  (%d windows -> %.0f windows/s incl. process start, parse, K1, scan, format, write)' % (dt, N, N / dt))
print(open(out).readline().strip()); print(open(out).readlines()[N // 2].strip())
# one-shot host-buffer call
grid = Grids(None, None, False, False, None, None)
xs, ab, As = grid.scan_order()
sp = {(a, b): f for a, b, f in synth.spect_from_counts(k, nn)}
model = engine.ModelArrays('B2', int(k.min()), [100], sp, {100: 1.0}, xs, ab)
ctx = engine.Context(0); ctx.set_model(model, As)
rows = model.rows_of(k, nn)
for rep in range(2):
    t = time.time()
    ctx.set_sites(gen, rows)
    ctx.set_tests(gen, np.zeros(N, np.int64), np.full(N, N - 1, np.int64))
    ctx.scan(); res = ctx.fetch()
    dt = time.time() - t
print('host buffers in -> host buffers out (H2D + locate + scan + D2H): %.3f s = %.0f windows/s; scan kernel alone %.1f ms'
      % (dt, N / dt, ctx.last_scan_ms()))
