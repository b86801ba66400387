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

# the CLI's stages in this process, timed one by one (imports done, GPU context warm)
import hashlib
from ballermixplus_amd.hostmodel import InputData, NeutralSFS
from ballermixplus_amd import scan as scanmod
import io, contextlib
quiet = io.StringIO()
with contextlib.redirect_stdout(quiet):
    t0 = time.time(); data = InputData(inp, False, False, False, 1, phys=False, Rrate=1e-6); t1 = time.time()
    neut = NeutralSFS(spect, False, False, False); neut.get_neut_probs(data); t2 = time.time()
    sel = engine.NormalizedBetaBinom(data, grid, False, False, False, device=0).bind(neut); t3 = time.time()
    out2 = '/tmp/bmx_e2e_out2.txt'
    scanmod.Scan(data, neut, sel, grid, out2); t4 = time.time()
    # the batch path (scan everything, fetch, then format and write) for comparison
    out3 = '/tmp/bmx_e2e_out3.txt'
    ts = scanmod.sites_alpha(data, 1)
    t5 = time.time()
    res = engine.scan_batch(sel, ts.arrays[1], ts.arrays[2], ts.arrays[3]); t6 = time.time()
    scanmod.write_rows(out3, ts, res, sel); t7 = time.time()
print('stages (s): parse %.3f | neutral model %.3f | K1 + site upload %.3f | streamed scan+write %.3f (kernel %.3f) || batch: scan+fetch %.3f, format+write %.3f'
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, sel.ctx.last_scan_ms() / 1e3, t6 - t5, t7 - t6))
md5 = lambda p: hashlib.md5(open(p, 'rb').read()).hexdigest()
print('output files identical (CLI / streamed / batch):', md5(out) == md5(out2) == md5(out3), md5(out))
