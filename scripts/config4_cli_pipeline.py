"""BASELINE config 4 through the drop-in file pipeline, whole genome, one process, one GPU: 22 input FILES ->
22 output FILES, with the reference's stages per chromosome (InputData -> NeutralSFS.get_neut_probs ->
NormalizedBetaBinom -> Scan, BalLeRMix+_v1.py:777-799).  The host stages of chromosome c+1 (parse, neutral
probabilities, row indices) run on a helper thread while chromosome c is being scanned and written (the native calls
release the GIL), so the wall time approaches the sum of the scan kernels.

    python scripts/config4_cli_pipeline.py [total_snps=40000000] [workdir=/tmp/bmx_cfg4]

Prints the wall time of the pipeline (input files already on disk, helper file already made) next to the sum of the
scan kernels' times, and the MD5 of the concatenated outputs."""
import hashlib
import io
import os
import sys
import threading
import time
import contextlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ballermixplus_amd import engine, helpers, scan as scanmod, synth
from ballermixplus_amd.hostmodel import Grids, InputData, NeutralSFS

total = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
work = sys.argv[2] if len(sys.argv) > 2 else '/tmp/bmx_cfg4'
os.makedirs(work, exist_ok=True)
sizes = synth.config4_sizes(total)

# ---- inputs on disk (not timed): 22 chromosome files + the helper file of their concatenation
t0 = time.time()
cat = os.path.join(work, 'all.txt')
files = []
with open(cat, 'w') as fc:
    fc.write('physPos\tgenPos\tx\tn\n')
    for c, N in enumerate(sizes, start=1):
        phys, gen, k, nn = synth.synth_chromosome(N, 100, c)
        path = os.path.join(work, 'chr%d.txt' % c)
        synth.write_input(path, phys, gen, k, nn)
        with open(path) as f:
            next(f)
            fc.write(f.read())
        files.append(path)
spect = os.path.join(work, 'spect.txt')
with contextlib.redirect_stdout(io.StringIO()):
    helpers.getSpect(cat, spect, False, False)
os.remove(cat)
print('inputs written + --getSpect on the concatenation: %.1f s (not part of the pipeline time)' % (time.time() - t0), flush=True)

grid = Grids(None, None, False, False, None, None)
quiet = io.StringIO()


def host_stage(path):
    """everything the CLI does before NormalizedBetaBinom: parse the file, attach the neutral probabilities"""
    data = InputData(path, False, False, False, 1, phys=False, Rrate=1e-6)
    neut = NeutralSFS(spect, False, False, False)
    neut.get_neut_probs(data)
    return data, neut


# warm-up: library load, HIP context (what a resident service pays once)
engine.Context(0).close()

sys.stdout = quiet            # the stages' progress messages (two threads print): silenced for the whole pipeline
t_start = time.time()
kernel_ms = 0.0
nxt = {}


def prefetch(i):
    nxt[i] = host_stage(files[i])


ctx = None
th = threading.Thread(target=prefetch, args=(0,))
th.start()
outs = []
for i in range(len(files)):
    th.join()
    data, neut = nxt.pop(i)
    if i + 1 < len(files):
        th = threading.Thread(target=prefetch, args=(i + 1,))
        th.start()
    out = os.path.join(work, 'chr%d.out.txt' % (i + 1))
    sel = engine.NormalizedBetaBinom(data, grid, False, False, False, device=0)
    scanmod.Scan(data, neut, sel, grid, out, keep_results=False, reuse_ctx=ctx)     # one context, one table for the genome
    ctx = sel.ctx
    kernel_ms += ctx.last_scan_ms()
    outs.append(out)
wall = time.time() - t_start
sys.stdout = sys.__stdout__
W = sum(sizes)
print('whole-genome file pipeline: %d windows, %d files in -> %d files out: %.2f s wall = %.3f M windows/s; scan kernels %.2f s '
      '(wall / kernels = %.3f)' % (W, len(files), len(outs), wall, W / wall / 1e6, kernel_ms / 1e3, wall / (kernel_ms / 1e3)))
h = hashlib.md5()
for o in outs:
    with open(o, 'rb') as f:
        h.update(f.read())
print('md5 of the concatenated outputs:', h.hexdigest(), '| rows:', sum(sizes))
