"""BASELINE config 4 as FILES through the product's multi-file command, whole genome, one process, one GPU:

    python BalLeRMixPlus_amd.py --inputs <list of 22 chromosome files> --spect <helper file> -o <directory>

22 input files -> 22 output files with the reference's stages per chromosome (InputData -> NeutralSFS.get_neut_probs ->
NormalizedBetaBinom -> Scan, BalLeRMix+_v1.py:777-799); the next file is parsed while the current one is scanned and written.
This script only prepares the inputs (not timed), runs that ONE command and reports its wall time next to the scan kernels' time
the command prints, and the MD5 of the concatenated outputs.

    python scripts/config4_cli_pipeline.py [total_snps=40000000] [workdir=/tmp/bmx_cfg4]"""
import contextlib
import hashlib
import io
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ballermixplus_amd import helpers, synth

total = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
work = sys.argv[2] if len(sys.argv) > 2 else '/tmp/bmx_cfg4'
os.makedirs(work, exist_ok=True)
sizes = synth.config4_sizes(total)

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
lst = os.path.join(work, 'inputs.txt')
with open(lst, 'w') as f:
    f.write('\n'.join(files) + '\n')
outdir = os.path.join(work, 'out')
print('inputs written + --getSpect on the concatenation: %.1f s (not part of the pipeline time)' % (time.time() - t0), flush=True)

cmd = [sys.executable, os.path.join(ROOT, 'BalLeRMixPlus_amd.py'), '--inputs', lst, '--spect', spect, '-o', outdir]
print('command:', ' '.join(cmd), flush=True)
t0 = time.time()
r = subprocess.run(cmd, capture_output=True, text=True)
wall = time.time() - t0
if r.returncode != 0:
    print(r.stdout[-2000:], r.stderr[-3000:])
    raise SystemExit(r.returncode)
if os.environ.get('BMX_TRACE'):          # the CLI's own time stamps (stderr), one per file
    print(r.stderr[-4000:])
m = re.search(r'selection table built (\d+) time\(s\), scan kernels ([0-9.]+) s', r.stdout)
kern = float(m.group(2)) if m else float('nan')
W = sum(sizes)
print('whole-genome file pipeline (one command, process start included): %d windows, %d files in -> %d files out: %.2f s wall = %.3f M '
      'windows/s; scan kernels %.2f s (wall / kernels = %.3f); selection table built %s time(s)'
      % (W, len(files), len(files), wall, W / wall / 1e6, kern, wall / kern, m.group(1) if m else '?'))
h = hashlib.md5()
for f in files:
    with open(os.path.join(outdir, os.path.basename(f) + '.out.txt'), 'rb') as g:
        h.update(g.read())
print('md5 of the concatenated outputs:', h.hexdigest(), '| rows:', W)
