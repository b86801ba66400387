"""Registers / scratch / LDS of every kernel in ballermixplus_amd/csrc/bmxscan.gfx950.s (`make -C ballermixplus_amd/csrc asm`),
from the code-object metadata.   python scripts/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, 'ballermixplus_amd', 'csrc', 'bmxscan.gfx950.s')).read()
md = s[s.index('amdhsa.kernels:'):]
rows = []
for blk in md.split('  - .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    try:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dn = name
    dn = dn.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    dn = re.sub(r'\((anonymous namespace)?[^<>]*\)$', '', dn)
    rows.append((dn, g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size')))
print('%-52s %5s %5s %8s %8s %8s' % ('kernel', 'VGPR', 'SGPR', 'v-spill', 's-spill', 'scratch'))
for r in sorted(rows):
    print('%-52s %5s %5s %8s %8s %8s' % r)
