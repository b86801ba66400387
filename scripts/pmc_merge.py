"""Collects the PMC summaries of scripts/pmc_collect.sh (gpurun_out/r04/<tag>/summary.txt) into profiles/:
    python scripts/pmc_merge.py c3 c4 c5 ...
copies each summary to profiles/r04_pmc_<tag>_summary.txt and writes the JSON entries they end with to
profiles/r04_pmc_calibration.json (what bench.py reads)."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = {}
for tag in sys.argv[1:]:
    src = os.path.join(ROOT, 'gpurun_out', 'r04', tag, 'summary.txt')
    txt = open(src).read()
    i = txt.index('\n{\n')
    cal.update(json.loads(txt[i:]))
    shutil.copy(src, os.path.join(ROOT, 'profiles', 'r04_pmc_%s_summary.txt' % tag))
with open(os.path.join(ROOT, 'profiles', 'r04_pmc_calibration.json'), 'w') as f:
    json.dump(cal, f, indent=1)
print('entries:', {k: (v['kernel'], v['build_id'], round(v['valu_per_64_evals'], 4)) for k, v in cal.items()})
