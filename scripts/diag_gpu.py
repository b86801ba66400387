"""Step-by-step timing of the GPU path on Example 1 (diagnostic, prints as it goes)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
t0 = time.time()
def say(*a):
    print('[%7.2fs]' % (time.time() - t0), *a, flush=True)
from ballermixplus_amd import engine, _lib
say('lib loaded, devices', _lib.lib().bmx_device_count())
import cases
argv, gold = cases.ALL_CASES['ex1_B2']
opt, case, ts = cases.host_side(argv)
say('host side done', len(ts))
sel = engine.NormalizedBetaBinom(case.data, case.grid, False, False, False)
sel.bind(case.neut)
say('K1 + sites resident')
psel, R = sel.ctx.fetch_lut()
say('lut fetched', psel.shape, float(np.nanmax(R)))
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sel.ctx.set_tests(ts.test_gen[:nt], ts.lo[:nt], ts.hi[:nt])
say('tests set')
sel.ctx.scan(); sel.ctx.sync()
say('scan done, kernel ms', sel.ctx.last_scan_ms())
print(sel.ctx.fetch())
sel.ctx.set_tests(ts.test_gen, ts.lo, ts.hi)
sel.ctx.scan(); sel.ctx.sync()
say('full scan done, kernel ms', sel.ctx.last_scan_ms())
