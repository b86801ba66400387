"""Run the randomised GPU-vs-oracle scenario of tests/test_gpu_parity.py over many seeds."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.time()
for seed in range(lo, hi):
    try:
        T.test_randomised_scenarios_against_oracle(seed)
    except Exception as e:
        bad.append(seed)
        print('seed', seed, 'FAILED:', repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)
print('%d seeds, %d failures %s, %.0f s' % (hi - lo, len(bad), bad[:20], time.time() - t0))
