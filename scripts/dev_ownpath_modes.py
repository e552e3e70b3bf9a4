"""which of the two opt-in solver settings moves the free-running path away from the oracle?  per-step count of instances beyond 1e-4 (tests/test_gpu_ownpath.py)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import test_gpu_ownpath as T
for mode in [(0.0, 0.0), (1e-5, 0.0), (0.0, 0.1), (1e-5, 0.1), (3e-6, 0.1)]:
    err, err_s, ok, ctr = T.own_path_run(True, B=64, steps=70, mode=mode)
    e = np.fmax(err, err_s)
    v = e[np.isfinite(e)]
    over = np.sum(np.where(np.isfinite(e), e, 0.0) > 1e-4, axis=1)
    print('mode', mode, 'median %.2e p90 %.2e p99 %.2e; steps 40-69: p99 %.2e max %.2e' % (np.median(v), np.percentile(v, 90), np.percentile(v, 99),
          np.nanpercentile(e[40:], 99), np.nanmax(e[40:])), ctr)
    print('   beyond 1e-4 per step:', over.tolist())
