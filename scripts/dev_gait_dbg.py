"""GPU scratch driver: dH/dtheta of the N = 50 gait configuration at run 4 for several gap tolerances of the device solver, beside the
oracle's (every step re-synchronised to the oracle's trajectory)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_gpu_gait as t
from srbm_loader import host
np.set_printoptions(precision=4, linewidth=220, suppress=True)
go = None
for tol in (1e-15, 1e-13, 1e-11, 1e-9):
    cfg, g, o, state, ee, tt = t.run_pair('a1_gait_opt_config', 5, tol=tol)
    if go is None:
        go = o.gait_gradient(); print('oracle', go)
    gait = host.BatchGaitOptimizer(g); gait.compute_gradient(); gg, valid = gait.gradient()
    print('tol %.0e valid %d' % (tol, valid[0]), gg[0, :len(go)])
