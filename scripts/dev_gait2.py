import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from test_gpu_gait import run_pair, host
np.set_printoptions(precision=5, linewidth=220, suppress=False)
cfg, g, o, state, ee, t = run_pair('a1_configuration', 5)
go = o.gait_gradient(); step_o, new_o = o.gait_optimize(t); nv = len(go)
gait = host.BatchGaitOptimizer(g); gait.compute_gradient(); gg, valid = gait.gradient(); gait.optimize_contact_times(t)
st, pred = gait.lp_result(); step = gait.step()
xk, counts = gait.contact_times()
print('t', t, 'counts', counts[0])
print('xk    ', xk[0, :nv])
print('grad o', go); print('grad g', gg[0, :nv]); print('rel diff', np.abs(go - gg[0, :nv]).max() / np.abs(go).max())
print('step o', step_o[:nv]); print('step g', step[0, :nv])
