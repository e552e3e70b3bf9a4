import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, ctypes as C
from srbm_loader import host
host.LIB_PATH = os.path.join(ROOT, 'bilevel-gait-gen_amd', 'libsrbm_rti_prof.so')
import bench
cfg = host.load_config(); B = 64
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.create_initial_run(states, ees)
g.rti_advance(0, 20); g.synchronize()
out = np.zeros(64); g.L.srbm_debug_get_profile2(g.h, 0, out.ctypes.data_as(C.POINTER(C.c_double)))
names = ['staging/F,R/Bu', 'rollout', 'costates', 'assemble+row duals', 'candidates', 'spline cost', 'ddp+Armijo', 'x update+states', 'spline nodes']
tot = out[32:41].sum()
for n, v in zip(names, out[32:41]): print('%-22s %10.0f cycles/step %5.1f%%' % (n, v / 30, 100 * v / tot))
print('total per step %.0f cycles = %.1f us' % (tot / 30, tot / 30 / 2400))
names1 = ['staging', 'horizon shift', 'bookkeeping', 'lin. point + column map', 'node records + samples', 'equality rows', 'node blocks + affine']
tot1 = out[44:51].sum()
print('kernel 1:')
for n, v in zip(names1, out[44:51]): print('%-24s %10.0f cycles/step %5.1f%%' % (n, v / 30, 100 * v / tot1))
print('total per step %.0f cycles = %.1f us' % (tot1 / 30, tot1 / 30 / 2400))
