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
print('fine stamps 0..12:', np.round(out[:13] / 1e6, 2), 'total', out[:13].sum() / 1e6)
