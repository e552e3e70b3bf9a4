import importlib.util, os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
host.LIB_PATH = os.path.join(ROOT, 'bilevel-gait-gen_amd', os.environ.get('SRBM_PROF_LIB', 'libsrbm_rti_prof.so'))
from srbm_loader import workloads as bench
cfg = host.load_config()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 11
s0, ee = bench.config_b_instance(cfg, b)
g = host.BatchMPC(cfg, 1); g.set_state_trajectory_warm_start(s0)
if 'AB_STEP' in os.environ: g.set_solver_step_rule(float(os.environ['AB_STEP']), float(os.environ.get('AB_MU', 0)))
g.create_initial_run(s0, ee)
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for i in range(nsteps):
    g.rti_advance(i, 1)
g.synchronize()
print('counters', g.solver_counters()); print('status', g.status(), 'stats', g.stats()[0], 'sizes', g.sizes()[0])
out = np.zeros(384)
g.L.srbm_debug_get_trace(g.h, 0, out.ctypes.data_as(C.POINTER(C.c_double)))
for it in range(int(g.stats()[0, 4]) + 1):
    if it < 32: print(it, 'mu %.3e sigma %.3e alpha %.3e gap %.3e res_p %.2e res_d %.2e ir %d err %.2e' % tuple(out[8 * it:8 * it + 8]), 'alpha_aff %.3f |du_aff|/|u| %.2e  worst refinement row %d e2 %.2e' % tuple(out[256 + 4 * it:256 + 4 * it + 4]))
