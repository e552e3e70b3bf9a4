"""GPU scratch driver: time of the gait segment of bench.py with the candidates of the line search on either kernel set, step rule on / off"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, ctypes as C
from srbm_loader import host
import bench
FREQ, B, STEPS = 5, 256, 30
cfg = host.load_config('a1_gait_opt_config', num_nodes=20, integrator_dt=0.05)
sc, ec = zip(*[bench.config_c_instance(cfg, b) for b in range(B)])
sc, ec = np.array(sc), np.array(ec).reshape(B, 12)
for rule in ((1e-5, 0.1),):
    for ks in (0, 1, 0, 1):
        gm = host.BatchMPC(cfg, B)
        gm.set_state_trajectory_warm_start(sc)
        gm.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
        gm.set_solver_step_rule(*rule)
        gm.create_initial_run(sc, ec)
        gait = host.BatchGaitOptimizer(gm)
        gm.L.srbm_gait_debug_candidates.restype = C.c_void_p
        ls = C.c_void_p(gm.L.srbm_gait_debug_candidates(gait.g))
        assert gm.L.srbm_set_kernel_set(ls, ks) == 0
        gait.rti_advance(0, 6, FREQ); gm.synchronize(); gm.clear_status_accumulators()
        t0 = time.perf_counter(); gait.rti_advance(6, STEPS, FREQ); gm.synchronize(); el = time.perf_counter() - t0
        acc = gm.status_accumulated()
        c4 = (C.c_longlong * 4)(); gm.L.srbm_get_solver_counters(ls, c4)
        print('   main batch counters', gm.solver_counters(), ' candidates: solves %d step rule %d low tried %d failed %d' % tuple(c4))
        print('step rule %s candidates on kernel set %d: %.3f ms per step, err bits %d, not solved %d' % (rule, ks, 1e3 * el / STEPS, int(np.bitwise_or.reduce(acc[:, 0])), int(acc[:, 2].sum())), flush=True)
        del gait, gm
