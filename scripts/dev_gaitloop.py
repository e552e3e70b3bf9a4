import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from test_gpu_gait import host, OracleMPC, load_config, EE0, relerr
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'a1_configuration'
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
F = 5
cfg = load_config(cfgname); s0 = np.array(cfg['srb_init'], float)
g = host.BatchMPC(cfg, 2); g.set_state_trajectory_warm_start(s0); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
o = OracleMPC(cfg); o.set_warmstart(s0)
g.create_initial_run(s0, EE0); o.initial_run(s0, EE0)
gait = host.BatchGaitOptimizer(g)
dt = cfg['integrator_dt']; ready = False
for run in range(R):
    t = run * dt
    state = o.states()[1]
    ee = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
    what = 'rti'
    if run % F == 0 and run > 0 and ready:
        k, costs = o.gait_line_search(state, t, ee); ready = False; what = 'LS k=%d' % k
    elif (run + 1) % F == 0 and run > 0:
        o.rti(state, t, ee)
        gr = o.gait_gradient()
        if gr is not None: o.gait_optimize(t); ready = True
        else: ready = False
        what = 'rti+gaitopt ready=%s' % ready
    else:
        o.rti(state, t, ee); ready = False
    gait.rti_advance(run, 1, F); g.synchronize()
    tr = g.trajectory_states()[0]
    kg = g.knots(0)
    dk = max(np.abs(kg['times'][e, :o.knots(e)['K']] - o.knots(e)['times']).max() if kg['nk'][e] == o.knots(e)['K'] else 9.9 for e in range(4))
    print(run, what, 'status', g.status()[0][0], o.stats()['status'], 'traj rel %.2e' % relerr(tr, o.states()), 'knot time diff %.2e' % dk)
