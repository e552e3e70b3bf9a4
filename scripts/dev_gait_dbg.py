import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle_py import OracleMPC, load_config
from srbm_loader import host
EE0 = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)
cfg = load_config(sys.argv[1]); F = 5
s0 = np.array(cfg['srb_init'], float)
g = host.BatchMPC(cfg, 2); g.set_state_trajectory_warm_start(s0)
o = OracleMPC(cfg); o.set_warmstart(s0)
g.create_initial_run(s0, EE0); o.initial_run(s0, EE0)
gait = host.BatchGaitOptimizer(g)
dt = cfg['integrator_dt']
np.set_printoptions(precision=5, linewidth=200, suppress=True)
for run in range(5):
    t = run * dt
    state = o.states()[1]
    ee = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
    if len(sys.argv) > 2: g.set_warm_start_trajectory([o.trajectory_record(host)] * 2)
    o.rti(state, t, ee)
    if run == 4:
        go = o.gait_gradient(); step_o, new_o = o.gait_optimize(t)
    gait.rti_advance(run, 1, F); g.synchronize()
nv = len(go)
gg, valid = gait.gradient()
xk, counts = gait.contact_times()
print('counts', counts[0], 'valid', valid, 'lp', gait.lp_result())
print('xk   ', xk[0, :nv]); print('orc ct', np.concatenate([o.contact_times(e)[0] for e in range(4)]))
print('grad dev', gg[0, :nv]); print('grad orc', go)
print('step dev', gait.step()[0, :nv]); print('step orc', step_o[:nv])
