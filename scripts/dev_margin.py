"""Parity margins: relative primal / trajectory error against the oracle for the seeded Config-B instances after the cold
start + one RTI step (the case of tests/test_gpu_parity.py::test_batch_of_distinct_instances_matches_per_instance_oracle)
and over the first RTI steps (fixture test); prints the worst case so that solver changes can be judged against 1e-4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from srbm_loader import host
from oracle_py import OracleMPC, load_config
import bench
cfg = load_config()
B = int(os.environ.get('MARGIN_B', 24))
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
g.create_initial_run(states, ees.reshape(B, 12))
rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
os_ = []
for b in range(B):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); o.initial_run(states[b], ees[b]); os_.append(o)
worst = 0
for step in range(int(os.environ.get("MARGIN_STEPS", 4))):
    if step == 0:
        g.get_real_time_update(states, 0.0, ees.reshape(B, 12))
        for b in range(B): os_[b].rti(states[b], 0.0, ees[b])
    else:
        t = step * cfg['integrator_dt']
        tr = g.trajectory_states()
        eev = np.array([[[os_[b].ee_value(e, 1, c, t) for c in range(3)] for e in range(4)] for b in range(B)])
        for b in range(B): os_[b].rti(os_[b].states()[1], t, eev[b])
        g.get_real_time_update(tr[:, 1, :], t, eev.reshape(B, 12))
    xs = g.qp_solution()
    errs = np.array([rel(xs[b, :os_[b].sizes()['n']], os_[b].x()) for b in range(B)])
    worst = max(worst, errs.max())
    print('step', step, 'max rel primal err %.2e (inst %d)  median %.2e  status gpu %s' % (errs.max(), errs.argmax(), np.median(errs), np.unique(g.status()[0])))
print('worst %.2e' % worst)
