"""Which instances set the time of a 20-step window of the bench protocol, and on what: per-instance factorisations per step (one step per launch),
the slowest instances with their per-step counts and attempt flags (a = attempt Solved, R = attempt repeated from the standard start, - = no attempt).
    python scripts/dev_window_tail.py [first_step [steps [tol_step start_mu]]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srbm_loader import host, workloads
first = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ts = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
mu = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
cfg = host.load_config()
B = 256
st, ee = zip(*[workloads.config_b_instance(cfg, b) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200); g.set_solver_step_rule(ts, mu)
g.create_initial_run(st, ee)
g.rti_advance(0, first); g.synchronize()
its = np.zeros((B, steps), int); fl = np.zeros((B, steps), int)
prev = None
for i in range(steps):
    g.rti_advance(first + i, 1); g.synchronize()
    its[:, i] = g.stats()[:, 4]; fl[:, i] = g.solve_flags()
tot = its.sum(axis=1)
print('window steps %d..%d mode (%g, %g): per-instance total iterations mean %.1f  median %.0f  p90 %.0f  max %d  (the fused launch ends with the max: %.2f x the mean)' % (
    first, first + steps - 1, ts, mu, tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), tot.max() / tot.mean()))
print('repeated attempts per instance: histogram', np.bincount(((fl & 4) != 0).sum(axis=1)))
for b in np.argsort(-tot)[:8]:
    print('inst %3d total %3d :' % (b, tot[b]), ' '.join('%2d%s' % (its[b, i], 'R' if fl[b, i] & 4 else ('a' if fl[b, i] & 2 else '-')) for i in range(steps)))
print('mean per step       :', ' '.join('%4.1f' % v for v in its.mean(axis=0)))
