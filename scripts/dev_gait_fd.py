"""dH/dtheta . step (the LP's predicted reduction) against the finite difference of the device MPC cost along the step, for the 32 seeded Config-C
instances of tests/test_gpu_gait.py (study behind test_gradient_predicts_the_cost_change_along_the_lp_step)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from oracle_py import OracleMPC, load_config
from srbm_loader import host
from bench import config_c_instance
cfg = load_config('a1_gait_opt_config', num_nodes=20, integrator_dt=0.05)
B, NSTEPS = 32, 4
states, ees = zip(*[config_c_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
os_ = []
for b in range(B):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); os_.append(o)
pool = ThreadPoolExecutor(16)
list(pool.map(lambda b: os_[b].initial_run(states[b], ees[b].reshape(4, 3)), range(B)))
g.create_initial_run(states, ees.reshape(B, 12))
dt = cfg['integrator_dt']
for i in range(NSTEPS):
    t = i * dt
    st_in = np.array([o.states()[1] for o in os_])
    ee_in = np.array([[[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)] for o in os_])
    g.set_warm_start_trajectory((host.Trajectory * B)(*[o.trajectory_record(host) for o in os_]))
    list(pool.map(lambda b: os_[b].rti(st_in[b], t, ee_in[b]), range(B)))
    g.get_real_time_update(st_in, t, ee_in.reshape(B, 12))
st = g.status()[0]
n = g.sizes()[:, 0].astype(float)
cost0 = g.cost().copy()
gait = host.BatchGaitOptimizer(g)
gait.set_contact_times_from_trajectory()
gait.compute_gradient()
gg, valid = gait.gradient()
gait.optimize_contact_times(t)
lp_st, pred = gait.lp_result()
step = gait.step()
xk, counts = gait.contact_times()
# the line search evaluates the cost of an RTI solve at x_k + (i / 10) s: with s = eps * step its first candidates are a finite difference along the
# step.  Every eps on a CLONE of the batch (a line search installs the winner's trajectory)
for eps in (1.0, 0.1, 0.01, 0.001):
    gc = g.clone()
    gaitc = host.BatchGaitOptimizer(gc)
    gaitc.set_contact_times_from_trajectory()
    gaitc.set_step(eps * step)
    imin, costs = gaitc.line_search(st_in, t, ee_in.reshape(B, 12))
    cst, cerr = gaitc.candidate_status()
    h = 0.1 * eps
    c = costs * n[:, None]                       # GetCost() / GetNumDecisionVars() -> cost
    fd1 = (c[:, 1] - c[:, 0]) / h
    fd2 = (-3 * c[:, 0] + 4 * c[:, 1] - c[:, 2]) / (2 * h)
    print('eps', eps)
    for b in range(B):
        nv = counts[b].sum()
        pr = gg[b, :nv] @ step[b, :nv]
        print('  inst %2d st %d valid %d lp %d  g.step %+.5e  fd1 %+.5e fd2 %+.5e  fd1/pred %.3f fd2/pred %.3f  cost0 %.6e c0 %.6e c1-c0 %.3e cand status %s' % (
            b, st[b], valid[b], lp_st[b], pr, fd1[b], fd2[b], fd1[b] / pr if pr != 0 else float('nan'), fd2[b] / pr if pr != 0 else float('nan'), cost0[b], c[b, 0], c[b, 1] - c[b, 0], cst[b, :3].tolist()))
    gaitc.close(); gc.close()
