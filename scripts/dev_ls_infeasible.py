"""Line-search candidates the device reports PrimalInfeasible: what do the oracle's solver and an independent phase-1 LP (scipy / HiGHS) say about
the SAME exported QP?  32 seeded Config-C instances, the protocol of tests/test_gpu_gait.py::test_gait_step_of_a_seeded_batch_of_32..."""
import os, sys
import numpy as np
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
from oracle_py import OracleMPC, load_config, qp_solve
from bench import config_c_instance
from scipy.optimize import linprog
cfg = load_config('a1_gait_opt_config', num_nodes=20, integrator_dt=0.05)
B, NSTEPS = 32, 4
states, ees = zip(*[config_c_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states)
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
gait = host.BatchGaitOptimizer(g)
gait.set_contact_times_from_trajectory()
steps = np.zeros((B, gait.NV))
for b in range(B):
    try:
        gr = os_[b].gait_gradient()
    except RuntimeError:
        gr = None
    if gr is not None:
        so, _ = os_[b].gait_optimize(t); steps[b, :len(gr)] = so[:len(gr)]
gait.set_step(steps)
imin, costs = gait.line_search(st_in, t, ee_in.reshape(B, 12))
cst, cerr = gait.candidate_status()
cand = gait.candidates()
sz = cand.sizes()
N = cfg['num_nodes']; nx = 12 * (N + 1)
bad = np.argwhere(cst == 3)
print('%d of %d candidates PrimalInfeasible on the device' % (len(bad), cst.size))
for b, c in bad[:10]:
    idx = b * 10 + c
    A, bb, P, q = cand.export_qp(idx)
    nsamp, ntd = int(sz[idx, 7]), int(sz[idx, 6])
    cones = [k for k in [(0, nx), (1, 2 * nsamp), (1, 4 * nsamp), (1, 2 * (N - 3) * 8), (0, ntd), (0, 8)] if k[1] > 0]
    r = qp_solve(P, q, A, bb, cones, tol_gap=1e-15, tol_feas=1e-10)
    # phase 1: min t  s.t.  A_eq x = b_eq,  A_in x - t <= b_in,  t >= -1   (t* <= 0  <=>  feasible; t* > 0: the smallest uniform violation)
    m, n = A.shape
    is_eq = np.zeros(m, bool); off = 0
    for nn, d in cones:
        if nn == 0: is_eq[off:off + d] = True
        off += d
    Aeq, beq, Ain, bin_ = A[is_eq], bb[is_eq], A[~is_eq], bb[~is_eq]
    cvec = np.zeros(n + 1); cvec[-1] = 1
    res = linprog(cvec, A_ub=np.hstack([Ain, -np.ones((Ain.shape[0], 1))]), b_ub=bin_, A_eq=np.hstack([Aeq, np.zeros((Aeq.shape[0], 1))]), b_eq=beq,
                  bounds=[(None, None)] * n + [(-1, None)], method='highs')
    viol_or = np.maximum(A[~is_eq] @ r['x'] - bb[~is_eq], 0).max() if r['status'] <= 2 else float('nan')
    print('instance %2d candidate %d: oracle status %d (iters %d, its worst inequality violation %.1e, equality residual %.1e) | phase-1 LP: status %d, smallest uniform violation t* = %.3e' %
          (b, c, r['status'], r['iters'], viol_or, np.abs(Aeq @ r['x'] - beq).max() if r['status'] <= 2 else float('nan'), res.status, res.fun if res.status == 0 else float('nan')))
