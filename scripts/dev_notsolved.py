"""Config D (N = 50, 512 instances, pushes): the solves of the open-loop protocol that do not end Solved -- which status, and what does the
oracle's solver say about the SAME QP (srbm_export_qp)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
from oracle_py import qp_solve
import bench
cfg = host.load_config('a1_config_distr_rejection')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
SKIP = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # steps whose failures are not reported (the bench warms up for 5)
states, ees = zip(*[bench.config_d_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states)
g.create_initial_run(states, ees)
N = cfg['num_nodes']; nx = 12 * (N + 1)
found = 0
for i in range(STEPS):
    g.rti_advance(i, 1); g.synchronize()
    st, err = g.status()
    bad = np.where(st > 1)[0] if i >= SKIP else []
    for b in bad:
        sz = g.sizes()[b]; stats = g.stats()[b]
        A, bb, P, q = g.export_qp(b)
        nsamp, ntd = int(sz[7]), int(sz[6])
        cones = [c for c in [(0, nx), (1, 2 * nsamp), (1, 4 * nsamp), (1, 2 * (N - 3) * 8), (0, ntd), (0, 8)] if c[1] > 0]
        r = qp_solve(P, q, A, bb, cones, tol_gap=1e-15, tol_feas=1e-10)
        print('step %3d instance %3d: device status %d err %d iters %d gap %.1e res_p %.1e res_d %.1e | oracle on the same QP: status %d iters %d' %
              (i, b, st[b], err[b], stats[4], stats[7], stats[5], stats[6], r['status'], r['iters']))
        found += 1
    if found >= 12: break
print('not solved found:', found, 'in', i + 1, 'steps')
