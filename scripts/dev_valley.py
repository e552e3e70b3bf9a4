"""Where the device minimiser and the oracle solver's minimiser of the SAME exported QP differ by more than 1e-4: is it the
flat directions of the weakly convex QP?  For every seeded Config-B instance (cold start + one RTI step): max-norm
difference, objective difference, feasibility of both points."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from srbm_loader import host
from oracle_py import load_config, qp_solve
import bench
cfg = load_config()
B = int(os.environ.get('VALLEY_B', 64)); OFF = int(os.environ.get('VALLEY_OFF', 0))
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(OFF, OFF + B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
g.create_initial_run(states, ees)
g.get_real_time_update(states, 0.0, ees)
st, err = g.status(); sz = g.sizes(); xr = g.raw_qp_minimiser()
nx = (cfg['num_nodes'] + 1) * 12
rows = []
for b in range(B):
    n, m, ntd, nsamp = int(sz[b, 0]), int(sz[b, 1]), int(sz[b, 6]), int(sz[b, 7])
    A, bb, P, q = g.export_qp(b)
    cones = [c for c in [(0, nx), (1, 2 * nsamp), (1, 4 * nsamp), (1, 2 * (cfg['num_nodes'] - 3) * 8), (0, ntd), (0, 8)] if c[1] > 0]
    r = qp_solve(P, q, A, bb, cones, tol_gap=1e-15, tol_feas=1e-10)
    xd, xo = xr[b, :n], r['x']
    f = lambda x: 0.5 * x @ P @ x + q @ x
    def viol(x):
        res = A @ x - bb; v = 0.0; o = 0
        for is_nn, d in cones:
            v = max(v, (np.maximum(res[o:o + d], 0).max() if is_nn else np.abs(res[o:o + d]).max())); o += d
        return v
    d = np.abs(xd - xo); j = int(d.argmax())
    rows.append((d.max() / max(1.0, np.abs(xo).max()), (f(xd) - f(xo)) / max(1.0, abs(f(xo))), viol(xd), viol(xo), j, int(st[b]), r['status']))
rows = np.array(rows)
bad = np.where(rows[:, 0] > 1e-4)[0]
print('instances %d..%d: median rel diff %.2e, max %.2e; above 1e-4: %s' % (OFF, OFF + B - 1, np.median(rows[:, 0]), rows[:, 0].max(), (bad + OFF).tolist()))
for b in bad:
    print(' inst %d: rel diff %.2e at variable %d (%s)  objective diff (dev - orc)/|f| %.2e  violation dev %.1e orc %.1e  status dev %d orc %d' % (
        b + OFF, rows[b, 0], rows[b, 4], 'state' if rows[b, 4] < nx else 'spline var', rows[b, 1], rows[b, 2], rows[b, 3], rows[b, 5], rows[b, 6]))
print('objective diff over all: max |.| %.2e ; violation dev max %.1e, orc max %.1e' % (np.abs(rows[:, 1]).max(), rows[:, 2].max(), rows[:, 3].max()))
