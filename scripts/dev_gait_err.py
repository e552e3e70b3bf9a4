"""GPU scratch driver: the gait segment of bench.py step by step -- which instance raises an error bit / loses its LP first, and what preceded it"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from srbm_loader import host
import bench
FREQ = 5
B = 256
cfg = host.load_config('a1_gait_opt_config', num_nodes=20, integrator_dt=0.05)
sc, ec = zip(*[bench.config_c_instance(cfg, b) for b in range(B)])
sc, ec = np.array(sc), np.array(ec).reshape(B, 12)
gm = host.BatchMPC(cfg, B)
gm.set_state_trajectory_warm_start(sc)
gm.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
if 'AB_STEP' in os.environ: gm.set_solver_step_rule(float(os.environ['AB_STEP']), float(os.environ.get('AB_MU', 0)))
gm.create_initial_run(sc, ec)
gait = host.BatchGaitOptimizer(gm)
bad_seen = set()
for r in range(0, 36):
    gait.rti_advance(r, 1, FREQ); gm.synchronize()
    st, err = gm.status(); acc = gm.status_accumulated()
    lp, pred = gait.lp_result()
    grad, valid = gait.gradient()
    kind = 'LS' if (r % FREQ == 0 and r > 0) else ('GO' if (r + 1) % FREQ == 0 and r > 0 else 'rti')
    bad = np.nonzero((acc[:, 0] != 0) | (st > 1) | ((kind == 'GO') & (lp != 0)))[0]
    print('run %2d %-3s statuses %s  err insts %s  lp!=0 %s  valid %d' % (r, kind, dict(zip(*np.unique(st, return_counts=True))), list(np.nonzero(acc[:, 0])[0][:6]), list(np.nonzero(lp)[0][:6]) if kind == 'GO' else '-', int(valid.sum())))
    for b in bad:
        if b not in bad_seen:
            bad_seen.add(b)
            xk, counts = gait.contact_times(); step = gait.step()
            print('   first trouble at instance %d: status %d err %d acc %s lp %d counts %s' % (b, st[b], err[b], acc[b], lp[b], counts[b]))
            print('   xk  ', np.round(xk[b, :counts[b].sum()], 4)); print('   step', step[b, :counts[b].sum()]); print('   grad', grad[b, :counts[b].sum()])
