import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
import bench
cfg = host.load_config()
B = 256
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
g.create_initial_run(states, ees)
g.rti_advance(0, 5); g.synchronize()
mx = []
for i in range(5, 105):
    g.rti_advance(i, 1); g.synchronize()
    it = g.stats()[:, 4]
    mx.append((it.mean(), it.max(), np.percentile(it, 90)))
mx = np.array(mx)
print('mean of means %.2f  mean of max %.2f  mean of p90 %.2f' % tuple(mx.mean(0)))
print('hist of per-step max:', np.unique(mx[:, 1], return_counts=True))
it = g.stats()[:, 4]; print('last step hist', np.unique(it, return_counts=True))
st = g.stats(); sl = np.argsort(-it)[:5]
print('slowest: iters', it[sl], 'gap', st[sl, 7], 'resp', st[sl, 5], 'resd', st[sl, 6], 'status', g.status()[0][sl])
