"""Which instances set the launch time?  A K-step launch ends when its slowest workgroup ends: per-instance SUM of IPM
iterations over K = 20 steps (default tolerances), mean over instances vs max, and how persistent the slow instances are."""
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
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states)
g.create_initial_run(states, ees)
for i in range(9):
    g.get_real_time_update(states, np.zeros(B), ees)
g.rti_advance(0, 5); g.synchronize()
its = []
for i in range(5, 105):
    g.rti_advance(i, 1); g.synchronize()
    its.append(g.stats()[:, 4].copy())
its = np.array(its)                                   # [step][instance]
print('per-step: mean %.2f  mean of per-step max %.2f' % (its.mean(), its.max(1).mean()))
for w in range(5):
    s = its[20 * w:20 * (w + 1)].sum(0)
    top = np.argsort(-s)[:6]
    print('window %d: mean of sums %.1f  max %.0f (%.2fx)  p99 %.0f  p90 %.0f  slowest instances %s sums %s' %
          (w, s.mean(), s.max(), s.max() / s.mean(), np.percentile(s, 99), np.percentile(s, 90), top, s[top]))
tot = its.sum(0); top = np.argsort(-tot)[:8]
print('over 100 steps: mean %.1f max %.0f; slowest', top, tot[top])
for b in top[:3]:
    print('instance %d per-step iterations:' % b, its[:, b].astype(int))
