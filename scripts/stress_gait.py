"""Stress run (diagnostic): the controller loop with the gait step over many steps on a full batch; prints status / error histograms."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'B'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = host.load_config('a1_configuration' if wl == 'B' else 'a1_config_distr_rejection')
B = 256 if wl == 'B' else 512
mk = bench.config_b_instance if wl == 'B' else bench.config_d_instance
states, ees = zip(*[mk(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
g.create_initial_run(states, ees)
gait = host.BatchGaitOptimizer(g)
t0 = time.time()
hist = {}
for r0 in range(0, steps, 10):
    gait.rti_advance(r0, 10, 5); g.synchronize()
    st, err = g.status(); stats = g.stats()
    key = tuple(sorted({int(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))}.items()))
    print('run', r0 + 10, 'status', dict(key), 'err bits', int(np.bitwise_or.reduce(err)), 'n err', int((err != 0).sum()), 'max iters %d' % stats[:, 4].max(),
          'cost mean %.1f' % np.nanmean(stats[:, 1]), 'finite', bool(np.isfinite(g.trajectory_states()).all()))
print('elapsed %.1f s' % (time.time() - t0))
