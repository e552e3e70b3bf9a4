"""Stress run of the closed-loop harness: Config B and Config D batches, pushes drawn per instance, several hundred steps;
prints status / error-bit histograms every 50 steps and the spread of the plant states."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from srbm_loader import host
import bench
for wl, steps in (('B', 300), ('D', 150)):
    cfg = host.load_config() if wl == 'B' else host.load_config('a1_config_distr_rejection')
    B = 256
    inst = bench.config_b_instance if wl == 'B' else bench.config_d_instance
    states, ees = zip(*[inst(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.create_initial_run(states, ees)
    g.plant_set_state(states)
    rng = np.random.default_rng(11)
    imp = np.zeros((B, 6)); imp[:, 0:2] = np.clip(rng.normal(0, 2.5, (B, 2)), -7.5, 7.5); imp[:, 5] = rng.normal(0, 0.2, B)
    g.plant_set_push(rng.uniform(0.0, 1.0, B), imp)
    for i0 in range(0, steps, 50):
        g.closed_loop_advance(i0, 50, 10, True); g.synchronize()
        st, err = g.status(); x = g.plant_state()
        print(wl, 'steps', i0 + 50, 'status', dict(zip(*np.unique(st, return_counts=True))), 'err', dict(zip(*np.unique(err, return_counts=True))),
              'finite', bool(np.all(np.isfinite(x))), 'z range %.3f..%.3f' % (x[:, 2].min(), x[:, 2].max()), 'iters %.1f' % g.stats()[:, 4].mean(), flush=True)
