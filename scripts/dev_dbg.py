"""Diagnostic: one cold-start solve (stand-alone kernels) then one fused step on Config B / D; prints status, error bits, stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from srbm_loader import host
import bench
for wl in ('B', 'D'):
    cfg = host.load_config() if wl == 'B' else host.load_config('a1_config_distr_rejection')
    B = 8
    inst = bench.config_b_instance if wl == 'B' else bench.config_d_instance
    states, ees = zip(*[inst(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states)
    g.create_initial_run(states, ees)
    st, er = g.status()
    print(wl, 'cold  status', st, 'err', er); print(np.array2string(g.stats()[:3], precision=3, max_line_width=200))
    g.rti_advance(0, 1); g.synchronize()
    st, er = g.status()
    print(wl, 'fused status', st, 'err', er); print(np.array2string(g.stats()[:3], precision=3, max_line_width=200))
