"""Developer script: wall time of the device-resident RTI protocol, fused kernel vs one launch per phase (no oracle)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
import bench
cfg = host.load_config()
B = 256
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
for mode in ('fused', 'unfused', 'fused', 'unfused'):
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
    g.create_initial_run(states, ees)
    adv = g.rti_advance if mode == 'fused' else g.rti_advance_unfused
    adv(0, 5); g.synchronize()
    t0 = time.perf_counter()
    adv(5, 100); g.synchronize()
    el = time.perf_counter() - t0
    it, fl = g.work_counters()
    print(mode, '%.3f ms/step  %.0f it/s' % (1e3 * el / 100, B * 100 / el), 'statuses', np.unique(g.status()[0], return_counts=True))
