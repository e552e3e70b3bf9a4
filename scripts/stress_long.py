"""Long open-loop runs of the bench protocol, certified through the sticky accumulators: every solve of every instance accounted for, error bits, statuses.
    python scripts/stress_long.py [B|D [steps [tol_step start_mu]]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srbm_loader import host, workloads
wl = sys.argv[1] if len(sys.argv) > 1 else 'B'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ts = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
mu = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
cfg = host.load_config() if wl == 'B' else host.load_config('a1_config_distr_rejection')
B = 256 if wl == 'B' else 512
inst = workloads.config_b_instance if wl == 'B' else workloads.config_d_instance
st, ee = zip(*[inst(cfg, b) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200); g.set_solver_step_rule(ts, mu)
g.create_initial_run(st, ee)
g.clear_status_accumulators()
t0 = time.perf_counter()
for k in range(0, steps, 50):
    g.rti_advance(k, min(50, steps - k))
g.synchronize()
el = time.perf_counter() - t0
acc = g.status_accumulated()
stt, err = g.status()
print('workload %s mode (%g, %g): %d instances x %d steps in %.2f s = %.1f k it/s; solves %d (expected %d), not solved %d in instances %s, max-iter %d, error bits (all steps) %d; final statuses %s; finite %s' % (
    wl, ts, mu, B, steps, el, B * steps / el / 1e3, int(acc[:, 1].sum()), B * steps, int(acc[:, 2].sum()), np.nonzero(acc[:, 2])[0][:10].tolist(), int(acc[:, 3].sum()),
    int(np.bitwise_or.reduce(acc[:, 0])), dict(zip(*np.unique(stt, return_counts=True))), bool(np.all(np.isfinite(g.qp_solution())))))
