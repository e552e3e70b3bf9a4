"""which instance sets the time of a 20-step window of the bench protocol: per-instance sums of IPM iterations (attempt + repeat) over the window,
   from step-by-step fused launches.  usage (GPU box): python scripts/dev_window_tail.py [B|D]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bilevel-gait-gen_amd'))
import host, bench
WL = sys.argv[1] if len(sys.argv) > 1 else 'B'
B, K = (256, 20) if WL == 'B' else (512, 20)
cfg = host.load_config() if WL == 'B' else host.load_config('a1_config_distr_rejection')
st, ee = zip(*[(bench.config_b_instance if WL == 'B' else bench.config_d_instance)(cfg, b) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200); g.enable_fast_termination()
g.create_initial_run(st, ee); g.rti_advance(0, 5); g.synchronize()
for w in range(8 if WL == 'B' else 6):
    its = np.zeros((K, B)); fl = np.zeros((K, B), int)
    for k in range(K):
        g.rti_advance(5 + K * w + k, 1); g.synchronize()
        its[k] = g.stats()[:, 4] + 1; fl[k] = g.solve_flags()
    tot = its.sum(0)
    top = np.argsort(-tot)[:3]
    print('steps %3d..%3d  iterations per instance over the window: mean %.1f  max %d (instance %d)  99th pct %.0f' % (5 + K * w, 5 + K * (w + 1), tot.mean(), tot.max(), top[0], np.percentile(tot, 99)))
    for b in top:
        print('    instance %3d: total %d, per step %s, repeated attempts at steps %s' % (b, tot[b], its[:, b].astype(int).tolist(), [5 + K * w + k for k in range(K) if fl[k, b] & 4]))
