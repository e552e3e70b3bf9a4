"""per 20-step window of the Config-B bench protocol: time, IPM factorisations per solve, attempts / repeats, solves ended by the step rule
   usage (GPU box): python scripts/dev_windows.py [windows]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bilevel-gait-gen_amd'))
import host, bench
B, K = 256, 20
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = host.load_config()
st, ee = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200); g.enable_fast_termination()
g.create_initial_run(st, ee); g.rti_advance(0, 5); g.synchronize()
prev_w = g.work_counters(); prev_c = g.solver_counters()
for w in range(nw):
    t0 = time.perf_counter(); g.rti_advance(5 + K * w, K); g.synchronize(); el = time.perf_counter() - t0
    wc = g.work_counters(); c = g.solver_counters()
    its = (wc[0] - prev_w[0]) / (B * K)
    print('steps %3d..%3d  %.2f ms  %.3f ms/step  factorisations per solve %.2f  attempts %d repeated %d  step rule %d' % (
        5 + K * w, 5 + K * (w + 1), 1e3 * el, 1e3 * el / K, its, c['low_tried'] - prev_c['low_tried'], c['low_failed'] - prev_c['low_failed'], c['step_rule'] - prev_c['step_rule']), flush=True)
    prev_w, prev_c = wc, c
