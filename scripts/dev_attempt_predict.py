"""which solves' lower-start attempts fail during the first 100 steps of the bench protocol, and what was known BEFORE the solve:
   the previous step's SQP step norm / Armijo step / IPM iterations.  usage (GPU box): python scripts/dev_attempt_predict.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'bilevel-gait-gen_amd'))
import host, bench
B = 256
cfg = host.load_config()
st, ee = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200); g.enable_fast_termination()
g.create_initial_run(st, ee); g.rti_advance(0, 5); g.synchronize()
prev = g.stats()
rows = []
for i in range(5, 125):
    g.rti_advance(i, 1); g.synchronize()
    fl = g.solve_flags(); s = g.stats()
    for b in range(B):
        if fl[b] & 2:
            rows.append((i, b, 1 if fl[b] & 4 else 0, prev[b, 3], prev[b, 0], prev[b, 4], s[b, 4], s[b, 3]))
    prev = s
r = np.array(rows)
fail = r[:, 2] == 1
print('attempts %d failed %d' % (len(r), fail.sum()))
print('failed attempts (step, instance, prev step norm, prev alpha, prev iters, iters of the accepted solve, this step norm):')
late = r[:, 0] >= 25
print('attempts from step 25 on: %d, failed %d' % (late.sum(), (late & fail).sum()))
for x in r[fail & late][:60]:
    print('  step %3d inst %3d  prev step norm %.3e  prev alpha %.3f  prev iters %2d  iters %2d  step norm %.3e' % (x[0], x[1], x[3], x[4], x[5], x[6], x[7]))
for name, col in (('prev step norm', 3), ('prev iters', 5)):
    for q in (50, 90, 99, 99.9):
        print('%-16s pct %5.1f: ok %.3e   failed-min %.3e failed-median %.3e' % (name, q, np.percentile(r[~fail][:, col], q), r[fail][:, col].min(), np.median(r[fail][:, col])))
for thr in (120, 150, 180, 200, 250):
    flagged = (r[:, 3] > thr) & late
    print('steps >= 25, threshold on prev step norm %d: flags %5d of %d solves (%.2f %%), catches %d of %d failures' % (thr, flagged.sum(), late.sum(), 100 * flagged.sum() / late.sum(), (flagged & fail).sum(), (late & fail).sum()))
for thr in (14, 16, 18, 20, 22):
    flagged = (r[:, 5] >= thr) & late
    print('steps >= 25, threshold on prev iters %d: flags %5d of %d solves (%.2f %%), catches %d of %d failures' % (thr, flagged.sum(), late.sum(), 100 * flagged.sum() / late.sum(), (flagged & fail).sum(), (late & fail).sum()))
