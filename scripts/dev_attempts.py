"""Per RTI step of the bench protocol (256 Config-B instances, one step per launch): how many solves began with a lower-start attempt, how many
attempts were repeated from the standard start, factorisations per solve (mean / max over the instances) -- the data behind the attempt policy.
    python scripts/dev_attempts.py [tol_step start_mu [steps]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srbm_loader import host, workloads
ts = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cfg = host.load_config()
B = 256
st, ee = zip(*[workloads.config_b_instance(cfg, b) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200); g.set_solver_step_rule(ts, mu)
g.create_initial_run(st, ee)
prev_it = g.work_counters()[0]
tot = np.zeros(B)
print('mode', g.solver_step_rule())
even_mu = float(os.environ.get('EVEN_MU', mu))
for i in range(steps):
    g.set_solver_step_rule(ts, even_mu if i % 2 == 0 else mu)
    g.rti_advance(i, 1); g.synchronize()
    fl = g.solve_flags()
    it = g.stats()[:, 4]
    alpha = g.stats()[:, 0]
    snorm = g.stats()[:, 3]
    gapf = g.stats()[:, 7]; byst = gapf > 1e-15
    z_, s_ = g.dual_solution(); sz_ = g.sizes()
    nact = np.mean([(s_[b, 252:252 + sz_[b, 3]] < 1e-7).sum() for b in range(B)])
    tot += it
    tried, failed = int(((fl & 2) != 0).sum()), int(((fl & 4) != 0).sum())
    print('step %3d  attempts %3d  repeated %3d  iterations/solve mean %.2f max %d   of the repeated: mean %.1f   alpha<1: %d   nu %s  |p| mean %.2f  active rows mean %.1f  ended above gap 1e-15 (stall exit): %3d, their iterations %.1f vs %.1f' % (
        i, tried, failed, it.mean(), it.max(), it[(fl & 4) != 0].mean() if failed else 0.0, int((alpha < 1).sum()), np.unique(g.sizes()[:, 0]), snorm.mean(), nact, int(byst.sum()), it[byst].mean() if byst.any() else 0.0, it[~byst].mean() if (~byst).any() else 0.0))
print('per-instance total iterations over the run: mean %.1f  max %.1f  (the fused launch ends with the max)' % (tot.mean(), tot.max()))
