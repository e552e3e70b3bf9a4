"""Data for an attempt policy keyed on how settled the SQP iteration is: per (instance, step) of the bench protocol (Config B, one step per launch) the
factorisations executed with lower-start attempts (0, 0.1) and without (0, 0), the attempt flags and the size of the PREVIOUS RTI step |p|; then, offline,
what a rule `attempt only when the previous |p| < tau` would cost per 20-step window (a launch ends with its slowest instance): the counterfactual cost of a
skipped attempt is the (0, 0) run's count of the same solve (the trajectories of the two runs agree to the solver's tolerance).
    python scripts/dev_attempt_policy.py [steps]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srbm_loader import host, workloads
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 105
cfg = host.load_config()
B = 256
st, ee = zip(*[workloads.config_b_instance(cfg, b) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
def run(mu):
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_step_rule(0.0, mu)
    for _ in range(10): g.create_initial_run(st, ee)
    def totals():
        out = np.zeros(B); g._chk(g.L.srbm_debug_get_instance_iters(g.h, out.ctypes.data_as(C.POINTER(C.c_double)))); return out
    fact = np.zeros((B, steps)); tried = np.zeros((B, steps), bool); failed = np.zeros((B, steps), bool); pn = np.zeros((B, steps))
    t0 = totals()
    for i in range(steps):
        g.rti_advance(i, 1); g.synchronize()
        t1 = totals(); fact[:, i] = t1 - t0; t0 = t1
        fl = g.solve_flags(); tried[:, i] = (fl & 2) != 0; failed[:, i] = (fl & 4) != 0
        pn[:, i] = g.stats()[:, 3]
    return fact, tried, failed, pn
fa, tr, fa_failed, pn = run(0.1)
fs, _, _, _ = run(0.0)
prev = np.concatenate([np.full((B, 1), np.inf), pn[:, :-1]], axis=1)
print('attempted solves %d, repeated %d; factorisations per solve: attempts that held %.1f, repeated %.1f, the same solves from the standard start %.1f / %.1f' % (
    tr.sum(), fa_failed.sum(), fa[tr & ~fa_failed].mean(), fa[fa_failed].mean(), fs[tr & ~fa_failed].mean(), fs[fa_failed].mean()))
print('per step: mean factorisations with attempts / without, max with / without, attempts, repeated')
for i in range(0, min(steps, 50)):
    print('  step %2d: mean %.1f / %.1f  max %3d / %3d  attempts %3d repeated %3d' % (i, fa[:, i].mean(), fs[:, i].mean(), fa[:, i].max(), fs[:, i].max(), tr[:, i].sum(), fa_failed[:, i].sum()))
edges = [0, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, np.inf]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = tr & (prev >= lo) & (prev < hi)
    if m.sum(): print('previous |p| in [%g, %g): %5d attempts, %4.1f %% repeated, factorisations with attempt %.1f, from the standard start %.1f' % (lo, hi, m.sum(), 100.0 * fa_failed[m].mean(), fa[m].mean(), fs[m].mean()))
def windows(cost):
    return [cost[:, 5 + 20 * w:25 + 20 * w].sum(axis=1).max() for w in range(5)]
print('windows (max over instances of the factorisations of steps 5+20w .. 24+20w): with attempts %s sum %d | without %s sum %d' % (windows(fa), sum(windows(fa)), windows(fs), sum(windows(fs))))
for tau in (0.02, 0.05, 0.1, 0.2, 0.5, 1.0, 2.0):
    cost = np.where(tr & (prev >= tau), fs, fa)
    print('rule: no attempt when the previous |p| >= %-5g -> windows %s sum %d' % (tau, windows(cost), sum(windows(cost))))
for first in (10, 15, 20, 25, 30):
    cost = fa.copy(); cost[:, :first] = np.where(tr[:, :first], fs[:, :first], fa[:, :first])
    print('rule: no attempt in the first %2d steps after the cold start -> windows %s sum %d' % (first, windows(cost), sum(windows(cost))))
