"""Free-running (own-path) behaviour of the solver mode bench.py times, against the oracle, with a STATED bound.

Both sides start from the same cold start and then run the open-loop protocol of test/gait_opt_playground.cpp:113-126 on their OWN
trajectories: each relinearises around its own previous solution, so what is compared is the SQP path, which amplifies per-solve
differences along the flat directions of the weakly convex QP and contains discrete decisions (Armijo step, foot-box size).  The
re-synchronised tests (tests/test_gpu_resync.py) are the statement about the solve; this one is the statement about the path:

    device: srbm_rti_advance(i, 1), tol_step 1e-5, start_mu 0.1 (step rule + lower-start attempts: the bench mode)
    oracle: Clarabel restatement at the reference's criterion (gap 1e-15)

for 128 seeded Config-B instances over 100 steps.  Asserted: the distribution of the relative difference (decision vector and node states)
over all (instance, step) pairs, that the paths agree to the parity tolerance again once the transient of the first horizon extension is over,
and that the fast mode departs from the oracle no more than a gap-criterion run of the device does.
The same protocol at the reference's criterion on the device (step rule off) is run beside it, so that the numbers say how much of the
divergence belongs to the termination rule and how much to two IPMs walking a flat valley."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle_py import OracleMPC, load_config
from srbm_loader import host
from srbm_loader.workloads import config_b_instance

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def own_path_run(fast, B=128, steps=100, mode=None):
    cfg = load_config()
    dt = cfg['integrator_dt']
    nx = (cfg['num_nodes'] + 1) * 12
    states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees)
    g = host.BatchMPC(cfg, B)
    g.set_state_trajectory_warm_start(states)
    g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
    if mode is not None:
        g.set_solver_step_rule(*mode)
    elif fast:
        g.enable_fast_termination()
    oracles = []
    for b in range(B):
        o = OracleMPC(cfg); o.set_warmstart(states[b]); oracles.append(o)
    pool = ThreadPoolExecutor(16)
    list(pool.map(lambda b: oracles[b].initial_run(states[b], ees[b].reshape(4, 3)), range(B)))
    g.create_initial_run(states, ees.reshape(B, 12))
    g.clear_status_accumulators()
    err = np.full((steps, B), np.nan)
    err_states = np.full((steps, B), np.nan)
    ok = np.ones(B, bool)           # both sides Solved / SolvedInacc so far

    def ostep(b, t):
        o = oracles[b]
        ee = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
        return o.rti(o.states()[1], t, ee)

    probe = (0.25, 0.5, 0.95)          # foot positions are compared through the SPLINES, at these offsets into the horizon (s)
    for i in range(steps):
        t = i * dt
        g.rti_advance(i, 1)
        so = list(pool.map(lambda b: ostep(b, t), range(B)))
        g.synchronize()
        st, e = g.status()
        assert np.all(e == 0), (i, np.nonzero(e)[0][:8])
        x = g.qp_solution(); tr = g.trajectory_states(); sz = g.sizes()
        t_first = t + dt                 # (the trajectory after the step starts at the solve's init_time; stay inside its knot range)
        pos_dev = [g.eval_trajectory(t_first + off)[1] for off in probe]
        assert np.all(g.status()[1] == 0)
        for b in range(B):
            ok[b] = ok[b] and st[b] <= 1 and so[b] <= 1
            o = oracles[b]
            osz = o.sizes()
            n = osz['n']
            if ok[b] and n == sz[b, 0]:       # (schedules are fixed in this protocol: sizes agree unless a side left the comparison)
                # states and force variables entry by entry.  The POSITION variables are compared through what they mean -- the foot
                # positions over the horizon -- because a knot the window is about to leave carries a variable whose coefficients in the QP are
                # ~1e-15 (a lift-off position one rounding error before the touch-down that ends its swing): every value of it is a minimiser, the
                # oracle's solver returns 0, the device keeps the previous value (seen at steps 66, 72, ... of this protocol)
                nxf = nx + osz['n_force']
                err[i, b] = relerr(x[b, :nxf], o.x()[:nxf])
                pe = 0.0
                for k, off in enumerate(probe):
                    po = np.array([[o.ee_value(e_, 1, c, t_first + off) for c in range(3)] for e_ in range(4)])
                    pe = max(pe, np.abs(pos_dev[k][b] - po).max())
                err_states[i, b] = max(relerr(tr[b], o.states()), pe)
    return err, err_states, ok, g.solver_counters()


def test_free_running_bench_mode_stays_within_a_stated_bound_of_the_oracle():
    err, err_s, ok, ctr = own_path_run(fast=True)
    err_r, err_rs, ok_r, _ = own_path_run(fast=False)
    def q(a):
        v = a[np.isfinite(a)]
        return dict(n=int(v.size), median=float(np.median(v)), p90=float(np.percentile(v, 90)), p99=float(np.percentile(v, 99)), max=float(v.max()))
    ef, er = np.fmax(err, err_s), np.fmax(err_r, err_rs)          # per (step, instance): decision vector and node states together
    d, dr = q(ef), q(er)
    settled, settled_r = q(ef[SETTLED_FROM:]), q(er[SETTLED_FROM:])
    over = lambda a: np.sum(np.where(np.isfinite(a), a, 0.0) > 1e-4, axis=1)          # per step: instances beyond the parity tolerance
    of, orr = over(ef), over(er)
    print('own path, 128 x 100, bench mode (tol_step 1e-5, start_mu 0.1) vs oracle:', d, 'from step %d on:' % SETTLED_FROM, settled, 'instances compared to the end %d' % ok.sum(), ctr)
    print('own path, 128 x 100, reference criterion on the device vs oracle:       ', dr, 'from step %d on:' % SETTLED_FROM, settled_r, 'instances compared to the end %d' % ok_r.sum())
    print('instances beyond 1e-4 per step, bench mode:         ', of.tolist())
    print('instances beyond 1e-4 per step, reference criterion:', orr.tolist())
    assert ok.sum() >= 0.95 * ok.size and d['n'] >= 0.95 * err.size
    assert ctr['low_tried'] >= 0.8 * ctr['solves'] and ctr['step_rule'] >= 0.8 * ctr['solves'], ctr
    # ---- the stated bounds of the fast mode on its own path (what was observed is in DESIGN.md section 4) ----
    frac_over = of.sum() / float(d['n'])
    print('pairs beyond 1e-4: bench mode %.1f %%, reference criterion %.1f %%' % (100 * frac_over, 100 * orr.sum() / float(dr['n'])))
    # (1) over all (instance, step) pairs: typical difference an order below the parity tolerance, nine in ten pairs within it, at most one in
    #     eight beyond it (observed: median 8e-6, 90th percentile 6e-5, 7 % beyond)
    assert d['median'] <= 2e-5 and d['p90'] <= 1e-4 and frac_over <= 0.125, (d, frac_over)
    # (2) the paths part TRANSIENTLY after the first extension of the horizon (steps 6-30: a new polynomial enters the window, each side's SQP
    #     finds the weakly determined new variables on its own path) and come together again; in the settled regime (from step SETTLED_FROM on) the
    #     fast mode stays within 5e-4 in 99 of 100 pairs (observed 1.1e-4; at the steps where a knot enters, every sixth, a third of the
    #     instances sits between 1e-4 and 9e-4 for one step) and nothing ever leaves the scale of the problem
    assert settled['p99'] <= 5e-4 and settled['max'] <= 5e-3, settled
    assert d['max'] <= 0.3, d
    # (3) the transient is a property of two SQP paths, not of the termination rule: two GAP-CRITERION solvers (device with the rule off, oracle)
    #     show the same tail, and the fast mode's is no worse than twice that
    assert d['p99'] <= 2.0 * dr['p99'] + 1e-3 and d['max'] <= 2.0 * dr['max'] + 1e-3, (d, dr)
    # ... whereas the gap-criterion run of the device itself is an order closer to the oracle in the settled regime: the price of the fast mode on
    # its own path, stated (the bench line carries both modes)
    assert settled_r['p99'] <= 1e-4 and dr['median'] <= 5e-6, (settled_r, dr)


SETTLED_FROM = 45
