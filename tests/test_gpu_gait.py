"""GPU parity tests of the bilevel (gait) step against the oracle's GaitOptimizer restatement (oracle/srbm_gait.hpp),
through the C-ABI (include/srbm_rti.h, srbm_gait_*).  Tolerances as tests/test_gpu_parity.py: contact times bit-exact,
argmin index exact, costs / trajectories <= 1e-4 relative."""
import numpy as np
import pytest

from oracle_py import OracleMPC, load_config
from srbm_loader import host

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4
EE0 = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def run_pair(cfgname, nsteps, batch=2):
    """GPU batch + oracle brought to the same point of an open-loop RTI run (test/gait_opt_playground.cpp:113-126)"""
    cfg = load_config(cfgname)
    s0 = np.array(cfg['srb_init'], float)
    g = host.BatchMPC(cfg, batch)
    g.set_state_trajectory_warm_start(s0)
    g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
    o = OracleMPC(cfg)
    o.set_warmstart(s0)
    g.create_initial_run(s0, EE0); o.initial_run(s0, EE0)
    dt = cfg['integrator_dt']
    state, ee, t = s0, EE0, 0.0
    for i in range(nsteps):
        t = i * dt
        state = o.states()[1]
        ee = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
        o.rti(state, t, ee)
        g.get_real_time_update(state, t, ee)
    return cfg, g, o, state, ee, t


@pytest.mark.parametrize('cfgname,nsteps', [('a1_configuration', 3), ('a1_gait_opt_config', 2)])
def test_candidate_line_search_matches_oracle(cfgname, nsteps):
    cfg, g, o, state, ee, t = run_pair(cfgname, nsteps)
    assert o.stats()['status'] == 0
    grad = o.gait_gradient()
    assert grad is not None
    step, new_times = o.gait_optimize(t)
    nv = len(grad)
    gait = host.BatchGaitOptimizer(g)
    gait.set_contact_times_from_trajectory()
    xk, counts = gait.contact_times()
    assert counts[0].sum() == nv and np.array_equal(counts[0], counts[1])
    xo = np.concatenate([o.contact_times(e)[0] for e in range(4)])
    assert np.array_equal(xk[0, :nv], xo)                   # contact schedule: bit-exact
    gait.set_step(step[:nv])
    k_o, costs_o = o.gait_line_search(state, t, ee)
    imin, costs = gait.line_search(state, t, ee)
    st, err = gait.candidate_status()
    assert np.all(err == 0)
    assert np.array_equal(costs[0], costs[1])               # identical instances -> identical candidates
    assert imin[0] == k_o and imin[1] == k_o
    assert np.abs(costs[0] - costs_o).max() <= REL_TOL * max(1.0, np.abs(costs_o).max())
    # the winner's trajectory was installed: same knot tables and node states as the oracle's
    kg = g.knots(0)
    for e in range(4):
        ko = o.knots(e)
        assert kg['nk'][e] == ko['K'] and np.array_equal(kg['times'][e, :ko['K']], ko['times'])
    tr = g.trajectory_states()[0]
    assert relerr(tr, o.states()) < REL_TOL
    # and the next RTI step from the installed trajectory agrees too
    t2 = t + cfg['integrator_dt']
    state2 = o.states()[1]
    ee2 = np.array([[o.ee_value(e, 1, c, t2) for c in range(3)] for e in range(4)])
    o.rti(state2, t2, ee2)
    g.get_real_time_update(state2, t2, ee2)
    n = o.sizes()['n']
    assert (g.sizes()[0, 0], g.sizes()[0, 1]) == (n, o.sizes()['m'])
    assert relerr(g.qp_solution()[0, :n], o.x()) < REL_TOL
