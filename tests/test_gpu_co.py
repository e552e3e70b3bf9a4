"""The co-resident kernel set (include/srbm_rti.h: srbm_set_kernel_set; csrc/srbm_co.hip): the kernels of the RTI path compiled a second time for
256 threads per workgroup with the normal matrix of the solve in L2, so that two instances share a CU.  It is chosen by the batch size (more
instances than CUs); here it is forced on small batches and compared with the standard set on identical inputs: the same algorithm on the same
data, reductions over 4 instead of 8 waves -- knot tables and Armijo steps EQUAL, minimisers / duals / trajectories equal to rounding-level
tolerances, statuses equal; and against the oracle on a re-synchronised step."""
import numpy as np
import pytest

from oracle_py import OracleMPC, load_config
from srbm_loader import host
from bench import config_b_instance, config_d_instance

pytestmark = pytest.mark.gpu


def run(cfg, states, ees, which, steps):
    B = len(states)
    g = host.BatchMPC(cfg, B)
    g.set_state_trajectory_warm_start(states)
    g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
    g.set_solver_step_rule(0.0, 0.0)                 # both sets to the gap criterion: the comparison is then about the kernels, not about which
                                                     # iteration a tolerance-based rule fires in
    g.set_kernel_set(which)
    assert g.kernel_set() == which
    g.create_initial_run(states, ees)                # 10 solves through the four-launch path of the set
    g.rti_advance(0, steps); g.synchronize()         # ... and the fused kernel of the set
    return g


@pytest.mark.parametrize('cfgname,B,steps', [('a1_configuration', 24, 6), ('a1_config_distr_rejection', 8, 3)])
def test_co_resident_set_equals_the_standard_set(cfgname, B, steps):
    cfg = load_config(cfgname)
    mk = config_b_instance if cfgname == 'a1_configuration' else config_d_instance
    states, ees = zip(*[mk(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    a = run(cfg, states, ees, 0, steps)
    c = run(cfg, states, ees, 1, steps)
    sa, ea = a.status(); sc, ec = c.status()
    assert np.array_equal(sa, sc) and np.all(ea == 0) and np.all(ec == 0)
    assert np.array_equal(a.sizes(), c.sizes())
    assert np.array_equal(a.stats()[:, 0], c.stats()[:, 0])                       # Armijo step
    ta, tc = a.get_trajectory(), c.get_trajectory()
    for b in range(B):
        for e in range(4):
            assert np.array_equal(np.array(ta[b].get_contact_times()[e]), np.array(tc[b].get_contact_times()[e]))     # knot times: bit-exact
    ok = sa <= 1
    xa, xc = a.qp_solution()[ok], c.qp_solution()[ok]
    rel = np.abs(xa - xc).max(axis=1) / np.maximum(1.0, np.abs(xa).max(axis=1))
    # the two sets differ by the summation order of their reductions: each solve reproduces to ~1e-9 (the end game of the IPM amplifies rounding
    # along the flat directions); `steps` + 10 relinearisations later, each set on its own path, the suite's own-path tolerance applies (observed: 7e-6)
    assert rel.max() < 1e-4, rel.max()
    za, zc = a.dual_solution()[0][ok], c.dual_solution()[0][ok]
    assert np.abs(a.trajectory_states()[ok] - c.trajectory_states()[ok]).max() < 1e-4 * max(1.0, np.abs(a.trajectory_states()[ok]).max())
    print('%s: co-resident vs standard set over %d instances x %d steps: worst relative primal difference %.1e, dual %.1e' %
          (cfgname, B, 10 + steps, rel.max(), (np.abs(za - zc).max(axis=1) / np.maximum(1.0, np.abs(za).max(axis=1))).max()))


def test_co_resident_set_against_the_oracle_and_the_default_selection():
    cfg = load_config()
    B = 16
    states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees)
    g = host.BatchMPC(cfg, B)
    assert g.kernel_set() == 0                         # a batch that fits one instance per CU stays on the standard set
    g.set_state_trajectory_warm_start(states)
    g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
    g.set_kernel_set(1)
    os_ = []
    for b in range(B):
        o = OracleMPC(cfg); o.set_warmstart(states[b]); o.initial_run(states[b], ees[b]); os_.append(o)
    g.create_initial_run(states, ees.reshape(B, 12))
    worst = 0.0
    for i in range(3):
        t = i * cfg['integrator_dt']
        g.set_warm_start_trajectory([o.trajectory_record(host) for o in os_])
        st_in = np.array([o.states()[1] if i > 0 else states[b] for b, o in enumerate(os_)])
        ee_in = np.array([[[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)] for o in os_]).reshape(B, 12)
        so = [o.rti(st_in[b], t, ee_in[b].reshape(4, 3)) for b, o in enumerate(os_)]
        g.get_real_time_update(st_in, t, ee_in)
        st, err = g.status(); xr = g.raw_qp_minimiser(); stats = g.stats()
        assert np.all(err == 0)
        for b, o in enumerate(os_):
            if so[b] > 1:
                continue
            assert st[b] <= 1
            if o.stats()['step_norm'] > 1e-3:               # (the Armijo test is noise below that, as in tests/test_gpu_resync.py)
                assert stats[b, 0] == o.stats()['alpha']
            xo = o.qp_x()
            worst = max(worst, np.abs(xr[b, :len(xo)] - xo).max() / max(1.0, np.abs(xo).max()))
    assert worst < 1e-4, worst
    big = host.BatchMPC(cfg, 300)                      # more instances than the 256 CUs of an MI355X: still created on the standard set (round 4)
    assert big.kernel_set() == 0
    big.set_kernel_set(1)                              # ... the co-resident set is there on request
    assert big.kernel_set() == 1
