"""The LARGE-capacity build (libsrbm_rti_large.so: the same sources with N <= 100, n_u <= 240, 20 stance phases; the packed normal
matrix in the work record instead of LDS) against the oracle: BASELINE config 5 as SURVEY.md section 8d re-defines it (the
reference's centroidal MPC is dead code) -- the SRBM path at N = 40, dt = 0.05 (2 s horizon), a share of 128 instances -- and the
short-phase schedules that the standard build refuses with SRBM_ERR_CAPACITY.  'SRBM stand-in, no reference parity beyond the SRBM
restatement' (SURVEY.md Config E)."""
import numpy as np
import pytest

from oracle_py import OracleMPC, load_config
from srbm_loader import host
from srbm_loader.workloads import config_b_instance

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4
EE0 = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def test_capacities():
    assert host.lib(False).capacity == dict(N=50, nu=160, samples=120, knots=32)
    assert host.lib(True).capacity == dict(N=100, nu=240, samples=200, knots=32)


def test_large_build_equals_standard_build_on_config_b():
    """same sources, same algorithm: on a problem both builds hold, the results agree to rounding (the summation orders of the dense
    blocks differ with the tile distribution, nothing else)"""
    cfg = load_config()
    B = 8
    states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    res = []
    for large in (False, True):
        g = host.BatchMPC(cfg, B, large=large)
        g.set_state_trajectory_warm_start(states)
        g.set_solver_step_rule(0.0, 0.0)          # fourteen consecutive solves on each build's own path: compared at the gap criterion (rounding-level
                                                  # differences between the builds must not decide in which iteration a tolerance-based rule fires)
        g.create_initial_run(states, ees)
        g.rti_advance(0, 4); g.synchronize()
        st, err = g.status()
        assert np.all(err == 0) and np.all(st <= 1)
        n = int(g.sizes()[0, 0])
        res.append((g.qp_solution()[:, :n], g.sizes(), g.get_trajectory()))
    assert np.array_equal(res[0][1], res[1][1])
    assert relerr(res[1][0], res[0][0]) < REL_TOL
    for b in range(B):
        assert np.array_equal(np.array(res[0][2][b].knot_time), np.array(res[1][2][b].knot_time))


def test_n40_horizon_share_of_128_against_the_oracle():
    cfg = load_config(num_nodes=40)                      # dt = 0.05: a 2 s horizon, three more phases per foot
    B = 128
    states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    g = host.BatchMPC(cfg, B, large=True)
    g.set_state_trajectory_warm_start(states)
    g.create_initial_run(states, ees)
    sample = [0, 17, 64, 127]
    oracles = []
    for b in sample:
        o = OracleMPC(cfg); o.set_warmstart(states[b]); o.initial_run(states[b], ees[b].reshape(4, 3)); oracles.append(o)
    dt = cfg['integrator_dt']
    seen = set()
    for i in range(4):
        t = i * dt
        # the sampled instances start every step from the oracle's trajectory (tests/test_gpu_resync.py); the rest run open loop
        own = g.get_trajectory()
        tr, (_, own_ee, _) = g.trajectory_states(), g.eval_trajectory(t)
        st_in = np.array([states[b] if i == 0 else tr[b, 1] for b in range(B)])
        ee_in = ees.copy() if i == 0 else own_ee.reshape(B, 12).copy()
        for k, b in enumerate(sample):
            o = oracles[k]
            own[b] = o.trajectory_record(host)
            st_in[b] = o.states()[1] if i > 0 else states[b]
            ee_in[b] = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)]).reshape(-1)
        g.set_warm_start_trajectory(own)
        g.get_real_time_update(st_in, t, ee_in)
        st, err = g.status(); sz = g.sizes(); x = g.qp_solution()
        assert np.all(err == 0), (i, err[err != 0])
        assert np.all(st <= 2), (i, np.unique(st, return_counts=True))
        seen.add(int(sz[0, 0]))
        for k, b in enumerate(sample):
            o = oracles[k]
            so = o.rti(st_in[b], t, ee_in[b].reshape(4, 3))
            osz = o.sizes()
            assert (sz[b, 0], sz[b, 1], sz[b, 6]) == (osz['n'], osz['m'], osz['n_td']), (i, b)
            if so <= 1 and st[b] <= 1:
                n = osz['n']
                assert relerr(x[b, :n], o.x()) < REL_TOL, (i, b, relerr(x[b, :n], o.x()))
                if i == 0:
                    A, bv, P, q = g.export_qp(b); Ao, bo, Po, qo = o.qp_dense()
                    assert max(np.abs(A - Ao).max(), np.abs(bv - bo).max(), np.abs(q - qo).max()) < 1e-12, (i, b)
    nu = max(seen) - 41 * 12
    assert nu > 160, nu                                    # beyond the standard build's capacity
    acc = g.status_accumulated()
    print('N = 40 share: n_u up to %d, statuses of the last step %s, not-solved solves %d of %d' % (nu, dict(zip(*np.unique(st, return_counts=True))), acc[:, 2].sum(), acc[:, 1].sum()))


def test_short_phase_schedule_runs_in_the_large_build():
    """the schedule of tests/test_gpu_parity.py::test_capacity_overflow_fails_loudly is legal for the reference (MIN_TIME 0.2 is
    enforced by the gait LP, not by the MPC); it needs more than 160 variables"""
    cfg = load_config()
    s0 = np.array(cfg['srb_init'], float)
    g = host.BatchMPC(cfg, 2, large=True)
    g.set_state_trajectory_warm_start(s0)
    o = OracleMPC(cfg); o.set_warmstart(s0)
    g.create_initial_run(s0, EE0); o.initial_run(s0, EE0)
    ct = [o.contact_times(e)[0] for e in range(4)]
    new = [0.1 * np.arange(len(c)) for c in ct]            # 100 ms phases
    o.set_contact_times(new)
    arr = np.zeros((2, 4, 8))
    for e in range(4):
        arr[:, e, :len(new[e])] = new[e]
    g.update_contact_times(arr)
    g.set_warm_start_trajectory([o.trajectory_record(host)] * 2)
    g.get_real_time_update(s0, 0.0, EE0)
    so = o.rti(s0, 0.0, EE0)
    st, err = g.status()
    sz, osz = g.sizes()[0], o.sizes()
    assert err[0] == 0 and (sz[0], sz[1]) == (osz['n'], osz['m'])
    assert sz[0] - 21 * 12 > 160
    if so <= 1 and st[0] <= 1:
        assert relerr(g.qp_solution()[0, :osz['n']], o.x()) < REL_TOL
