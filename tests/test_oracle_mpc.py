"""Pins the oracle's MPC restatement: structural invariants of the assembled QP (derived from
/root/reference/mpc/mpc.cpp:610-624,1101-1127,1205-1214 and mpc_single_rigid_body.cpp:323-341) and the reference's
finite-difference test of the contact-time partials (/root/reference/test/mpc_test.cpp:114-270).  CPU only."""
import math

import numpy as np
import pytest

from oracle_py import OracleMPC, load_config

# foot positions used by the reference's test: test/mpc_test.cpp:97-101
EE_TEST = np.array([[0.1526, 0.12523, 0.011089], [0.1526, -0.12523, 0.011089],
                    [-0.208321844, 0.1363286, 0.01444], [-0.208321844, -0.1363286, 0.01444]])


def make(cfg_name='a1_configuration', **over):
    cfg = load_config(cfg_name, **over)
    m = OracleMPC(cfg)
    s0 = np.array(cfg['srb_init'], float)
    m.set_warmstart(s0)          # warm start = initial state at all nodes (test/mpc_test.cpp:90)
    return cfg, m, s0


@pytest.mark.parametrize('N,dt,n,m,n_eq', [(10, 0.05, 252, 732, 140), (20, 0.05, 372, 1012, 260), (50, 0.02, 732, 1852, 620)])
def test_qp_sizes_at_t0(N, dt, n, m, n_eq):
    cfg, mpc, s0 = make(num_nodes=N, integrator_dt=dt)
    assert mpc.solve(s0, 0.0, EE_TEST) in (0, 1, 2)
    sz = mpc.sizes()
    assert (sz['n'], sz['m'], sz['n_eq'], sz['n_ineq']) == (n, m, n_eq, m - n_eq)
    assert (sz['n_force'], sz['n_pos'], sz['n_force_box'], sz['n_cone'], sz['n_td'], sz['n_start']) == (96, 24, 160, 320, 0, 8)
    assert sz['n_ee_loc'] == 2 * (N - 3) * 2 * 4


def test_assembled_rows_follow_reference_layout():
    cfg, mpc, s0 = make()
    mpc.solve(s0, 0.0, EE_TEST)
    A, b, P, q = mpc.qp_dense()
    N = 20
    # initial-condition rows: -I x0 = -x0_hat   (msrb.cpp:220-222)
    np.testing.assert_array_equal(A[:12, :12], -np.eye(12))
    np.testing.assert_allclose(b[:12], -np.array([0, 0, 0.3, 0, 0, 0, 0, 0, 0, 0, 0, 0.]))
    # dynamics rows: (I + dt A_k) x_k - x_{k+1} + dt B_k u = -dt C_k   (msrb.cpp:246-262)
    for k in range(N):
        blk = A[12 * (k + 1):12 * (k + 2)]
        np.testing.assert_array_equal(blk[:, 12 * (k + 1):12 * (k + 2)], -np.eye(12))
        Ak = blk[:, 12 * k:12 * (k + 1)]
        np.testing.assert_allclose(Ak[0:3, 3:6], np.eye(3) * 0.05 / cfg['mass'], rtol=1e-15)
        np.testing.assert_allclose(np.diag(Ak), 1.0)
    # cost: P = blkdiag(Q x N, Phi, force_cost I, 0) + 1e-3 I   (mpc.cpp:542-564,1094)
    Q = np.array(cfg['Q_srbd_diag'], float)
    expect = np.concatenate([np.tile(Q, N + 1), np.full(96, cfg['force_cost']), np.zeros(24)]) + 1e-3
    np.testing.assert_allclose(np.diag(P), expect, rtol=1e-15)
    assert np.count_nonzero(P - np.diag(np.diag(P))) == 0
    # force box: rows j<80 "f_z(t_s) <= force_bound", rows 80..159 "-f_z(t_s) <= 0"   (mpc.cpp:367-410, qp_data.cpp:256)
    fb = slice(252, 252 + 160)
    np.testing.assert_array_equal(b[fb], np.concatenate([np.full(80, cfg['force_bound']), np.zeros(80)]))
    np.testing.assert_array_equal(A[252:332], -A[332:412])
    # friction pyramid rhs zero (mpc.cpp:196), EE box rhs (msrb.cpp:403-406, qp_data.cpp:240)
    np.testing.assert_array_equal(b[412:732], 0)
    hip = np.array([[0.2055, 0.147], [0.2055, -0.147], [-0.1555, 0.147], [-0.1555, -0.147]])
    ub = np.tile((hip + 0.075).reshape(-1), 17)
    lb = np.tile((hip - 0.075).reshape(-1), 17)
    np.testing.assert_allclose(b[732:732 + 272], np.concatenate([ub, -lb]), rtol=1e-14)
    # EE start rows (msrb.cpp:451-471)
    np.testing.assert_allclose(b[1004:1012], EE_TEST[:, :2].reshape(-1))
    # linear cost (mpc.cpp:554-564): w = -Q x_des
    des = np.array([0, 0, 0.3] + [0] * 9, float)
    np.testing.assert_allclose(q[:252], np.tile(-Q * des, 21))
    np.testing.assert_array_equal(q[252:], 0)


def test_qp_solution_satisfies_kkt():
    cfg, mpc, s0 = make()
    mpc.initial_run(s0, EE_TEST)
    st = mpc.rti(s0, 0.0, EE_TEST)
    assert st == 0
    A, b, P, q = mpc.qp_dense()
    x, z, s = mpc.qp_x(), mpc.z(), mpc.s()
    sz = mpc.sizes()
    scale = max(1.0, np.abs(q).max())
    assert np.abs(P @ x + q + A.T @ z).max() / scale < 1e-8
    assert np.abs(A @ x + s - b).max() < 1e-8
    ineq = np.r_[252:252 + sz['n_ineq']]
    assert s[ineq].min() > -1e-9 and z[ineq].min() > -1e-9
    assert abs(s[ineq] @ z[ineq]) < 1e-6


def test_rti_loop_sizes_change_with_the_horizon():
    # open-loop protocol of /root/reference/test/gait_opt_playground.cpp:113-130 (state := node 1, t_i = i dt)
    cfg, mpc, s0 = make()
    mpc.initial_run(s0, EE_TEST)
    seen = set()
    state = s0
    for i in range(14):
        t = i * cfg['integrator_dt']
        ee = np.array([[mpc.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
        assert mpc.rti(state, t, ee) in (0, 1, 2)
        sz = mpc.sizes()
        seen.add((sz['n'], sz['m']))
        state = mpc.states()[1]
        assert mpc.stats()['eq_violation'] < 0.5
    assert (372, 1012) in seen and len(seen) >= 3          # sizes are ragged in time


def test_contact_time_partials_match_finite_differences():
    """test/mpc_test.cpp:114-270: FD of the assembled constraint matrix w.r.t. one contact time vs
    ComputeParamPartialsClarabel, for the dynamics, force-box and friction-cone blocks (abs 1e-4)."""
    TOL = 1e-4
    cfg, mpc, s0 = make()
    mpc.initial_run(s0, EE_TEST)
    traj_holder = mpc.clone()                    # `Trajectory traj = mpc.GetTrajectory();`  (:121)
    mpc2_base = mpc.clone()
    assert mpc.rti(s0, 0.0, EE_TEST) == 0         # (:122)
    A1, b1, _, _ = mpc.qp_dense()
    sz = mpc.sizes()
    ndyn, nfb, ncone = 252, sz['n_force_box'], sz['n_cone']
    ct = [mpc2_base.contact_times(e)[0] for e in range(4)]
    dt = math.sqrt(1e-16)
    for ee in range(4):
        for idx in range(1, len(ct[ee])):
            mod = [c.copy() for c in ct]
            mod[ee][idx] += dt
            mpc2 = mpc2_base.clone()              # mpc2.SetWarmStartTrajectory(traj)  (:130)
            mpc2.set_contact_times(mod)           # (:132)
            mpc2.rti(s0, 0.0, EE_TEST)
            A2, _, _, _ = mpc2.qp_dense()
            sz2 = mpc2.sizes()
            assert sz2['n_force_box'] == nfb and sz2['n_cone'] == ncone
            dA, dG, db, dh = mpc.param_partials(ee, idx, traj_src=traj_holder)
            fd_dyn = (A2[:ndyn] - A1[:ndyn]) / dt
            assert np.abs(dA[:ndyn] - fd_dyn).max() < TOL, (ee, idx, np.abs(dA[:ndyn] - fd_dyn).max())
            fd_fb = (A2[ndyn:ndyn + nfb] - A1[ndyn:ndyn + nfb]) / dt
            assert np.abs(dG[:nfb] - fd_fb).max() < TOL, (ee, idx)
            fd_cone = (A2[ndyn + nfb:ndyn + nfb + ncone] - A1[ndyn + nfb:ndyn + nfb + ncone]) / dt
            assert np.abs(dG[nfb:nfb + ncone] - fd_cone).max() < TOL, (ee, idx)


def test_config_b_fixture_is_reproducible_from_the_oracle():
    """tests/golden/config_b_rti.json holds ORACLE outputs (not reference outputs: the reference cannot be built here,
    SURVEY.md section 8c).  Re-derive a sample of it, including one primal-infeasible cold start."""
    import json, os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle', 'tools'))
    import make_config_b_golden as mk
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'config_b_rti.json')))
    X = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'config_b_rti_x.npz'))['x']
    assert X.shape == (256, gold['steps'], 412)
    statuses = [r['steps'][0]['status'] for r in gold['instances']]
    assert len(statuses) == 256 and statuses.count(3) >= 4
    for b in (0, 11, statuses.index(3), 255):
        got = mk.run(b)
        ref = gold['instances'][b]
        for a, r in zip(got['steps'], ref['steps']):
            assert (a['status'], a['iters'], a['n'], a['m'], a['alpha']) == (r['status'], r['iters'], r['n'], r['m'], r['alpha'])
            assert abs(a['x_sum'] - r['x_sum']) <= 1e-9 * max(1.0, abs(r['x_sum']))
        assert np.array_equal(np.array(got['_x']), X[b])           # the full minimisers, bit for bit


def test_plant_integral_restates_the_euler_integrator():
    """RKIntegrator::CalcIntegral (rk_integrator.cpp:14-30) over CalcDynamics (single_rigid_body_model.cpp:222-256): n
    sub-steps at the fixed time are n single steps; the position rate is momentum / mass; the linear-momentum rate is
    m g + sum of the foot forces of the trajectory (no stored vector exists in the reference for this function)."""
    cfg = load_config(num_nodes=10)
    s0 = np.array(cfg['srb_init'], float)
    ee = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)
    o = OracleMPC(cfg); o.set_warmstart(s0); o.initial_run(s0, ee)
    x = s0.copy(); x[3:6] = [1.0, -2.0, 0.5]
    t, h = 0.02, 0.004
    a = o.plant_integrate(x, t, h, 3, 0)
    b = x.copy()
    for _ in range(3): b = o.plant_integrate(b, t, h, 1, 0)
    assert np.abs(a - b).max() < 1e-14
    one = o.plant_integrate(x, t, h, 1, 0)
    mass = cfg['mass']
    assert np.allclose(one[0:3], x[0:3] + h * x[3:6] / mass, atol=1e-15)
    F = np.array([[o.ee_value(e, 0, c, t) for c in range(3)] for e in range(4)]).sum(axis=0)
    assert np.allclose(one[3:6], x[3:6] + h * (np.array([0, 0, -9.81 * mass]) + F), atol=1e-12)
    # time moving with the sub-steps differs from the frozen-time integral as soon as the forces vary
    c = o.plant_integrate(x, t, h, 3, 1)
    assert np.abs(c - a).max() > 0
