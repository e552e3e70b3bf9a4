"""Pins the oracle's QP solver restatement (oracle/clarabel_like.hpp) and the KKT sensitivity (oracle/srbm_gait.hpp)
on the reference's own 3-variable fixture: /root/reference/test/mpc_test.cpp:857-904 (data), :916-1003 (assertions).
The reference cross-checks Clarabel against OSQP (neither is available here); the numeric answer below was derived
independently by eliminating the two equalities and is re-derived in the test with numpy."""
import numpy as np

from oracle_py import qp_solve, qp_sensitivity

MARGIN = 1e-4   # mpc_test.cpp:918


def fixture():
    P = np.diag([3.001, 4, 0.5]); q = np.array([0.1, 4.6, 2])            # :868-875
    A = np.array([[1, 1, 0], [1.3, 0, 0.2]]); b = np.array([1., 3])       # :877-884
    G = np.array([[-2, 0, 0.9], [1, 8, 5]])                                # :886
    ub = np.array([3.1, 13.3]); lb = np.array([-2., -5])                   # :887-890
    # Clarabel form (qp_data.cpp:253-258): rows [A; G; -G], rhs [b; ub; -lb]; cones Zero(2) x Nonneg(4)
    return P, q, np.vstack([A, G, -G]), np.concatenate([b, ub, -lb])


def test_fixture_primal_known_answer():
    P, q, Aall, ball = fixture()
    r = qp_solve(P, q, Aall, ball, [(0, 2), (1, 4)])
    assert r['status'] == 0   # Solved
    np.testing.assert_allclose(r['x'], [1.974522293, -0.974522293, 2.165605096], atol=MARGIN)
    # independent re-derivation: active set = {both equalities, lower side of box row 0}
    K = np.zeros((6, 6)); K[:3, :3] = P
    act = np.vstack([Aall[0], Aall[1], Aall[4]])
    K[:3, 3:] = act.T; K[3:, :3] = act
    sol = np.linalg.solve(K, np.concatenate([-q, [ball[0], ball[1], ball[4]]]))
    np.testing.assert_allclose(r['x'], sol[:3], atol=1e-7)
    np.testing.assert_allclose([r['z'][0], r['z'][1], r['z'][4]], sol[3:], atol=1e-6)
    obj = 0.5 * r['x'] @ P @ r['x'] + q @ r['x']
    assert abs(obj - 8.96776543) < 1e-6
    # KKT conditions of the Clarabel form
    np.testing.assert_allclose(P @ r['x'] + q + Aall.T @ r['z'], 0, atol=1e-7)
    np.testing.assert_allclose(Aall @ r['x'] + r['s'], ball, atol=1e-7)
    assert np.all(r['s'][2:] >= -1e-9) and np.all(r['z'][2:] >= -1e-9)
    assert abs(r['s'][2:] @ r['z'][2:]) < 1e-7


def test_fixture_sensitivity_matches_finite_differences():
    """dA/dG of clarabel_interface.cpp:198-260 are the gradients of the outer loss l(x*) = 1/2 x*'Px* + q'x*
    (dl/dx = Px*+q, :604-612).  The reference checks them against OSQP's adjoint derivatives on the sparsity pattern
    (mpc_test.cpp:984-1003); here they are checked against central finite differences of re-solved QPs."""
    P, q, Aall, ball = fixture()
    r = qp_solve(P, q, Aall, ball, [(0, 2), (1, 4)], tol_gap=1e-13, tol_feas=1e-13)
    dA, dG, dq, db, dh = qp_sensitivity(P, Aall, q, r['x'], r['z'], r['s'], 2, 4)

    def loss(Am, bm):
        rr = qp_solve(P, q, Am, bm, [(0, 2), (1, 4)], tol_gap=1e-13, tol_feas=1e-13)
        return 0.5 * rr['x'] @ P @ rr['x'] + q @ rr['x']

    eps = 1e-6
    full = np.vstack([dA, dG])
    for i in range(6):
        for j in range(3):
            if Aall[i, j] == 0:
                continue          # the reference only asserts on the non-zero pattern (:986, :996)
            Ap = Aall.copy(); Ap[i, j] += eps
            Am = Aall.copy(); Am[i, j] -= eps
            fd = (loss(Ap, ball) - loss(Am, ball)) / (2 * eps)
            assert abs(full[i, j] - fd) < MARGIN, (i, j, full[i, j], fd)
    dvec = np.concatenate([db, dh])
    for i in range(6):
        bp = ball.copy(); bp[i] += eps
        bm = ball.copy(); bm[i] -= eps
        fd = (loss(Aall, bp) - loss(Aall, bm)) / (2 * eps)
        assert abs(dvec[i] - fd) < MARGIN, (i, dvec[i], fd)


def test_infeasible_and_lp_statuses():
    # primal infeasible: x <= -1 and x >= 1
    r = qp_solve(np.array([[1.0]]), np.array([0.0]), np.array([[1.0], [-1.0]]), np.array([-1.0, -1.0]), [(1, 2)])
    assert r['status'] in (3, 5)     # PrimalInfeasible / PrimalInfeasibleInacc (qp_interface.h:12-22)
    # LP with a unique vertex: min x0 + x1  s.t. x >= 1
    r = qp_solve(np.zeros((2, 2)), np.array([1.0, 1.0]), -np.eye(2), -np.ones(2), [(1, 2)])
    assert r['status'] == 0
    np.testing.assert_allclose(r['x'], [1, 1], atol=1e-6)
    # dual infeasible (unbounded): min -x s.t. x >= 0
    r = qp_solve(np.zeros((1, 1)), np.array([-1.0]), -np.eye(1), np.zeros(1), [(1, 1)])
    assert r['status'] in (4, 6)
