"""Certified minimiser of a strictly convex QP in the reference's form (test infrastructure, numpy only).

    min 1/2 x'Px + q'x   s.t.  A x + s = b,  s = 0 on the zero-cone rows, s >= 0 on the non-negative rows

Two interior-point codes stopped at a 1e-13 gap agree on this QP only to about 1e-4 along its flat directions (curvature
1e-3 against Q ~ 4000), which is exactly the tolerance of the parity tests.  To say WHICH side an outlier belongs to, the
tests polish a solution: guess the active set from it, solve the equality-constrained QP by a dense KKT solve with iterative
refinement, and keep the result only if it carries a full KKT certificate (primal feasibility of the inactive rows, sign of
the multipliers of the active ones).  P is positive definite (1e-3 I is added to every diagonal entry, mpc.cpp:1094), so a
certified point is THE minimiser, whatever produced the active-set guess.
"""
import numpy as np


def _dual_certificate(P, q, A, is_eq, x, tight, scale):
    """is there nu (free on zero-cone rows, >= 0 on tight inequality rows) with P x + q + A_tight' nu = 0 ?  Non-negative least
    squares, so that dependent active rows (a foot with zero force: its lower box row and four pyramid rows are tight together)
    are handled: ANY valid split of the multipliers certifies."""
    from scipy.optimize import nnls
    g = P @ x + q
    idx = np.nonzero(tight)[0]
    eq = is_eq[idx]
    At = A[idx].T                                         # n x m_t
    M = np.concatenate([At, -At[:, eq]], axis=1)          # free multipliers as a difference of two non-negative ones
    nu, rn = nnls(M, -g, maxiter=20 * M.shape[1])
    return rn / max(1.0, np.abs(g).max()), idx, nu


def polish(P, q, A, b, is_eq, x0, z0, s0, rounds=5):
    """returns (x, info) -- x is None if no certificate was obtained.  is_eq: bool mask of the zero-cone rows."""
    n = len(q)
    ineq = ~is_eq
    nz = np.abs(A).sum(axis=1) > 0
    scale = max(1.0, np.abs(z0).max())
    active = is_eq | (ineq & nz & (z0 > s0))
    for rnd in range(rounds):
        idx = np.nonzero(active)[0]
        Aa = A[idx]
        ma = len(idx)
        # dependent active rows make the KKT matrix singular (x stays unique, P > 0): least-squares solve + one refinement
        K = np.zeros((n + ma, n + ma))
        K[:n, :n] = P
        K[:n, n:] = Aa.T
        K[n:, :n] = Aa
        rhs = np.concatenate([-q, b[idx]])
        sol = np.linalg.lstsq(K, rhs, rcond=1e-13)[0]     # minimum-norm solution of the (consistent, possibly singular) system
        sol = sol + np.linalg.lstsq(K, rhs - K @ sol, rcond=1e-13)[0]
        x, nu = sol[:n], sol[n:]
        slack = b - A @ x
        viol = ineq & nz & ~active & (slack < -1e-9)
        if viol.any():
            active = active | viol
            continue
        if np.abs(slack[is_eq]).max() > 1e-8:
            return None, dict(reason='equality rows open: %g' % np.abs(slack[is_eq]).max())
        tight = is_eq | (ineq & nz & (slack < 1e-9))
        rn, tidx, tnu = _dual_certificate(P, q, A, is_eq, x, tight, scale)
        if rn < 1e-8:
            return x, dict(rounds=rnd + 1, n_active=int(ma), stationarity=float(rn), min_slack=float(slack[ineq & nz].min()))
        # over-constrained guess: release the rows whose multiplier in the KKT solve is negative
        wrong = np.zeros(len(b), bool)
        wrong[idx] = ineq[idx] & (nu < 0)
        if not wrong.any():
            return None, dict(reason='stationarity %g without a negative multiplier' % rn)
        active = active & ~wrong
    return None, dict(reason='no certificate after %d rounds' % rounds)
