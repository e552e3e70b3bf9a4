"""Pins of the numpy restatement of the whole-body QP controller (oracle/wbc_numpy.py).  No fixture exists in the
reference for this step and its rigid-body arithmetic lives in pinocchio (absent, unpinned): the restatement is pinned by what the
quantities ARE -- M is the Hessian of the kinetic energy, g the gradient of the potential energy, the support Jacobian the derivative
of the foot positions, 'Jdot v' the second derivative of the foot position along a motion of constant generalized velocity -- each
against finite differences of an independent forward-kinematics computation, and by the KKT conditions of the QP."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ik_numpy as ik
import wbc_numpy as wbc
from oracle_py import qp_solve

CFG = json.load(open(os.path.join(ROOT, 'bilevel-gait-gen_amd', 'configs', 'a1_configuration.json')))
Q0 = np.array(CFG['init_config'], float)


def random_state(seed):
    rng = np.random.default_rng(seed)
    q = Q0.copy()
    q[:3] += rng.normal(size=3) * 0.05
    q[3:7] = q[3:7] + np.concatenate([rng.normal(size=3) * 0.1, [0]]); q[3:7] /= np.linalg.norm(q[3:7])
    q[7:] += rng.normal(size=12) * 0.2
    v = rng.normal(size=18) * 0.5
    return q, v


def body_points(robot, q):
    """world position of every body's centre of mass and its world rotation, by plain forward kinematics (independent of the RNEA)"""
    Rb = ik.quat_to_R(q[3:7])
    out = [(q[:3] + Rb @ np.array(CFG['body_model'][0]['com']), Rb)]
    for ee in range(4):
        R, p = Rb.copy(), q[:3].copy()
        for k in range(3):
            p = p + R @ robot.legs[ee][k]
            R = R @ ik.rot(0 if k == 0 else 1, q[7 + 3 * ee + k])
            out.append((p + R @ np.array(CFG['body_model'][1 + 3 * ee + k]['com']), R))
    return out


def energies(robot, q, v, h=1e-6):
    pts0, pts1 = body_points(robot, ik.integrate(q, v, -h)), body_points(robot, ik.integrate(q, v, h))
    T = 0.0
    for b, ((c0, R0), (c1, R1)) in enumerate(zip(pts0, pts1)):
        m, I = CFG['body_model'][b]['mass'], np.array(CFG['body_model'][b]['inertia'])
        vc = (c1 - c0) / (2 * h)
        w_world, _ = ik.log3(R1 @ R0.T)
        w_body = (0.5 * (R0 + R1)).T @ (w_world / (2 * h))
        T += 0.5 * m * vc @ vc + 0.5 * w_body @ I @ w_body
    return T


def potential(robot, q):
    return sum(CFG['body_model'][b]['mass'] * wbc.GRAV * c[2] for b, (c, _) in enumerate(body_points(robot, q)))


def test_mass_matrix_is_the_kinetic_energy_hessian_and_gravity_the_potential_gradient():
    robot = wbc.Robot(CFG)
    assert abs(robot.mass - 13.741) < 1e-9
    for seed in range(3):
        q, v = random_state(seed)
        M, Cv, g = robot.dynamics_terms(q, v)
        assert np.abs(M - M.T).max() < 1e-12 and np.linalg.eigvalsh(M).min() > 0
        assert np.abs(M[:3, :3] - robot.mass * np.eye(3)).max() < 1e-12
        assert abs(0.5 * v @ M @ v - energies(robot, q, v)) < 1e-6 * (0.5 * v @ M @ v)
        gfd = np.zeros(18)
        for j in range(18):
            e = np.zeros(18); e[j] = 1
            gfd[j] = (potential(robot, ik.integrate(q, e, 1e-6)) - potential(robot, ik.integrate(q, e, -1e-6))) / 2e-6
        assert np.abs(g - gfd).max() < 1e-6
        # power balance of the Coriolis terms: along a motion with zero generalized acceleration tau = C v + g, and the kinetic energy
        # changes by the power of the non-gravity forces: d/dt (1/2 v'M v) = v'(tau - g) = v'(C v)   (Mdot - 2C is skew)
        h = 1e-5
        Mp = robot.dynamics_terms(ik.integrate(q, v, h), v)[0]; Mm = robot.dynamics_terms(ik.integrate(q, v, -h), v)[0]
        dT = 0.5 * v @ (Mp - Mm) @ v / (2 * h)
        assert abs(dT - v @ Cv) < 1e-5 * max(1.0, abs(dT))


def test_support_jacobian_and_foot_acceleration_by_finite_differences():
    robot = wbc.Robot(CFG)
    q, v = random_state(7)
    for ee in range(4):
        J = robot.foot_jacobian_lwa(q, ee)
        Jfd = np.zeros((3, 18))
        for j in range(18):
            e = np.zeros(18); e[j] = 1
            Jfd[:, j] = (ik.forward_kinematics(robot.legs, ik.integrate(q, e, 1e-6))[ee] - ik.forward_kinematics(robot.legs, ik.integrate(q, e, -1e-6))[ee]) / 2e-6
        assert np.abs(J - Jfd).max() < 1e-8
        h = 1e-4
        p = [ik.forward_kinematics(robot.legs, ik.integrate(q, v, s * h))[ee] for s in (-1, 0, 1)]
        acc_fd = (p[2] - 2 * p[1] + p[0]) / h ** 2
        assert np.abs(robot.foot_classical_acceleration(q, v, ee) - acc_fd).max() < 1e-5


def test_whole_body_qp_solution_satisfies_its_kkt_conditions():
    robot = wbc.Robot(CFG)
    q, v = random_state(11)
    v *= 0.2
    q_des, v_des = Q0.copy(), np.zeros(18)
    for contact in ([1, 1, 1, 1], [1, 0, 0, 1], [0, 1, 1, 0]):
        nc = sum(contact)
        fdes = np.tile([0, 0, 13.741 * 9.81 / nc], nc)
        A, lb, ub, P, w, (M, Cv, g, Js) = wbc.build_qp(robot, CFG, q, v, contact, q_des, v_des, fdes)
        assert A.shape == (6 + 7 * nc + 12 + nc, 18 + 3 * nc)
        x, status = wbc.solve_qp(A, lb, ub, P, w, qp_solve)
        assert status == 0
        Ax = A @ x
        assert np.all(Ax <= ub + 1e-7) and np.all(Ax >= lb - 1e-7)
        # stationarity: P x + w is a combination of the rows that are active
        act = (np.abs(Ax - ub) < 1e-6) | (np.abs(Ax - lb) < 1e-6)
        grad = P @ x + w
        coef = np.linalg.lstsq(A[act].T, -grad, rcond=None)[0]
        assert np.abs(A[act].T @ coef + grad).max() < 1e-5 * max(1.0, np.abs(grad).max())
        # the base dynamics hold with the solution: M a + C v + g = Js' f on the unactuated rows
        a, f = x[:18], x[18:]
        assert np.abs((M @ a + Cv + g - Js.T @ f)[:6]).max() < 1e-7
