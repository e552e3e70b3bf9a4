"""ORACLE (test infrastructure, NOT product code): numpy restatement of the reference's 1 kHz whole-body QP controller
(/root/reference/controllers/qp_control.cpp:74-135 ComputeControlAction, :156-263 constraints, :265-327 costs, :329-402 set-up,
:404-415 RecoverControlInputs, :417-473 support Jacobian) -- SURVEY.md section 8 row f3, second half.

The reference obtains M, C v, g, the frame Jacobians and the frame accelerations from pinocchio (crba, computeCoriolisMatrix,
computeGeneralizedGravity, getFrameJacobian / getFrameClassicalAcceleration LOCAL_WORLD_ALIGNED) and solves the QP with OSQP
(absent, unpinned: SURVEY.md 8c).  Here the rigid-body algorithms are Featherstone's recursive Newton-Euler algorithm in BODY
coordinates with 6-D spatial vectors ([angular; linear] internally, pinocchio's [linear; angular] at the interface), and the QP goes
to the oracle's generic interior-point restatement (oracle_py.qp_solve).  PARITY UNPINNED: the reference holds no fixture for this
step; tests pin it by the properties of the algorithm (M symmetric positive definite and equal to the kinetic-energy Hessian, g the
gradient of the potential energy, J the derivative of the foot position, Jdot v by finite differences, KKT conditions of the QP).
Pure Python / numpy: small cases only."""
import numpy as np

import ik_numpy as ik

GRAV = 9.81
INF = 1e30


def crm(v):      # spatial cross product (motion), v = [w; v]
    w, l = v[:3], v[3:]
    return np.block([[ik.skew(w), np.zeros((3, 3))], [ik.skew(l), ik.skew(w)]])


def crf(v):      # spatial cross product (force)
    return -crm(v).T


def xlt(E, r):   # Pluecker transform to a frame rotated by E (child = E parent) and translated by r (in parent coordinates)
    return np.block([[E, np.zeros((3, 3))], [-E @ ik.skew(r), E]])


def spatial_inertia(m, c, Ic):
    C = ik.skew(np.asarray(c, float))
    return np.block([[np.asarray(Ic, float) + m * C @ C.T, m * C], [m * C.T, m * np.eye(3)]])


class Robot:
    """floating base + 4 legs x (hip about x, thigh about y, calf about y); bodies: cfg['body_model'], joint origins: cfg['leg_origins']"""

    def __init__(self, cfg):
        self.legs = np.array(cfg['leg_origins'], float)
        self.I = [spatial_inertia(b['mass'], b['com'], b['inertia']) for b in cfg['body_model']]
        self.mass = sum(b['mass'] for b in cfg['body_model'])

    def joint_transforms(self, q):
        """per leg and joint: (X from the parent body frame to this body frame, motion subspace S)"""
        out = []
        for ee in range(4):
            row = []
            for k in range(3):
                axis = 0 if k == 0 else 1
                E = ik.rot(axis, q[7 + 3 * ee + k]).T
                S = np.zeros(6); S[axis] = 1
                row.append((xlt(E, self.legs[ee][k]), S))
            out.append(row)
        return out

    def rnea(self, q, v, a, gravity=True):
        """tau (18, pinocchio order [base force; base torque; joints], base wrench in the base frame) for generalized acceleration a"""
        Rb = ik.quat_to_R(q[3:7])
        vb = np.concatenate([v[3:6], v[0:3]])
        ab = np.concatenate([a[3:6], a[0:3]])
        if gravity:
            ab = ab + np.concatenate([np.zeros(3), Rb.T @ np.array([0, 0, GRAV])])      # a_0 = -a_gravity, in base coordinates
        JT = self.joint_transforms(q)
        f0 = self.I[0] @ ab + crf(vb) @ (self.I[0] @ vb)
        tau = np.zeros(18)
        for ee in range(4):
            vs, as_, fs, Xs = [], [], [], []
            vp, ap = vb, ab
            for k in range(3):
                X, S = JT[ee][k]
                qd, qdd = v[6 + 3 * ee + k], a[6 + 3 * ee + k]
                vi = X @ vp + S * qd
                ai = X @ ap + S * qdd + crm(vi) @ (S * qd)
                Ii = self.I[1 + 3 * ee + k]
                fs.append(Ii @ ai + crf(vi) @ (Ii @ vi))
                vs.append(vi); as_.append(ai); Xs.append(X)
                vp, ap = vi, ai
            for k in (2, 1, 0):
                tau[6 + 3 * ee + k] = JT[ee][k][1] @ fs[k]
                if k > 0:
                    fs[k - 1] = fs[k - 1] + Xs[k].T @ fs[k]
                else:
                    f0 = f0 + Xs[0].T @ fs[0]
        tau[0:3] = f0[3:]; tau[3:6] = f0[:3]
        return tau

    def dynamics_terms(self, q, v):
        """M (18 x 18), C v, g as pinocchio's crba / computeCoriolisMatrix * v / computeGeneralizedGravity"""
        g = self.rnea(q, np.zeros(18), np.zeros(18))
        h = self.rnea(q, v, np.zeros(18))
        M = np.zeros((18, 18))
        for j in range(18):
            e = np.zeros(18); e[j] = 1
            M[:, j] = self.rnea(q, np.zeros(18), e, gravity=False)
        return M, h - g, g

    def foot_jacobian_lwa(self, q, ee):
        """linear rows of getFrameJacobian(LOCAL_WORLD_ALIGNED) for the foot of leg ee"""
        Rb = ik.quat_to_R(q[3:7])
        _, R3, _ = ik.leg_chain(self.legs[ee], q[7 + 3 * ee:10 + 3 * ee])
        return Rb @ R3 @ ik.frame_jacobian_local(self.legs, q, ee)

    def foot_classical_acceleration(self, q, v, ee):
        """getFrameClassicalAcceleration(LOCAL_WORLD_ALIGNED).linear() after forwardKinematics(q, v, 0): Jdot v"""
        vb = np.concatenate([v[3:6], v[0:3]])
        vp, ap = vb, np.zeros(6)
        JT = self.joint_transforms(q)
        for k in range(3):
            X, S = JT[ee][k]
            qd = v[6 + 3 * ee + k]
            vi = X @ vp + S * qd
            ap = X @ ap + crm(vi) @ (S * qd)
            vp = vi
        Xf = xlt(np.eye(3), self.legs[ee][3])
        vf, af = Xf @ vp, Xf @ ap
        Rb = ik.quat_to_R(q[3:7])
        _, R3, _ = ik.leg_chain(self.legs[ee], q[7 + 3 * ee:10 + 3 * ee])
        return Rb @ R3 @ (af[3:] + np.cross(vf[:3], vf[3:]))


def log3_quat(q):
    """pinocchio::quaternion::log3 (xyzw)"""
    v = np.asarray(q[:3], float); n = np.linalg.norm(v); w = q[3]
    if n < 1e-8:
        return (2.0 / w) * (1.0 - n * n / (3.0 * w * w)) * v
    th = 2.0 * np.arctan2(n, w) if w >= 0 else -2.0 * np.arctan2(n, -w)
    return th / n * v


def exp3_quat(v):
    t = np.linalg.norm(v)
    if t > 1.220703125e-4:
        return np.concatenate([np.sin(t / 2) / t * v, [np.cos(t / 2)]])
    return np.concatenate([(0.5 - t * t / 48) * v, [1.0 - t * t / 8]])


def quat_inv(q):
    return np.concatenate([-q[:3], [q[3]]]) / (q @ q)


def build_qp(robot, cfg, q, v, contact, q_des, v_des, force_des):
    """QPControl::UpdateConstraintsAndCost (:329-402).  contact: 4 bools (measured AND desired, the controller passes the same set);
    force_des: [3 * num_contacts] stacked over the feet in contact.  Returns A, lb, ub, P, w and the dynamics terms."""
    idx = [i for i in range(4) if contact[i]]
    nc = len(idx)
    nv, na = 18, 12
    n = nv + 3 * nc
    m = 6 + 7 * nc + na + nc
    M, Cv, g = robot.dynamics_terms(q, v)
    Js = np.vstack([robot.foot_jacobian_lwa(q, i) for i in idx]) if nc else np.zeros((0, nv))
    A = np.zeros((m, n)); lb = np.zeros(m); ub = np.zeros(m)
    # dynamics of the floating base (:178-192)
    lb[:6] = -(g[:6] + Cv[:6]); ub[:6] = lb[:6]
    A[:6, :nv] = M[:6]
    A[:6, nv:] = -Js.T[:6]
    # contact motion (:194-219).  As coded the right-hand side of foot i is written at row 6 + 3 i (i = the FOOT index), the
    # equality block spans rows 6 .. 6 + 3 nc: with all four feet in contact the two coincide
    for i in range(4):
        if contact[i] and 6 + 3 * i + 3 <= m:
            lb[6 + 3 * i:9 + 3 * i] = -robot.foot_classical_acceleration(q, v, i)
    ub[6:6 + 3 * nc] = lb[6:6 + 3 * nc]
    A[6:6 + 3 * nc, :nv] = Js
    # torque limits (:221-237)
    r0 = 6 + 3 * nc
    A[r0:r0 + na, :nv] = M[6:]
    A[r0:r0 + na, nv:] = -Js.T[6:]
    tb = np.asarray(cfg['torque_bounds'], float)
    lb[r0:r0 + na] = -(Cv + g)[6:] - tb; ub[r0:r0 + na] = -(Cv + g)[6:] + tb
    # friction pyramid (:239-257) and 0 <= f_z <= max_grf (:259-267)
    mu = cfg['friction_coef']
    r1 = r0 + na
    pyr = np.array([[1, 0, -mu], [-1, 0, -mu], [0, 1, -mu], [0, -1, -mu]], float)
    for i in range(nc):
        A[r1 + 4 * i:r1 + 4 * i + 4, nv + 3 * i:nv + 3 * i + 3] = pyr
        lb[r1 + 4 * i:r1 + 4 * i + 4] = -INF
        A[r1 + 4 * nc + i, nv + 3 * i + 2] = 1
        ub[r1 + 4 * nc + i] = cfg['force_bound']
    # costs (:269-327)
    P = np.zeros((n, n)); w = np.zeros(n)
    kv_pos, kp_pos = cfg['base_pos_gains']; kv_ang, kp_ang = cfg['base_ang_gains']
    kp_j, kv_j = np.asarray(cfg['kp_joint_gains'], float), np.asarray(cfg['kd_joint_gains'], float)
    lw, tw, fw = cfg['leg_tracking_weight'], cfg['torso_tracking_weight'], cfg['force_tracking_weight']
    P[6:18, 6:18] = lw * 2 * np.eye(12)
    w[6:18] = -2 * lw * (kv_j * (v_des[6:] - v[6:]) + kp_j * (q_des[7:] - q[7:]))
    P[:6, :6] = tw * 2 * np.eye(6)
    w[:3] = -2 * tw * (kv_pos * (v_des[:3] - v[:3]) + kp_pos * (q_des[:3] - q[:3]))
    ori = q[3:7] / np.linalg.norm(q[3:7]); des = q_des[3:7] / np.linalg.norm(q_des[3:7])
    vel_frame = log3_quat(ik.quat_mul(quat_inv(ori), exp3_quat(v_des[3:6])))
    w[3:6] = -2 * tw * (kv_ang * (vel_frame - v[3:6]) + kp_ang * log3_quat(ik.quat_mul(quat_inv(ori), des)))
    P[18:, 18:] = fw * 2 * np.eye(3 * nc)
    w[18:] = -2 * fw * np.asarray(force_des, float)
    return A, lb, ub, P, w, (M, Cv, g, Js)


def solve_qp(A, lb, ub, P, w, qp_solve):
    """lb <= A x <= ub through the oracle's cone solver: equality rows -> zero cone, finite bounds -> non-negative rows"""
    eq = lb == ub
    rows, rhs = [A[eq]], [ub[eq]]
    up = (~eq) & (ub < INF / 2); lo = (~eq) & (lb > -INF / 2)
    rows += [A[up], -A[lo]]; rhs += [ub[up], -lb[lo]]
    Ac, bc = np.vstack(rows), np.concatenate(rhs)
    r = qp_solve(P, w, Ac, bc, [(0, int(eq.sum())), (1, int(up.sum() + lo.sum()))], tol_gap=1e-12, tol_feas=1e-12)
    return r['x'], r['status']


def control_action(robot, cfg, q, v, contact, q_des, v_des, force_des, qp_solve):
    """QPControl::ComputeControlAction: [q targets (12), v targets (12), torques (12)], the QP solution and its status"""
    A, lb, ub, P, w, (M, Cv, g, Js) = build_qp(robot, cfg, q, v, contact, q_des, v_des, force_des)
    x, status = solve_qp(A, lb, ub, P, w, qp_solve)
    nc = sum(bool(c) for c in contact)
    tau = (M @ x[:18] - Js.T @ x[18:18 + 3 * nc] + Cv + g)[6:]
    return np.concatenate([q_des[7:], v_des[6:], tau]), x, status
