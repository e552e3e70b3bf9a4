"""ORACLE (test infrastructure, NOT product code): numpy restatement of the reference's trajectory -> whole-body-target step
(SURVEY.md section 8 row f3).

Follows /root/reference/mpc/models/single_rigid_body_model.cpp:314-425 (InverseKinematics), :430-441 (ComputeJacobianForIK),
:443-455 (GetEndEffectorLocations) and /root/reference/controllers/mpc_controller.cpp:414-511 (GetTargetsFromTraj).  The reference
delegates the kinematics to pinocchio (absent, unpinned: SURVEY.md 8c); its published algorithms are restated here on generic SE(3)
objects (4x4 homogeneous matrices, 6x6 adjoint-free formulas of pinocchio/spatial/{explog,log}.hxx): forwardKinematics over the
URDF chain, computeFrameJacobian in the LOCAL frame, log6, Jlog6, integrate of the free-flyer.  PARITY UNPINNED: the reference
holds no fixture for this step; tests pin it by FK o IK round trips, finite differences of the Jacobian and the error map.
Pure Python / numpy: small cases only."""
import numpy as np

EPS, IT_MAX, DT, DAMP = 5e-6, 1000, 1e-1, 1e-6


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    t = np.trace(R)
    q = np.zeros(4)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q[3] = 0.25 * s; q[0] = (R[2, 1] - R[1, 2]) / s; q[1] = (R[0, 2] - R[2, 0]) / s; q[2] = (R[1, 0] - R[0, 1]) / s
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q[i] = 0.25 * s; q[3] = (R[k, j] - R[j, k]) / s; q[j] = (R[j, i] + R[i, j]) / s; q[k] = (R[k, i] + R[i, k]) / s
    return q


def quat_mul(a, b):
    av, aw, bv, bw = a[:3], a[3], b[:3], b[3]
    return np.concatenate([aw * bv + bw * av + np.cross(av, bv), [aw * bw - av @ bv]])


def first_order_normalize(q):
    return q * (3.0 - q @ q) / 2.0


def log3(R):
    c = min(max((np.trace(R) - 1) / 2, -1.0), 1.0)
    th = np.arccos(c)
    a = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-8:
        return 0.5 * a, th
    return th / (2 * np.sin(th)) * a, th


def exp6(v, w):
    th = np.linalg.norm(w)
    K = skew(w)
    if th < 1e-8:
        a, b, c = 1 - th ** 2 / 6, 0.5 - th ** 2 / 24, 1.0 / 6 - th ** 2 / 120
    else:
        a, b, c = np.sin(th) / th, (1 - np.cos(th)) / th ** 2, (th - np.sin(th)) / th ** 3
    return np.eye(3) + a * K + b * K @ K, (np.eye(3) + b * K + c * K @ K) @ v


def log6(R, p):
    w, th = log3(R)
    if th < 1e-8:
        alpha, beta = 1 - th ** 2 / 12 - th ** 4 / 720, 1.0 / 12 + th ** 2 / 720
    else:
        st, ct = np.sin(th), np.cos(th)
        alpha, beta = th * st / (2 * (1 - ct)), 1 / th ** 2 - st / (2 * th * (1 - ct))
    return np.concatenate([alpha * p - 0.5 * np.cross(w, p) + beta * (w @ p) * w, w])


def jlog6(R, p):
    w, th = log3(R)
    t2 = th * th
    if th < 1e-8:
        alpha, diag, beta, bdot = 1.0 / 12 + t2 / 720, 0.5 * (2 - t2 / 6), 1.0 / 12 + t2 / 720, 1.0 / 360
    else:
        st, ct = np.sin(th), np.cos(th)
        i22 = 1 / (2 * (1 - ct))
        alpha = 1 / t2 - st / th * i22; diag = 0.5 * (th * st / (1 - ct))
        beta = alpha; bdot = -2 / t2 ** 2 + (1 + st / th) / t2 * i22
    A = alpha * np.outer(w, w) + diag * np.eye(3) + skew(0.5 * w)
    wTp = w @ p
    v3 = (bdot * wTp) * w - (t2 * bdot + 2 * beta) * p
    C = np.outer(v3, w) + beta * np.outer(w, p) + wTp * beta * np.eye(3) + skew(0.5 * p)
    J = np.zeros((6, 6))
    J[:3, :3] = A; J[3:, 3:] = A; J[:3, 3:] = C @ A
    return J


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def leg_chain(origins, ang):
    """joint frames of one leg in the base frame: list of (R, p) for hip, thigh, calf joints (after rotation) and the foot frame"""
    R, p = np.eye(3), np.zeros(3)
    out = []
    for k in range(3):
        p = p + R @ np.asarray(origins[k], float)
        out.append((R.copy(), p.copy(), (R @ np.eye(3)[0 if k == 0 else 1]).copy()))      # frame before the rotation: centre + axis
        R = R @ rot(0 if k == 0 else 1, ang[k])
    p = p + R @ np.asarray(origins[3], float)
    return out, R, p


def forward_kinematics(legs, q):
    Rb = quat_to_R(q[3:7])
    return np.array([q[:3] + Rb @ leg_chain(legs[ee], q[7 + 3 * ee:10 + 3 * ee])[2] for ee in range(4)])


def frame_jacobian_local(legs, q, ee):
    """linear rows of pinocchio::computeFrameJacobian(..., LOCAL) for the foot of leg ee: 3 x 18"""
    joints, R3, r = leg_chain(legs[ee], q[7 + 3 * ee:10 + 3 * ee])
    J = np.zeros((3, 18))
    J[:, 0:3] = R3.T
    J[:, 3:6] = -R3.T @ skew(r)
    for k, (_, c, ax) in enumerate(joints):
        J[:, 6 + 3 * ee + k] = R3.T @ np.cross(ax, r - c)
    return J


def integrate(q, v, h):
    dR, dt = exp6(h * v[:3], h * v[3:6])
    Rb = quat_to_R(q[3:7])
    out = q.copy()
    out[:3] = q[:3] + Rb @ dt
    nq = quat_mul(q[3:7], R_to_quat(dR))
    if nq @ q[3:7] < 0:
        nq = -nq
    out[3:7] = first_order_normalize(nq)
    out[7:] = q[7:] + h * v[6:]
    return out


def ik_error_and_jacobian(legs, q, ee, p_des, R_des, e_des):
    Rb = quat_to_R(q[3:7])
    _, R3, r = leg_chain(legs[ee], q[7 + 3 * ee:10 + 3 * ee])
    Rf, pf = Rb @ R3, q[:3] + Rb @ r
    Re, pe = Rb.T @ R_des, Rb.T @ (p_des - q[:3])
    err = np.concatenate([Rf.T @ (e_des - pf), log6(Re, pe)])
    J = np.zeros((9, 18))
    J[:3] = -frame_jacobian_local(legs, q, ee)
    J[3:, :6] = -jlog6(Re.T, -Re.T @ pe)
    return err, J


def inverse_kinematics(legs, state13, ee_des, q_guess):
    """returns (q, iterations per foot, converged)"""
    q = np.array(q_guess, float).copy()
    q[:3] = state13[:3]; q[3:7] = state13[6:10]
    R_des, p_des = quat_to_R(state13[6:10]), np.asarray(state13[:3], float)
    # `bool success = false;` is declared ONCE, before the foot loop (single_rigid_body_model.cpp:356), and `if (!success) throw` sits inside it
    # (:414-416): once one foot has converged, a later foot that stops at IT_MAX no longer throws.  As coded: failure is reported iff no foot
    # up to and including the current one has converged, i.e. iff foot 0 fails (the per-foot iteration counts still show a foot at IT_MAX).
    iters, ok = [], True
    success = False
    for ee in range(4):
        it = 0
        for it in range(IT_MAX):
            q[3:7] = first_order_normalize(q[3:7])
            err, J = ik_error_and_jacobian(legs, q, ee, p_des, R_des, np.asarray(ee_des[ee], float))
            if np.linalg.norm(err) < EPS:
                success = True
                break
            JJt = J @ J.T + DAMP * np.eye(9)
            v = -(J.T @ np.linalg.solve(JJt, err))
            q = integrate(q, v, DT)
        else:
            it = IT_MAX
        iters.append(it)
        ok = ok and success
    return q, iters, ok


def targets_from_traj(legs, states, t0, dt, mass, Ir_inv, ee_at, force_at, time, q_des):
    """MPCController::GetTargetsFromTraj: states [(N+1) x 13], ee_at(ee, t) / force_at(ee, t): spline lookups of the trajectory"""
    time = max(time, t0)
    node = int(np.ceil((time - t0) / dt))
    T = lambda k: t0 + dt * k
    S = np.asarray(states, float)
    if node > 0:
        s1 = (S[node] - S[node - 1]) * (1 - (T(node) - time) / (T(node) - T(node - 1))) + S[node - 1]
        s2 = (S[node + 1] - S[node]) * (1 - (T(node + 1) - (time + dt)) / (T(node + 1) - T(node))) + S[node]
    else:
        s1 = (S[1] - S[0]) * (1 - (T(1) - time) / (T(1) - T(0))) + S[0]
        s2 = (S[1] - S[0]) * (1 - (T(1) - (time + dt)) / (T(1) - T(0))) + S[0]
    q1, _, ok1 = inverse_kinematics(legs, s1, [ee_at(e, time) for e in range(4)], q_des)
    q2, _, ok2 = inverse_kinematics(legs, s2, [ee_at(e, time + dt) for e in range(4)], q1)
    v = np.zeros(18)
    v[:3] = s1[3:6] / mass
    v[3:6] = Ir_inv @ s1[10:13]
    v[6:] = (q2 - q1)[7:] / dt
    return q1, v, np.array([force_at(e, time) for e in range(4)]), ok1 and ok2
