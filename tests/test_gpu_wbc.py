"""GPU parity of the batched whole-body QP controller (second half; csrc/srbm_wbc.hiph through the C-ABI) against
the numpy restatement oracle/wbc_numpy.py (pinned by tests/test_oracle_wbc.py; the reference holds no fixture for this step).
The assembled QP (A, lb, ub, P, w in the reference's row / column order) <= 1e-9, the minimiser and the torques <= 1e-6 relative."""
import os
import sys

import numpy as np
import pytest

from oracle_py import load_config, qp_solve
from srbm_loader import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ik_numpy as ik
import wbc_numpy as wbc

pytestmark = pytest.mark.gpu
CONTACTS = [[1, 1, 1, 1], [1, 0, 0, 1], [0, 1, 1, 0], [1, 1, 1, 0], [0, 0, 1, 0], [0, 0, 0, 0]]


def make(B, seed=3):
    cfg = load_config()
    rng = np.random.default_rng(seed)
    q0 = np.array(cfg['init_config'], float)
    q = np.tile(q0, (B, 1))
    q[:, :3] += rng.normal(size=(B, 3)) * 0.03
    quat = q[:, 3:7] + np.concatenate([rng.normal(size=(B, 3)) * 0.05, np.zeros((B, 1))], axis=1)
    q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    q[:, 7:] += rng.normal(size=(B, 12)) * 0.1
    v = rng.normal(size=(B, 18)) * 0.1
    q_des = np.tile(q0, (B, 1)); q_des[:, 7:] += rng.normal(size=(B, 12)) * 0.02
    v_des = rng.normal(size=(B, 18)) * 0.05
    return cfg, q, v, q_des, v_des, rng


def test_assembled_qp_and_solution_match_the_oracle():
    B = len(CONTACTS) * 2
    cfg, q, v, q_des, v_des, rng = make(B)
    contact = np.array([CONTACTS[b % len(CONTACTS)] for b in range(B)], np.int32)
    fdes = np.zeros((B, 12))
    for b in range(B):
        nc = contact[b].sum()
        if nc:
            fdes[b, :3 * nc] = np.tile([1.0, -2.0, cfg['mass'] * 9.81 / nc], nc)
    g = host.BatchMPC(cfg, B)
    ctl, sol, st, iters, qp = g.qp_control(q, v, contact, q_des, v_des, fdes, dump=True)
    robot = wbc.Robot(cfg)
    for b in range(B):
        nc = int(contact[b].sum())
        n, m = 18 + 3 * nc, 6 + 7 * nc + 12 + nc
        A, lb, ub, P, w, (M, Cv, gg, Js) = wbc.build_qp(robot, cfg, q[b], v[b], contact[b], q_des[b], v_des[b], fdes[b, :3 * nc])
        assert np.abs(qp['A'][b, :m, :n] - A).max() < 1e-9, b
        assert np.all(qp['A'][b, m:, :] == 0) and np.all(qp['A'][b, :, n:] == 0)
        fin = np.abs(lb) < 1e29
        assert np.abs(qp['lb'][b, :m][fin] - lb[fin]).max() < 1e-8 and np.all(qp['lb'][b, :m][~fin] < -1e29), b
        assert np.abs(qp['ub'][b, :m] - ub).max() < 1e-8, b
        assert np.abs(qp['P'][b, :n] - np.diag(P)).max() < 1e-12 and np.abs(qp['w'][b, :n] - w).max() < 1e-7 * max(1.0, np.abs(w).max()), b
        xo, so = wbc.solve_qp(A, lb, ub, P, w, qp_solve)
        assert st[b] <= 1 and so == 0, (b, st[b], so, iters[b])
        scale = max(1.0, np.abs(xo).max())
        assert np.abs(sol[b, :n] - xo).max() / scale < 1e-6, (b, np.abs(sol[b, :n] - xo).max() / scale)
        co, _, _ = wbc.control_action(robot, cfg, q[b], v[b], contact[b], q_des[b], v_des[b], fdes[b, :3 * nc], qp_solve)
        assert np.abs(ctl[b] - co).max() < 1e-6 * max(1.0, np.abs(co).max()), b
        assert np.array_equal(ctl[b, :12], q_des[b, 7:]) and np.array_equal(ctl[b, 12:24], v_des[b, 6:])


def test_full_batch_controller_properties():
    """256 instances: every QP solves; the solution satisfies the floating-base dynamics, the torque limits, the friction pyramid and the
    force bounds; standing still at the nominal configuration with the weight as the force target needs torques that hold the weight"""
    B = 256
    cfg, q, v, q_des, v_des, rng = make(B, seed=9)
    contact = np.array([CONTACTS[b % 3] for b in range(B)], np.int32)
    fdes = np.zeros((B, 12))
    for b in range(B):
        nc = contact[b].sum()
        fdes[b, :3 * nc] = np.tile([0, 0, cfg['mass'] * 9.81 / nc], nc)
    g = host.BatchMPC(cfg, B)
    ctl, sol, st, iters, qp = g.qp_control(q, v, contact, q_des, v_des, fdes, dump=True)
    assert np.all(st <= 1), np.unique(st, return_counts=True)          # Solved (or, at the noise floor, SolvedInacc)
    print('whole-body QP, 256 instances: IPM iterations min %d median %d max %d, statuses %s' % (iters.min(), np.median(iters), iters.max(), dict(zip(*np.unique(st, return_counts=True)))))
    assert iters.max() < 40
    tb = np.array(cfg['torque_bounds'], float)
    for b in range(0, B, 7):
        nc = int(contact[b].sum()); n, m = 18 + 3 * nc, 6 + 7 * nc + 12 + nc
        Ax = qp['A'][b, :m, :n] @ sol[b, :n]
        assert np.all(Ax <= qp['ub'][b, :m] + 1e-7) and np.all(Ax >= qp['lb'][b, :m] - 1e-7), b
        assert np.all(np.abs(ctl[b, 24:]) <= tb + 1e-6)
        f = sol[b, 18:18 + 3 * nc].reshape(nc, 3)
        assert np.all(f[:, 2] >= -1e-7) and np.all(np.abs(f[:, :2]) <= cfg['friction_coef'] * f[:, 2:3] + 1e-6)
    # standing still on four feet at the nominal pose
    q0 = np.array(cfg['init_config'], float)
    fz = cfg['mass'] * 9.81 / 4
    ctl, sol, st, iters = g.qp_control(q0, np.zeros(18), [1, 1, 1, 1], q0, np.zeros(18), np.tile([0, 0, fz], 4))
    assert st[0] == 0 and np.abs(sol[0, :18]).max() < 0.5             # (almost) no acceleration: the weight is carried
    assert abs(sol[0, 18:].reshape(4, 3)[:, 2].sum() - cfg['mass'] * 9.81) < 1.0
    assert np.all(ctl[0, 24:].reshape(4, 3)[:, 2] != 0)                # the calf joints work against gravity


def test_device_pointer_entries_of_the_control_tick_equal_the_host_pointer_entries():
    """srbm_get_targets_from_traj_dev / srbm_eval_trajectory_dev / srbm_qp_control_dev: the same kernels on buffers that are already in HBM (hipMalloc'ed here through the HIP
    runtime), one launch each and no PCIe hop -- bit-identical to the host-pointer entries"""
    import ctypes as C
    hip = C.CDLL('libamdhip64.so')

    class Dev:
        def __init__(self, a):
            self.a = np.ascontiguousarray(a); self.p = C.c_void_p()
            assert hip.hipMalloc(C.byref(self.p), C.c_size_t(self.a.nbytes)) == 0
            assert hip.hipMemcpy(self.p, self.a.ctypes.data_as(C.c_void_p), C.c_size_t(self.a.nbytes), 1) == 0      # hipMemcpyHostToDevice (synchronous)

        def get(self):
            out = np.empty_like(self.a)
            assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.p, C.c_size_t(self.a.nbytes), 2) == 0
            return out

        def __del__(self):
            hip.hipFree(self.p)
    B = 32
    cfg, q, v, q_des, v_des, rng = make(B, seed=21)
    from srbm_loader.workloads import config_b_instance
    states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    g = host.BatchMPC(cfg, B)
    g.set_state_trajectory_warm_start(states)
    g.create_initial_run(states, ees)
    t0 = g.get_trajectory(0, 1)[0].init_time + 2e-3
    q0 = np.tile(np.array(cfg['init_config'], float), (B, 1))
    qh, vh, fh, sth = g.get_targets_from_traj(t0, q0)
    tt, tq, tv, tf, ts = Dev(np.full(B, t0)), Dev(q0), Dev(np.zeros((B, 18))), Dev(np.zeros((B, 12))), Dev(np.zeros(B, np.int32))
    g.get_targets_from_traj_dev(tt.p.value, tq.p.value, tv.p.value, tf.p.value, ts.p.value)
    g.synchronize()
    assert np.array_equal(tq.get(), qh) and np.array_equal(tv.get(), vh) and np.array_equal(tf.get().reshape(B, 4, 3), fh) and np.array_equal(ts.get(), sth)
    # srbm_eval_trajectory_dev: forces, foot positions and the contact flags of the same instant, on device buffers
    fe, pe, ce = g.eval_trajectory(t0)
    ef, ep, ec = Dev(np.zeros((B, 12))), Dev(np.zeros((B, 12))), Dev(np.zeros((B, 4), np.int32))
    g.eval_trajectory_dev(tt.p.value, ef.p.value, ep.p.value, ec.p.value)
    g.synchronize()
    assert np.array_equal(ef.get().reshape(fe.shape), fe) and np.array_equal(ep.get().reshape(pe.shape), pe) and np.array_equal(ec.get().reshape(ce.shape), ce)
    contact = np.array([CONTACTS[b % 3] for b in range(B)], np.int32)
    fdes = np.zeros((B, 12))
    for b in range(B):
        nc = contact[b].sum()
        fdes[b, :3 * nc] = np.tile([0, 0, cfg['mass'] * 9.81 / nc], nc)
    ctl, sol, st, iters = g.qp_control(q, v, contact, q_des, v_des, fdes)
    ins = [Dev(q), Dev(v), Dev(contact), Dev(q_des), Dev(v_des), Dev(fdes)]
    tc, tsol, tst = Dev(np.zeros((B, 36))), Dev(np.zeros((B, 30))), Dev(np.zeros(B, np.int32))
    g.qp_control_dev(*[a.p.value for a in ins], tc.p.value, tsol.p.value, tst.p.value)
    g.synchronize()
    assert np.array_equal(tc.get(), ctl) and np.array_equal(tsol.get(), sol)
    assert np.array_equal(tst.get() & 0xff, st) and np.array_equal(tst.get() >> 8, iters)


def test_a_clone_carries_the_whole_body_model_and_close_releases_borrowers_first():
    """srbm_batch_clone copies the complete state, the whole-body model included (round 2: a clone failed srbm_qp_control with "whole-body model
    has not been set"); BatchMPC.close() releases the gait optimisers that borrow the batch before the batch (round 2: the library refused the
    destroy, the binding dropped its handle and the device batch leaked)"""
    B = 4
    cfg, q, v, q_des, v_des, rng = make(B, seed=5)
    contact = np.ones((B, 4), np.int32)
    fdes = np.tile([0, 0, cfg['mass'] * 9.81 / 4], (B, 4))
    g = host.BatchMPC(cfg, B)
    a = g.qp_control(q, v, contact, q_des, v_des, fdes)
    c = g.clone()
    b = c.qp_control(q, v, contact, q_des, v_des, fdes)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    gait = host.BatchGaitOptimizer(g)
    g.close()
    assert not g.h and gait.g is None
    c.close()
    assert not c.h


def test_a_row_structure_the_assembly_does_not_cover_is_reported():
    """The iteration's assembly relies on the controller's row order (12 two-sided torque rows, 4 friction rows and one two-sided force row per foot in
    contact).  A zero torque bound turns a torque row into an equality and breaks that order: the kernel reports it (status 8, zero control action --
    the path of "Could not solve WBC QP. Returning 0 control action.", qp_control.cpp:88-92) instead of assembling the wrong matrix"""
    B = 4
    cfg, q, v, q_des, v_des, rng = make(B, seed=11)
    cfg = dict(cfg); cfg['torque_bounds'] = list(cfg['torque_bounds']); cfg['torque_bounds'][3] = 0.0
    contact = np.ones((B, 4), np.int32)
    fdes = np.tile([0, 0, cfg['mass'] * 9.81 / 4], (B, 4))
    g = host.BatchMPC(cfg, B)
    ctl, sol, st, iters = g.qp_control(q, v, contact, q_des, v_des, fdes)
    assert np.all(st == 8) and np.all(ctl == 0) and np.all(sol == 0)
