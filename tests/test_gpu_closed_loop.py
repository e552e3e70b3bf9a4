"""Closed-loop rollout harness through the C-ABI: srbm_plant_*, srbm_closed_loop_advance.

The plant is RKIntegrator::CalcIntegral (/root/reference/mpc/rk_integrator.cpp:14-30) over SingleRigidBodyModel::CalcDynamics
(/root/reference/mpc/models/single_rigid_body_model.cpp:222-256) under the forces / foot locations of the current
trajectory; the oracle restates both (oracle/srbm_traj_model.hpp: CalcDynamics, CalcIntegral).  Tolerances: plant states and
trajectories <= 1e-4 relative (the north-star tolerance; a plant step alone agrees to 1e-12)."""
import numpy as np
import pytest

from oracle_py import OracleMPC, load_config
from srbm_loader import host
from srbm_loader.workloads import config_b_instance, config_d_instance

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def oracle_closed_loop(cfg, state, ee, steps, substeps, advance_time, push_time, impulse):
    o = OracleMPC(cfg); o.set_warmstart(state); o.initial_run(state, ee)
    dt = cfg['integrator_dt']
    x = np.array(state, float)
    plant = []
    for i in range(steps):
        t = i * dt
        x = o.plant_integrate(x, t, dt / substeps, substeps, advance_time)
        if t < push_time <= t + dt:
            x[3:6] += impulse[:3]; x[10:13] += impulse[3:]
        plant.append(x.copy())
        eev = np.array([[o.ee_value(e, 1, c, t + dt) for c in range(3)] for e in range(4)])
        o.rti(x, t + dt, eev)
    return o, np.array(plant)


@pytest.mark.parametrize('advance_time', [0, 1])
def test_closed_loop_rollout_matches_oracle(advance_time):
    cfg = load_config()
    B, K, SUB = 3, 6, 5
    states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    push_time = np.array([0.12, 0.07, 1e9])                      # instance 2 is never pushed
    impulse = np.array([[2.5, -1.0, 0.3, 0.05, -0.1, 0.2], [-1.5, 2.0, 0.0, 0.0, 0.1, -0.1], [9, 9, 9, 9, 9, 9]], float)
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
    g.create_initial_run(states, ees)
    g.plant_set_state(states); g.plant_set_push(push_time, impulse)
    for i in range(K):                                           # one step per call: the plant state after every step is compared
        g.closed_loop_advance(i, 1, SUB, advance_time); g.synchronize()
        if i == 0: first = g.plant_state()
    st, err = g.status()
    assert np.all(err == 0)
    xs, tr = g.plant_state(), g.trajectory_states()
    for b in range(B):
        o, plant = oracle_closed_loop(cfg, states[b], ees[b].reshape(4, 3), K, SUB, advance_time, push_time[b], impulse[b])
        assert relerr(first[b], plant[0]) < 1e-6                 # first plant step: the trajectories of the two cold starts agree to ~1e-6 (observed 8e-8)
        assert relerr(xs[b], plant[-1]) < REL_TOL
        assert relerr(tr[b], o.states()) < REL_TOL
        assert (int(st[b]) in (0, 1, 2)) == (int(o.stats()['status']) in (0, 1, 2))
    # the pushed instances left the unpushed path: the push is visible in the plant momentum
    assert np.abs(xs[0, 3:6] - xs[2, 3:6]).max() > 0.5


def test_closed_loop_fused_steps_equal_single_steps_and_push_distribution():
    """K closed-loop steps in one launch == K launches of one step (bitwise); a Config-D style batch (N = 50, pushes drawn per
    instance) runs without error bits and stays finite"""
    cfg = load_config()
    B, K = 8, 6
    states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    rng = np.random.default_rng(5)
    pt = rng.uniform(0.0, 0.3, B); imp = rng.normal(0, 1.0, (B, 6)) * np.array([2.5, 2.5, 0.5, 0.2, 0.2, 0.2])
    res = []
    for one_launch in (True, False):
        g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.create_initial_run(states, ees)
        g.plant_set_state(states); g.plant_set_push(pt, imp)
        if one_launch:
            g.closed_loop_advance(0, K, 4, True)
        else:
            for i in range(K): g.closed_loop_advance(i, 1, 4, True)
        g.synchronize()
        res.append((g.plant_state(), g.trajectory_states(), g.qp_solution(), g.status()[0]))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    # without a plant state the entry refuses loudly
    g = host.BatchMPC(cfg, 2)
    with pytest.raises(RuntimeError):
        g.closed_loop_advance(0, 1, 1, False)
    # Config D sizes
    cfgd = load_config('a1_config_distr_rejection')
    B = 16
    states, ees = zip(*[config_d_instance(cfgd, b) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    g = host.BatchMPC(cfgd, B); g.set_state_trajectory_warm_start(states); g.create_initial_run(states, ees)
    g.plant_set_state(states)
    g.plant_set_push(rng.uniform(0.0, 0.1, B), rng.normal(0, 1.0, (B, 6)) * np.array([2.5, 2.5, 0.3, 0.1, 0.1, 0.2]))
    g.closed_loop_advance(0, 10, 4, True); g.synchronize()
    st, err = g.status()
    assert np.all(err == 0) and np.all(np.isfinite(g.plant_state())) and np.all(np.isin(st, (0, 1, 2)))
