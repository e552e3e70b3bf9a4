"""SURVEY.md section 8 row a13 on the device, tested DIRECTLY: the contact-time partials of the QP
(MPCSingleRigidBody::ComputeParamPartialsClarabel, /root/reference/mpc/mpc_single_rigid_body.cpp:642-792) through
srbm_gait_get_param_partials, which runs the same per-item code the gradient kernel contracts.

(1) the reference's own procedure, /root/reference/test/mpc_test.cpp:114-270 ("Model Partials"): perturb ONE contact time by sqrt(1e-16)
    through MPC::UpdateContactTimes, finite-difference the assembled constraint matrix (srbm_export_qp) and compare the dynamics, force-box
    and friction-cone blocks entry by entry at abs 1e-4 -- for every (foot, contact index >= 1) of the schedule, at N = 20 and N = 50;
(2) entry-wise against the oracle's restatement (OracleMPC.param_partials), all four outputs dA, dG, db, dh.
One batch does all finite differences at once: instance 0 is the unperturbed problem, instance k has contact time k moved."""
import math

import numpy as np
import pytest

from oracle_py import OracleMPC, load_config
from srbm_loader import host

pytestmark = pytest.mark.gpu
EE0 = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)
DERIV_MARGIN = 1e-4            # test/mpc_test.cpp:122


@pytest.mark.parametrize('cfgname', ['a1_configuration', 'a1_gait_opt_config'])
def test_contact_time_partials_against_finite_differences_and_the_oracle(cfgname):
    cfg = load_config(cfgname)
    s0 = np.array(cfg['srb_init'], float)
    o = OracleMPC(cfg); o.set_warmstart(s0); o.initial_run(s0, EE0)
    ct = [o.contact_times(e)[0] for e in range(4)]
    pairs = [(ee, idx) for ee in range(4) for idx in range(1, len(ct[ee]))]       # mpc_test.cpp:130-131: idx starts at 1
    assert len(pairs) >= 12
    B = 1 + len(pairs)
    g = host.BatchMPC(cfg, B)
    g.set_state_trajectory_warm_start(s0)
    g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
    g.create_initial_run(s0, EE0)
    # the device's schedule is the oracle's, bit for bit
    tr = g.get_trajectory(0, 1)[0]
    for e in range(4):
        assert np.array_equal(np.array(tr.get_contact_times()[e]), ct[e])

    # ---- analytic partials on the trajectory of the initial run (`Trajectory traj = mpc.GetTrajectory()`, :121) ----
    holder = o.clone()
    analytic = {}
    for (ee, idx) in pairs:
        analytic[(ee, idx)] = g.param_partials(0, ee, idx)
    # (2) entry-wise against the oracle: its partials need the QP data of a solve (`mpc.GetRealTimeUpdate`, :122), the trajectory is `traj`.
    #     Each side evaluates on ITS trajectory of the initial run (ten SQP solves each: node values agree to ~1e-7 relative, the partials
    #     differentiate cubic segments of 0.1 s), hence 1e-5 of the largest entry of each output rather than round-off; the sparsity patterns coincide
    assert o.rti(s0, 0.0, EE0) == 0
    worst = 0.0
    for (ee, idx) in pairs:
        dA, dG, db, dh = analytic[(ee, idx)]
        oA, oG, ob, oh = o.param_partials(ee, idx, traj_src=holder)
        assert dA.shape == oA.shape and dG.shape == oG.shape
        for a, b_ in ((dA, oA), (dG, oG), (db, ob), (dh, oh)):
            sc = max(1.0, np.abs(b_).max())
            assert np.array_equal(np.abs(a) > 1e-5 * sc, np.abs(b_) > 1e-5 * sc), ('pattern', ee, idx)
            worst = max(worst, np.abs(a - b_).max() / sc)
            assert np.abs(a - b_).max() <= 1e-5 * sc, (ee, idx, np.abs(a - b_).max(), sc)
        assert np.abs(dA).max() > 0 or np.abs(dG).max() > 0, (ee, idx)            # every contact time moves something

    # ---- (1) finite differences of the assembled QP, the reference's test ----
    dt = math.sqrt(1e-16)
    maxc = max(len(c) for c in ct)
    times = np.zeros((B, 4, maxc))
    for e in range(4):
        times[:, e, :len(ct[e])] = ct[e]
    for k, (ee, idx) in enumerate(pairs):
        times[1 + k, ee, idx] += dt
    g.update_contact_times(times)                      # mpc2.UpdateContactTimes(mod_times), :132
    g.get_real_time_update(s0, 0.0, EE0)               # :133
    sz = g.sizes()
    assert (sz == sz[0]).all()                         # REQUIRE(num_force_box == ...), REQUIRE(num_cone == ...): sizes do not move
    n, ns = int(sz[0, 0]), int(sz[0, 7])
    ndyn, nfb, ncone = (cfg['num_nodes'] + 1) * 12, 2 * ns, 4 * ns
    A0 = g.export_qp(0)[0]
    worst_fd = 0.0
    for k, (ee, idx) in enumerate(pairs):
        Ak = g.export_qp(1 + k)[0]
        dA, dG, db, dh = analytic[(ee, idx)]
        fd_dyn = (Ak[:ndyn] - A0[:ndyn]) / dt
        fd_fb = (Ak[ndyn:ndyn + nfb] - A0[ndyn:ndyn + nfb]) / dt
        fd_cone = (Ak[ndyn + nfb:ndyn + nfb + ncone] - A0[ndyn + nfb:ndyn + nfb + ncone]) / dt
        for name, an, fd in (('dynamics', dA[:ndyn], fd_dyn), ('force box', dG[:nfb], fd_fb), ('cone', dG[nfb:nfb + ncone], fd_cone)):
            d = np.abs(an - fd).max()
            worst_fd = max(worst_fd, d)
            assert d < DERIV_MARGIN, (name, ee, idx, d, np.unravel_index(np.abs(an - fd).argmax(), an.shape))
        if max(np.abs(dA[:ndyn]).max(), np.abs(dG[:nfb + ncone]).max()) > 1e-2:      # (a contact time beyond the horizon moves nothing)
            assert np.abs(fd_dyn).max() + np.abs(fd_fb).max() + np.abs(fd_cone).max() > 1e-3, (ee, idx)   # the perturbation is seen by the QP
    print('%s: %d contact times; worst |analytic - oracle| %.2e (relative), worst |analytic - FD| %.2e' % (cfgname, len(pairs), worst, worst_fd))


def test_partials_entry_rejects_what_does_not_exist():
    cfg = load_config('a1_configuration')
    s0 = np.array(cfg['srb_init'], float)
    g = host.BatchMPC(cfg, 1)
    g.set_state_trajectory_warm_start(s0)
    with pytest.raises(RuntimeError):
        g.param_partials(0, 0, 1)                      # no QP solved yet
    g.create_initial_run(s0, EE0)
    with pytest.raises(RuntimeError):
        g.param_partials(0, 0, 31)                     # contact index beyond the schedule
    with pytest.raises(RuntimeError):
        g.param_partials(0, 4, 0)
