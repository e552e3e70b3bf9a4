"""which lower-start attempts end through the step rule with a point that is NOT within the parity tolerance?  Re-synchronised steps of Config B through
the fused launch; per solve: relative primal error against the oracle, iterations, residuals / gap of the returned iterate, flags, SQP step norm of
the PREVIOUS solve (how far the linearisation point was from its minimiser)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from oracle_py import OracleMPC, load_config
from srbm_loader import host
from bench import config_b_instance
cfg = load_config(); B = 256; STEPS = int(os.environ.get('STEPS', 4)); dt = cfg['integrator_dt']
states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200); g.enable_fast_termination()
oracles = []
for b in range(B):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); oracles.append(o)
pool = ThreadPoolExecutor(16)
list(pool.map(lambda b: oracles[b].initial_run(states[b], ees[b].reshape(4, 3)), range(B)))
g.create_initial_run(states, ees.reshape(B, 12))
prev_step_norm = g.stats()[:, 3].copy()
for i in range(STEPS):
    t = i * dt
    g.set_warm_start_trajectory((host.Trajectory * B)(*[o.trajectory_record(host) for o in oracles]))
    st_in = np.array([o.states()[1] for o in oracles])
    ee_in = np.array([[[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)] for o in oracles])
    prev_o = np.array([o.stats()['step_norm'] for o in oracles])
    g.rti_advance(i, 1); g.synchronize()
    so = list(pool.map(lambda b: oracles[b].rti(st_in[b], t, ee_in[b]), range(B)))
    x = g.raw_qp_minimiser(); stats = g.stats(); fl = g.solve_flags(); st = g.status()[0]
    err = np.array([np.abs(x[b, :oracles[b].sizes()['n']] - oracles[b].qp_x()).max() / max(1.0, np.abs(oracles[b].qp_x()).max()) if so[b] <= 1 and st[b] <= 1 else 0.0 for b in range(B)])
    order = np.argsort(-err)[:8]
    print('step %d: flags histogram %s' % (i, dict(zip(*np.unique(fl, return_counts=True)))))
    for b in order:
        print('   inst %3d err %.2e iters %3d res_p %.1e res_d %.1e gap %.1e flags %d  oracle step norm of the previous solve %.3e  device step norm now %.3e' % (
            b, err[b], stats[b, 4], stats[b, 5], stats[b, 6], stats[b, 7], fl[b], prev_o[b], stats[b, 3]))
    acc = (fl & 3) == 3          # attempts accepted (step rule, no repeat)
    acc &= (fl & 4) == 0
    print('   accepted attempts: %d; of them gap > 1e-9: %d, > 1e-8: %d, > 1e-7: %d; res_d > 1e-10: %d;  err of accepted with gap <= 1e-9: max %.2e' % (
        acc.sum(), (stats[acc, 7] > 1e-9).sum(), (stats[acc, 7] > 1e-8).sum(), (stats[acc, 7] > 1e-7).sum(), (stats[acc, 6] > 1e-10).sum(),
        err[acc & (stats[:, 7] <= 1e-9)].max() if (acc & (stats[:, 7] <= 1e-9)).any() else 0.0))
