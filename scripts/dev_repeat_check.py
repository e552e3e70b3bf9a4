"""is a solve that is REPEATED after a failed lower-start attempt the same as a solve without an attempt?  step 0 of the re-synchronised protocol"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from oracle_py import OracleMPC, load_config
from srbm_loader import host
from bench import config_b_instance
cfg = load_config(); B = 256
states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200); g.enable_fast_termination()
oracles = []
for b in range(B):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); oracles.append(o)
pool = ThreadPoolExecutor(16)
list(pool.map(lambda b: oracles[b].initial_run(states[b], ees[b].reshape(4, 3)), range(B)))
g.create_initial_run(states, ees.reshape(B, 12))
g.set_warm_start_trajectory((host.Trajectory * B)(*[o.trajectory_record(host) for o in oracles]))
c = g.clone(); c.set_solver_step_rule(host.FAST_TOL_STEP, 0.0)          # same rule, no attempt
g.rti_advance(0, 1); c.rti_advance(0, 1); g.synchronize(); c.synchronize()
xg, xc = g.raw_qp_minimiser(), c.raw_qp_minimiser()
fl = g.solve_flags(); sg, sc = g.stats(), c.stats(); stg, stc = g.status()[0], c.status()[0]
rep = (fl & 4) != 0
same = np.array([np.array_equal(xg[b], xc[b]) for b in range(B)])
print('repeated solves: %d, of them bitwise equal to the solve without an attempt: %d' % (rep.sum(), (rep & same).sum()))
for b in np.nonzero(rep & ~same)[0][:12]:
    print('  inst %3d: status %d / %d  iters %d / %d  gap %.1e / %.1e  max diff %.2e' % (b, stg[b], stc[b], sg[b, 4], sc[b, 4], sg[b, 7], sc[b, 7], np.abs(xg[b] - xc[b]).max()))
b = 38
print('inst 38: flags', fl[b], 'status', stg[b], stc[b], 'iters', sg[b, 4], sc[b, 4], 'gap', sg[b, 7], sc[b, 7], 'err bits', g.status()[1][b], c.status()[1][b])
