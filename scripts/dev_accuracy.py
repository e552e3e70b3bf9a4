"""GPU scratch driver: accuracy of the device IPM against the oracle on IDENTICAL QPs (device re-synchronised to the oracle's
trajectory before each step), for several gap tolerances.  The oracle's minimisers are within ~1e-7 of the certified
(active-set polished, tests/qp_polish.py) minimiser, so they serve as the truth here."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from oracle_py import OracleMPC, load_config
from srbm_loader import host
from bench import config_b_instance
cfg = load_config()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# variants: gap tolerance[:tol_step[:start_mu]] (srbm_set_solver_tolerances / srbm_set_solver_step_rule), comma separated
tols = [tuple(float(x) for x in v.split(':')) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [(1e-13,), (1e-14,), (1e-15,)]
tols = [t + (0.0,) * (3 - len(t)) for t in tols]
states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
pool = ThreadPoolExecutor(16)
os_ = []
for b in range(B):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); os_.append(o)
list(pool.map(lambda b: os_[b].initial_run(states[b], ees[b]), range(B)))
gs = []
for tol in tols:
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(tol[0], tol[0], 1e-10, 200)
    g.set_solver_step_rule(tol[1], tol[2])
    g.create_initial_run(states, ees.reshape(B, 12))
    gs.append(g)
for i in range(STEPS):
    t = i * cfg['integrator_dt']
    recs = (host.Trajectory * B)(*[o.trajectory_record(host) for o in os_])
    st_in = np.array([o.states()[1] if i > 0 else states[b] for b, o in enumerate(os_)])
    ee_in = np.array([[[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)] for o in os_]).reshape(B, 12)
    sos = list(pool.map(lambda b: os_[b].rti(st_in[b], t, ee_in[b].reshape(4, 3)), range(B)))
    xo = [o.qp_x() for o in os_]
    for tol, g in zip(tols, gs):
        g.set_warm_start_trajectory(recs)
        g.get_real_time_update(st_in, t, ee_in)
        xr = g.raw_qp_minimiser(); st, err = g.status(); stats = g.stats()
        e = np.array([np.abs(xr[b, :len(xo[b])] - xo[b]).max() / max(1.0, np.abs(xo[b]).max()) if sos[b] <= 1 and st[b] <= 1 else 0.0 for b in range(B)])
        cnt = g.solver_counters()
        print('step %d tol %s (step rule %d / %d, low tried %d failed %d): max %.2e  p99 %.2e  median %.2e  >1e-4: %d  >5e-5: %d  >1e-5: %d   mean iters %.1f  statuses %s  worst inst %d (iters %d, oracle st %d, gap %.1e)' %
              (i, ':'.join('%g' % v for v in tol), cnt['step_rule'], cnt['solves'], cnt['low_tried'], cnt['low_failed'], e.max(), np.percentile(e, 99), np.median(e), (e > 1e-4).sum(), (e > 5e-5).sum(), (e > 1e-5).sum(), stats[:, 4].mean(),
               dict(zip(*np.unique(st, return_counts=True))), e.argmax(), stats[e.argmax(), 4], sos[e.argmax()], stats[e.argmax(), 7]))
