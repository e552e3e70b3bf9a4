import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
from oracle_py import OracleMPC, load_config
import bench
cfg = load_config()
B = 8
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-13; g.set_solver_tolerances(tol, tol, 1e-10, 200)
os_ = []
for b in range(B):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); os_.append(o)
for it in range(11):
    g.get_real_time_update(states, 0.0, ees.reshape(B, 12))
    st, err = g.status(); gs = g.stats()
    line = []
    for b in range(B):
        so = os_[b].rti(states[b], 0.0, ees[b])
        n = os_[b].sizes()['n']
        rel = np.abs(g.qp_solution()[b, :n] - os_[b].x()).max() / max(1, np.abs(os_[b].x()).max())
        line.append('%d/%d e%d it%d/%d r%.0e g%.0e %.0e' % (st[b], so, err[b], gs[b, 4], os_[b].stats()['qp_iters'], rel, gs[b, 7], gs[b,5]))
    print(it, ' | '.join(line))
