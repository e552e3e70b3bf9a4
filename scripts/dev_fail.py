import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
from oracle_py import OracleMPC
import bench
if os.environ.get('SRBM_LIB'): host.LIB_PATH = os.path.join(ROOT, 'bilevel-gait-gen_amd', os.environ['SRBM_LIB'])
NO_ORACLE = bool(os.environ.get('NO_ORACLE'))
cfg = host.load_config()
B = 256
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
g.create_initial_run(states, ees)
bad = set()
for i in range(4):
    g.rti_advance(i, 1); g.synchronize()
    s, e = g.status(); st = g.stats()
    ids = np.nonzero(s > 2)[0]
    print('step', i, 'bad', list(ids), [(int(st[b, 4]), '%.1e' % st[b, 7], '%.1e' % st[b, 5], '%.1e' % st[b, 6]) for b in ids])
    bad |= set(ids.tolist())
dt = cfg['integrator_dt']
for b in ([] if NO_ORACLE else sorted(bad)[:8]):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); o.initial_run(states[b], ees[b].reshape(4, 3))
    state = o.states()[1]; line = []
    for i in range(4):
        t = i * dt
        eel = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
        so = o.rti(state, t, eel); state = o.states()[1]
        line.append((so, o.stats()['qp_iters']))
    print('oracle inst', b, line)
