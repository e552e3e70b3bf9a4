"""per-step picture of the free-running comparison (tests/test_gpu_ownpath.py): where along the path, and in which entries, device and oracle part"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from oracle_py import OracleMPC, load_config
from srbm_loader import host
from bench import config_b_instance
B, steps = int(os.environ.get('B', 32)), int(os.environ.get('STEPS', 60))
cfg = load_config(); dt = cfg['integrator_dt']
states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-15, 1e-15, 1e-10, 200)
if os.environ.get('FAST'): g.enable_fast_termination()
oracles = []
for b in range(B):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); oracles.append(o)
pool = ThreadPoolExecutor(16)
list(pool.map(lambda b: oracles[b].initial_run(states[b], ees[b].reshape(4, 3)), range(B)))
g.create_initial_run(states, ees.reshape(B, 12))
def ostep(b, t):
    o = oracles[b]
    ee = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
    return o.rti(o.states()[1], t, ee)
for i in range(steps):
    t = i * dt
    g.rti_advance(i, 1)
    so = list(pool.map(lambda b: ostep(b, t), range(B)))
    g.synchronize()
    st, e = g.status(); x = g.qp_solution(); sz = g.sizes(); stats = g.stats()
    errs, where = [], []
    for b in range(B):
        o = oracles[b]; n = o.sizes()['n']
        d = np.abs(x[b, :n] - o.x()) / max(1.0, np.abs(o.x()).max())
        errs.append(d.max()); where.append(int(d.argmax()))
    errs = np.array(errs)
    wb = int(errs.argmax())
    os_ = oracles[wb].stats()
    print('step %2d n %d  median %.2e max %.2e (inst %d entry %d of %d)  statuses dev %s oracle %s  alpha dev %.3g oracle %.3g  box dev %s oracle %s  n>1e-4: %d' % (
        i, sz[0, 0], np.median(errs), errs.max(), wb, where[wb], sz[wb, 0], dict(zip(*np.unique(st, return_counts=True))), dict(zip(*np.unique(so, return_counts=True))),
        stats[wb, 0], os_['alpha'], g.knots(wb)['box'], os_['box'], (errs > 1e-4).sum()))
    if errs.max() > 1e-3 and i > 40:
        o = oracles[wb]; n = o.sizes()['n']
        d = np.abs(x[wb, :n] - o.x())
        idx = np.argsort(-d)[:6]
        print('   worst entries', [(int(k), float(x[wb, k]), float(o.x()[k])) for k in idx])
        kd = g.knots(wb)
        for ee in range(4):
            ko = o.knots(ee)
            K = ko['K']
            print('   foot', ee, 'K dev', kd['nk'][ee], 'oracle', K, 'times equal', np.array_equal(kd['times'][ee][:K], ko['times']), 'dev', kd['times'][ee][:kd['nk'][ee]].tolist(), 'kinds', kd['kinds'][ee][:kd['nk'][ee]].tolist())
            print('        oracle times', list(ko['times']))
        print('   sizes dev', sz[wb].tolist(), 'oracle', o.sizes())
