import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
from oracle_py import OracleMPC, load_config
import bench
cfg = load_config()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 7
s0, ee = bench.config_b_instance(cfg, b)
g = host.BatchMPC(cfg, 1); g.set_state_trajectory_warm_start(s0); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
o = OracleMPC(cfg); o.set_warmstart(s0)
for it in range(8):
    g.get_real_time_update(s0, 0.0, ee)
    so = o.rti(s0, 0.0, ee)
    n = o.sizes()['n']; m = o.sizes()['m']
    A, bb, P, q = g.export_qp(0); Ao, bo, Po, qo = o.qp_dense()
    xq = g.raw_qp_minimiser()[0, :n]; xo = o.qp_x()
    gs = g.stats()[0]; os_ = o.stats()
    # objective of both minimisers on the ORACLE's QP, and feasibility
    def obj(x): return 0.5 * x @ Po @ x + qo @ x
    print(it, 'st %d/%d' % (g.status()[0][0], so), 'alpha %.4g/%.4g' % (gs[0], os_['alpha']), 'dA %.1e db %.1e' % (np.abs(A - Ao).max(), np.abs(bb - bo).max()),
          'raw rel %.1e' % (np.abs(xq - xo).max() / max(1, np.abs(xo).max())), 'x rel %.1e' % (np.abs(g.qp_solution()[0, :n] - o.x()).max() / max(1, np.abs(o.x()).max())),
          'obj gpu %.10g orc %.10g' % (obj(xq), obj(xo)), 'cost %.8g/%.8g' % (gs[1], os_['cost']), 'eq %.3g/%.3g' % (gs[2], os_['eq_violation']))
