import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
import bench
cfg = host.load_config()
B = 256
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
g.create_initial_run(states, ees)
for i in range(40):
    t0 = time.time(); g.rti_advance(i, 1); g.synchronize(); el = time.time() - t0
    st = g.stats(); s, e = g.status()
    it = st[:, 4]
    print('%2d %.1f ms iters mean %.1f max %d (inst %d) >40: %d  status %s gap of worst %.1e res %.1e %.1e' % (i, el * 1e3, it.mean(), it.max(), it.argmax(), (it > 40).sum(), dict(zip(*np.unique(s, return_counts=True))), st[it.argmax(), 7], st[it.argmax(), 5], st[it.argmax(), 6]))
