"""Diagnostic: fused vs one-launch-per-phase, step by step on Config B; prints the first step where they differ and what differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from srbm_loader import host
import bench
cfg = host.load_config()
B = int(os.environ.get('FVU_B', 32))
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
gs = []
for k in range(2):
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.create_initial_run(states, ees); gs.append(g)
print('after cold start equal:', np.array_equal(gs[0].qp_solution(), gs[1].qp_solution()))
for i in range(7):
    gs[0].rti_advance(i, 1); gs[0].synchronize()
    gs[1].rti_advance_unfused(i, 1); gs[1].synchronize()
    xa, xb = gs[0].qp_solution(), gs[1].qp_solution()
    sa, sb = gs[0].stats(), gs[1].stats()
    qa, qb = gs[0].export_qp(0), gs[1].export_qp(0)
    same_qp = [bool(np.array_equal(a, b)) for a, b in zip(qa, qb)]
    d = np.abs(xa - xb).max(axis=1)
    print('step %2d  max|dx| %.3e  worst inst %d  iters fused %s unfused %s  status f %s u %s  qp(inst0) equal %s' % (
        i, d.max(), d.argmax(), sa[:4, 4], sb[:4, 4], gs[0].status()[0][:6], gs[1].status()[0][:6], same_qp))
    print('         sizes f', gs[0].sizes()[0], 'u', gs[1].sizes()[0], 'err f', gs[0].status()[1][:4], 'stats f', np.array2string(sa[0], precision=3))
    tr = np.abs(gs[0].trajectory_states() - gs[1].trajectory_states()).max()
    print('         traj diff %.3e' % tr)
