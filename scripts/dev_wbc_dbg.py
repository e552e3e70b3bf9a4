import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_gpu_wbc as t
from srbm_loader import host
B = 256
cfg, q, v, q_des, v_des, rng = t.make(B, seed=9)
contact = np.array([t.CONTACTS[b % 3] for b in range(B)], np.int32)
fdes = np.zeros((B, 12))
for b in range(B):
    nc = contact[b].sum(); fdes[b, :3 * nc] = np.tile([0, 0, cfg['mass'] * 9.81 / nc], nc)
g = host.BatchMPC(cfg, B)
ctl, sol, st, iters = g.qp_control(q, v, contact, q_des, v_des, fdes)
bad = np.nonzero(st != 0)[0]
print('bad', bad, 'status', st[bad], 'iters', iters[bad], 'contacts', contact[bad].tolist())
print('iters hist', np.bincount(iters))
