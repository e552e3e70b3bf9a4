"""Cycles per phase of the whole-body QP solve (diagnostic build: make -C bilevel-gait-gen_amd/csrc ../libsrbm_rti_prof.so; run with
SRBM_RTI_LIB=bilevel-gait-gen_amd/libsrbm_rti_prof.so).  The stamps come back in the last row of A of the QP dump (free with < 4 feet in contact)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from srbm_loader import host
import test_gpu_wbc as T

B = 256
cfg, q, v, q_des, v_des, rng = T.make(B, seed=9)
contact = np.array([[[1, 0, 0, 1], [0, 1, 1, 0], [1, 1, 1, 0]][b % 3] for b in range(B)], np.int32)
fdes = np.zeros((B, 12))
for b in range(B):
    nc = contact[b].sum(); fdes[b, :3 * nc] = np.tile([0, 0, cfg['mass'] * 9.81 / nc], nc)
g = host.BatchMPC(cfg, B)
ctl, sol, st, iters, qp = g.qp_control(q, v, contact, q_des, v_des, fdes, dump=True)
pr = qp['A'][:, 49, :9]
names = ['start point', 'residual passes', 'reductions + test', 'assembly', 'elimination', 'substitutions', 'rest of the passes']
print('iterations min %d median %d max %d; statuses %s' % (iters.min(), np.median(iters), iters.max(), np.unique(st, return_counts=True)))
tot = pr[:, 7]
print('whole solve (cycles of s_memtime): mean %.0f max %.0f; assembly of the QP before it (dynamics by 20 recursive Newton-Euler passes, rows, split): mean %.0f max %.0f' % (tot.mean(), tot.max(), pr[:, 8].mean(), pr[:, 8].max()))
for k, nme in enumerate(names):
    per_it = pr[:, k] / np.maximum(1, iters) if k else pr[:, k]
    print('%-22s mean %9.0f  share %5.1f %%   per iteration %8.0f' % (nme, pr[:, k].mean(), 100 * pr[:, k].sum() / pr[:, :7].sum(), per_it.mean()))
print('stamped %.0f of %.0f' % (pr[:, :7].sum(1).mean(), tot.mean()))
