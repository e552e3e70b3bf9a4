"""GPU scratch driver: where does the exported QP differ from the oracle's after a resynchronised step?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle_py import OracleMPC, load_config
from srbm_loader import host
from bench import config_b_instance
cfg = load_config()
B = 8
states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
os_ = []
for b in range(B):
    o = OracleMPC(cfg); o.set_warmstart(states[b]); o.initial_run(states[b], ees[b]); os_.append(o)
g.create_initial_run(states, ees.reshape(B, 12))
for i in range(3):
    t = i * cfg['integrator_dt']
    recs = (host.Trajectory * B)(*[o.trajectory_record(host) for o in os_])
    g.set_warm_start_trajectory(recs)
    st_in = np.array([o.states()[1] if i > 0 else states[b] for b, o in enumerate(os_)])
    ee_in = np.array([[[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)] for o in os_]).reshape(B, 12)
    g.get_real_time_update(st_in, t, ee_in)
    for b, o in enumerate(os_):
        o.rti(st_in[b], t, ee_in[b].reshape(4, 3))
        A, bv, P, q = g.export_qp(b); Ao, bo, Po, qo = o.qp_dense()
        d = (A != 0) != (Ao != 0)
        if d.any():
            rr, cc = np.nonzero(d)
            print('step', i, 'inst', b, 'pattern diffs', len(rr))
            for r, c in list(zip(rr, cc))[:10]:
                print('   row', r, 'col', c, 'gpu', A[r, c], 'oracle', Ao[r, c])
        print('step', i, 'inst', b, 'max|dA|', np.abs(A - Ao).max(), 'db', np.abs(bv - bo).max(), 'dq', np.abs(q - qo).max(), 'n', A.shape)
