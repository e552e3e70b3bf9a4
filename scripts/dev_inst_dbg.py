"""GPU scratch driver: one Config-B instance, device re-synchronised to the oracle before every step; prints the IPM trace of
the diagnostic build (make ../libsrbm_rti_prof.so; SRBM_RTI_LIB=.../libsrbm_rti_prof.so) at a given step."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ctypes as C
from oracle_py import OracleMPC, load_config
from srbm_loader import host
from bench import config_b_instance
inst, step = int(sys.argv[1]), int(sys.argv[2])
cfg = load_config()
s0, ee = config_b_instance(cfg, inst)
o = OracleMPC(cfg); o.set_warmstart(s0); o.initial_run(s0, ee)
g = host.BatchMPC(cfg, 1); g.set_state_trajectory_warm_start(s0)
if len(sys.argv) > 3: g.set_solver_tolerances(float(sys.argv[3]), float(sys.argv[3]), 1e-10, 200)
g.create_initial_run(s0, ee)
for i in range(step + 1):
    t = i * cfg['integrator_dt']
    g.set_warm_start_trajectory([o.trajectory_record(host)])
    st_in = o.states()[1] if i > 0 else s0
    ee_in = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
    g.get_real_time_update(st_in, t, ee_in)
    so = o.rti(st_in, t, ee_in)
    st, err = g.status(); stats = g.stats()[0]
    n = o.sizes()['n']
    print('step', i, 'oracle status', so, 'iters', o.stats()['qp_iters'], '| gpu status', st[0], 'err', err[0], 'iters', stats[4], 'res_p %.1e res_d %.1e gap %.1e' % (stats[5], stats[6], stats[7]),
          'relerr', np.abs(g.raw_qp_minimiser()[0, :n] - o.qp_x()).max() / max(1, np.abs(o.qp_x()).max()), 'n', n, 'ntd', o.sizes()['n_td'])
if 'prof' in host.LIB_PATH:
    tr = np.zeros(384)
    g.L.srbm_debug_get_trace(g.h, 0, tr.ctypes.data_as(C.POINTER(C.c_double)))
    tr = tr[:256].reshape(32, 8)
    print('it   mu        sigma     alpha     gap_rel   res_p     res_d     refine  err')
    for k in range(32):
        if tr[k, 0] == 0 and k > 0: break
        print('%2d ' % k + ' '.join('%9.2e' % v for v in tr[k]))
