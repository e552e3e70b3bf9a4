"""Which spline variables carry the QP step of an RTI iteration: u - u_prev of one Config-B instance at consecutive steps, with the column descriptors.
    python scripts/dev_step_where.py [instance [first_step [count]]]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srbm_loader import host, workloads
b = int(sys.argv[1]) if len(sys.argv) > 1 else 11
first = int(sys.argv[2]) if len(sys.argv) > 2 else 56
count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = host.load_config()
s0, ee = workloads.config_b_instance(cfg, b)
g = host.BatchMPC(cfg, 1); g.set_state_trajectory_warm_start(s0); g.set_solver_step_rule(0.0, 0.1)
g.create_initial_run(s0, ee)
NU = g.NUMAX
for i in range(first + count):
    g.rti_advance(i, 1); g.synchronize()
    if i < first: continue
    up = np.zeros(NU); u = np.zeros(NU); cols = np.zeros((4, NU), np.int32); fix = np.zeros(NU, np.int32)
    assert g.L.srbm_debug_get_spline_step(g.h, 0, up.ctypes.data_as(C.POINTER(C.c_double)), u.ctypes.data_as(C.POINTER(C.c_double)),
                                          cols.ctypes.data_as(C.POINTER(C.c_int)), fix.ctypes.data_as(C.POINTER(C.c_int))) == 0
    nu = int(g.sizes()[0, 0]) - 252
    d = (u - up)[:nu]
    order = np.argsort(-np.abs(d))[:10]
    k = g.knots(0)
    print('step %d  t0 %.2f  iterations %d  |p|(stats) %.2f  |du|_2 %.2f  nu %d' % (i, i * cfg['integrator_dt'], g.stats()[0, 4], g.stats()[0, 3], np.linalg.norm(d), nu))
    for j in order:
        print('   var %3d  foot %d type %s coord %d local %2d  fix %d   u_prev %10.4f -> u %10.4f' % (j, cols[0, j], 'F' if cols[1, j] == 0 else 'P', cols[2, j], cols[3, j], fix[j], up[j], u[j]))
    for e in range(4):
        print('   foot %d knots' % e, ' '.join('%.3f%s' % (t, 'LTFM'[kd]) for t, kd in zip(k['times'][e][:k['nk'][e]], k['kinds'][e][:k['nk'][e]])))
