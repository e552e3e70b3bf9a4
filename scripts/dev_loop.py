"""Developer script: cold start + open-loop RTI parity over many steps, and a first timing."""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
spec = importlib.util.spec_from_file_location('srbm_host', os.path.join(ROOT, 'bilevel-gait-gen_amd', 'host.py'))
host = importlib.util.module_from_spec(spec); spec.loader.exec_module(host)
from oracle_py import OracleMPC, load_config
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'a1_configuration'
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = load_config(cfgname)
s0 = np.array(cfg['srb_init'], float)
ee = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)
g = host.BatchMPC(cfg, 2); g.set_state_trajectory_warm_start(s0); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
o = OracleMPC(cfg); o.set_warmstart(s0)
g.create_initial_run(s0, ee); o.initial_run(s0, ee)
dt = cfg['integrator_dt']
state = s0
for i in range(nsteps):
    t = i * dt
    sto = o.states()
    eel = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
    state = sto[1] if i > 0 else s0
    g.get_real_time_update(state, t, eel); so = o.rti(state, t, eel)
    sz = g.sizes()[0]; osz = o.sizes(); n = osz['n']
    st, err = g.status(); gs = g.stats()[0]; os_ = o.stats()
    x = g.qp_solution()[0, :n]; xo = o.x()
    kg = g.knots(0)
    kt = max(np.abs(kg['times'][e, :kg['nk'][e]] - o.knots(e)['times']).max() if kg['nk'][e] == o.knots(e)['K'] else 9e9 for e in range(4))
    print('%2d n %d/%d m %d/%d st %d/%d err %d it %d/%d alpha %.4g/%.4g relx %.2e cost %.8g/%.8g knots %.1e box %s' % (
        i, sz[0], osz['n'], sz[1], osz['m'], st[0], so, err[0], gs[4], os_['qp_iters'], gs[0], os_['alpha'],
        np.abs(x - xo).max() / max(1, np.abs(xo).max()), gs[1], os_['cost'], kt, kg['box']))
# timing: device-resident protocol
for B in (256, 1024):
    gb = host.BatchMPC(cfg, B); gb.set_state_trajectory_warm_start(s0); gb.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
    gb.create_initial_run(s0, ee)
    gb.rti_advance(0, 3); gb.synchronize()
    t0 = time.time(); K = 20
    gb.rti_advance(3, K); gb.synchronize()
    el = time.time() - t0
    print('batch %d: %d RTI steps in %.4f s -> %.1f it/s  (%.3f ms/step) bytes/inst %d' % (B, K, el, B * K / el, 1e3 * el / K, gb.L.srbm_bytes_per_instance()))
    gb.close()
