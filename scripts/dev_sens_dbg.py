# which part of the as-coded sensitivity system is noise: device LU against a dense numpy solve on the same exported data, plus a dump of the
# IPM end point for A/B runs of two library builds (SRBM_RTI_LIB=...; OUT=file.npz)
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import test_gpu_gait as T
from srbm_loader import host
out = {}
for cfgname, nsteps in [('a1_configuration', 3), ('a1_configuration', 8), ('a1_gait_opt_config', 2)]:
    cfg, g, o, state, ee, t = T.run_pair(cfgname, nsteps)
    gait = host.BatchGaitOptimizer(g)
    gait.compute_sensitivity()
    d = gait.sensitivity()
    sz = o.sizes(); n, mi, me = sz['n'], sz['n_ineq'], sz['n_eq']; nx = (cfg['num_nodes'] + 1) * 12
    dz = d[0, :n]
    A, bvec, P, q = g.export_qp(0)
    z, s = g.dual_solution()
    x = g.qp_solution()[0, :n]
    sol, live, lam = T.as_coded_sensitivity(A, P, q, x, z[0], s[0], nx, mi)
    e = np.abs(dz - sol[:n]); k = int(e.argmax())
    st = g.stats() if hasattr(g, 'stats') else None
    print(cfgname, nsteps, 'max err %.3e at %d (dev %.3e ref %.3e)' % (e.max(), k, dz[k], sol[k]), 'n big', int((e > 1e-7).sum()),
          'compl max %.2e' % np.abs(z[0] * s[0]).max(), 'status', g.status()[0], 'qp iters', getattr(g, 'qp_iterations', lambda: None)())
    out['%s_%d_x' % (cfgname, nsteps)] = x; out['%s_%d_z' % (cfgname, nsteps)] = z[0]; out['%s_%d_s' % (cfgname, nsteps)] = s[0]; out['%s_%d_dz' % (cfgname, nsteps)] = dz
    out['%s_%d_ref' % (cfgname, nsteps)] = sol[:n]
if os.environ.get('OUT'): np.savez(os.environ['OUT'], **out)
