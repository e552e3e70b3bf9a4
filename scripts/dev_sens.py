import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from test_gpu_gait import run_pair, host
cfg, g, o, state, ee, t = run_pair(sys.argv[1] if len(sys.argv) > 1 else 'a1_configuration', int(sys.argv[2]) if len(sys.argv) > 2 else 3)
print('status', o.stats()['status'], g.status()[0][0], 'alpha', o.stats()['alpha'], g.stats()[0, 0])
assert o.gait_gradient() is not None
do = o.gait_d()
gait = host.BatchGaitOptimizer(g); gait.compute_sensitivity(); d = gait.sensitivity()
sz = o.sizes(); n, mi, me = sz['n'], sz['n_ineq'], sz['n_eq']; nx = (cfg['num_nodes'] + 1) * 12
dz, dl, dn = d[0, :n], d[0, n:n + mi], d[0, n + mi:n + mi + me]
for name, a, bb in [('dz states', dz[:nx], do[:nx]), ('dz u', dz[nx:], do[nx:n]), ('dlam', dl, do[n:n + mi]), ('dnu', dn, do[n + mi:])]:
    diff = np.abs(a - bb); i = int(np.argmax(diff))
    print(name, 'max|oracle| %.3e max|gpu| %.3e maxdiff %.3e at %d (gpu %.6e oracle %.6e)' % (np.abs(bb).max(), np.abs(a).max(), diff.max(), i, a[i], bb[i]))
print('dz u gpu', np.array2string(dz[nx:], precision=3, max_line_width=200))
print('dz u ora', np.array2string(do[nx:n], precision=3, max_line_width=200))
z, s = g.dual_solution()
zo, so = o.z(), o.s()
lam_g, s_g = z[0, nx:nx + mi], s[0, nx:nx + mi]
lam_o, s_o = zo[nx:nx + mi], so[nx:nx + mi]
wg, wo = lam_g / np.maximum(s_g, 1e-300), lam_o / np.maximum(s_o, 1e-300)
mid = (wo > 1e-3) & (wo < 1e6)
print('rows with mid-range weights (oracle):', np.nonzero(mid)[0][:20], wo[mid][:20], wg[mid][:20])
mid = (wg > 1e-3) & (wg < 1e6)
print('rows with mid-range weights (gpu):', np.nonzero(mid)[0][:20], wg[mid][:20], wo[mid][:20])

# ---- full-space as-coded solve in numpy from the GPU's own QP data ----
A, bvec, Pm, q = g.export_qp(0)
xs = g.qp_solution()[0, :n]
m = A.shape[0]
ineq = np.arange(nx, nx + mi); eq = np.concatenate([np.arange(nx), np.arange(nx + mi, m)])
G, Ae = A[ineq], A[eq]
lam, sl = z[0, ineq], s[0, ineq]
Kf = np.zeros((n + mi + len(eq),) * 2)
Kf[:n, :n] = Pm; Kf[:n, n:n + mi] = G.T * lam[None, :]; Kf[:n, n + mi:] = Ae.T
Kf[n:n + mi, :n] = G; Kf[n:n + mi, n:n + mi] = np.diag(sl)
Kf[n + mi:, :n] = Ae
rhs = np.zeros(Kf.shape[0]); rhs[:n] = -(Pm @ xs + q)
live = np.abs(G).max(axis=1) > 0
keep = np.concatenate([np.ones(n, bool), live, np.ones(len(eq), bool)])
sol = np.zeros(Kf.shape[0])
sol[keep] = np.linalg.solve(Kf[np.ix_(keep, keep)], rhs[keep])
print('numpy(full, GPU data): |dz| max %.3e ; vs gpu dz diff %.3e ; dnu diff %.3e' % (np.abs(sol[:n]).max(), np.abs(sol[:n] - dz).max(), np.abs(sol[n + mi:] - dn).max()))
print('stationarity of GPU solution |P x + q + G lam + A nu|: %.3e' % np.abs(Pm @ xs + q + G.T @ lam + Ae.T @ z[0, eq]).max())
