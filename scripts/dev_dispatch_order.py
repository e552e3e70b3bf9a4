"""Would a dispatch order by expected length shorten the launches of a batch larger than the CU count?  Config D (512 instances, 256 CUs: two rounds),
the bench protocol (10 cold starts, 5 warm-up steps, 5 windows of 20 fused steps); per window the factorisations each instance executed.  The launch is
simulated as a greedy dispatch on 256 machines (the next workgroup goes to the first CU that frees up), job length = factorisations of the window:
   index order (what the hardware does today)  |  longest first by the PREVIOUS window's counts (what a permutation could do)  |  longest first by the
   window's own counts (the bound no predictor beats)  |  the mean load (perfect balance)
    python scripts/dev_dispatch_order.py [tol_step start_mu]"""
import ctypes as C, heapq, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srbm_loader import host, workloads
ts = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
cfg = host.load_config('a1_config_distr_rejection')
B, CUS = 512, 256
st, ee = zip(*[workloads.config_d_instance(cfg, b) for b in range(B)])
st, ee = np.array(st), np.array(ee).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_step_rule(ts, mu)
for _ in range(10): g.create_initial_run(st, ee)
def totals():
    out = np.zeros(B)
    g._chk(g.L.srbm_debug_get_instance_iters(g.h, out.ctypes.data_as(C.POINTER(C.c_double))))
    return out
def makespan(jobs, order):
    free = [0.0] * CUS
    heapq.heapify(free)
    end = 0.0
    for j in order:
        t = heapq.heappop(free) + jobs[j]
        end = max(end, t); heapq.heappush(free, t)
    return end
t0 = totals(); g.rti_advance(0, 5); g.synchronize(); t1 = totals()
prev = t1 - t0
print('mode (%g, %g); factorisations per instance and window' % (ts, mu))
for w in range(5):
    g.rti_advance(5 + 20 * w, 20); g.synchronize()
    t2 = totals(); jobs = t2 - t1; t1 = t2
    idx = makespan(jobs, range(B)); lp = makespan(jobs, np.argsort(-prev)); lo = makespan(jobs, np.argsort(-jobs))
    print('window %d: mean %.0f max %.0f | makespan: index order %.0f  longest-first by the previous window %.0f (%.3f x)  by its own counts %.0f (%.3f x)  perfect balance %.0f | correlation with the previous window %.2f' % (
        w, jobs.mean(), jobs.max(), idx, lp, lp / idx, lo, lo / idx, jobs.sum() / CUS, np.corrcoef(prev, jobs)[0, 1]))
    prev = jobs
