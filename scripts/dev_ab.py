"""A/B of library builds on Config B: python scripts/dev_ab.py build_ab/libA.so build_ab/libB.so ...
Each library runs in its own process (SRBM_RTI_LIB): 10 cold-start solves, 5 warm-up steps, 40 timed fused steps; prints
ms per step, IPM iterations, statuses and a checksum of the minimisers (equal checksums = bitwise equal results)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
    import numpy as np
    from srbm_loader import host
    from srbm_loader import workloads as bench
    wl = os.environ.get('AB_WORKLOAD', 'B')
    cfg = host.load_config() if wl == 'B' else host.load_config('a1_config_distr_rejection')
    B = int(os.environ.get('AB_BATCH', 256 if wl == 'B' else 512))
    mod = int(os.environ.get('AB_MOD', 1 << 30))          # AB_MOD=32: the batch is copies of the first 32 instances (same slowest instance at every batch size)
    if wl == 'B':
        states, ees = zip(*[bench.config_b_instance(cfg, b % mod) for b in range(B)])
    else:
        states, ees = zip(*[bench.config_d_instance(cfg, b % mod) for b in range(B)])
    states, ees = np.array(states), np.array(ees).reshape(B, 12)
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states)
    if 'AB_TOL' in os.environ: g.set_solver_tolerances(float(os.environ['AB_TOL']), float(os.environ['AB_TOL']), 1e-10, 200)
    if 'AB_STEP' in os.environ: g.set_solver_step_rule(float(os.environ['AB_STEP']), float(os.environ.get('AB_MU', 0)))
    for _ in range(10): g.create_initial_run(states, ees)
    g.rti_advance(0, 5); g.synchronize()
    if os.environ.get('AB_WINDOWS'):      # five 20-step launches timed one by one (the bench's timed regions)
        w = []
        for k in range(5):
            ta = time.perf_counter(); g.rti_advance(5 + 20 * k, 20); g.synchronize(); w.append((time.perf_counter() - ta) / 20 * 1e3)
        acc = g.status_accumulated()
        print('%-20s windows ms/step %s  median %.3f  not solved %d  err bits %d' % (os.path.basename(os.environ['SRBM_RTI_LIB']), ' '.join('%.3f' % v for v in w), sorted(w)[2],
              int(acc[:, 2].sum()), int(np.bitwise_or.reduce(acc[:, 0]))))
    t0 = time.perf_counter(); g.rti_advance(5, 40); g.synchronize(); t1 = time.perf_counter()
    st = g.status()[0]; x = g.qp_solution()
    print('%-30s tol %s step %s mu %s counters %s  %.3f ms/step  iters %.2f  statuses %s  checksum %.17g' % (os.path.basename(os.environ['SRBM_RTI_LIB']), os.environ.get('AB_TOL', 'default'),
          os.environ.get('AB_STEP', 'default'), os.environ.get('AB_MU', 'default'), g.solver_counters(), (t1 - t0) / 40 * 1e3,
          g.stats()[:, 4].mean(), dict(zip(*np.unique(st, return_counts=True))), float(np.nansum(x * np.cos(np.arange(x.size).reshape(x.shape))))))
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ, SRBM_RTI_LIB=os.path.abspath(lib))
        subprocess.call([sys.executable, os.path.abspath(__file__), '--child'], env=env)
