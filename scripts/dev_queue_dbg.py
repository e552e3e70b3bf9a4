"""step-queue launch against the per-instance launch, launch by launch: first difference (per instance) in solves / iterations / minimiser
    python scripts/dev_queue_dbg.py [batch [launches [steps_per_launch [tol_step start_mu]]]]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    from srbm_loader import host, workloads
    B, L, K, ts, mu = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
    cfg = host.load_config('a1_config_distr_rejection')
    st, ee = zip(*[workloads.config_d_instance(cfg, b % 512) for b in range(B)])
    st, ee = np.array(st), np.array(ee).reshape(B, 12)
    g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(st); g.set_solver_step_rule(ts, mu)
    for _ in range(10): g.create_initial_run(st, ee)
    rec = []
    plan = [(0, 5), (5, 20), (25, 20), (45, 20), (65, 20), (85, 20), (5, 40)] if K == 0 else [(l * K, K) for l in range(L)]
    for first, k in plan:
        g.rti_advance(first, k); g.synchronize()
        acc = g.status_accumulated()
        rec.append(dict(solves=acc[:, 1].tolist(), err=acc[:, 0].tolist(), iters=g.stats()[:, 4].tolist(), status=g.status()[0].tolist(), errlast=g.status()[1].tolist(),
                        x=np.nan_to_num(g.qp_solution()).sum(axis=1).tolist(), flags=g.solve_flags().tolist()))
    print('RESULT ' + json.dumps(rec))
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ts = sys.argv[4] if len(sys.argv) > 4 else '1e-5'
mu = sys.argv[5] if len(sys.argv) > 5 else '0.1'
res = []
for nq in ('0', '1'):
    env = dict(os.environ, SRBM_NO_STEP_QUEUE=nq)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(B), str(L), str(K), ts, mu], env=env, capture_output=True, text=True, timeout=600)
    if p.returncode: print(p.stderr[-3000:]); sys.exit(1)
    res.append(json.loads([l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1][7:]))
for l in range(len(res[0])):
    q, r = res[0][l], res[1][l]
    bad = [b for b in range(B) if q['solves'][b] != r['solves'][b] or q['iters'][b] != r['iters'][b] or q['x'][b] != r['x'][b] or q['status'][b] != r['status'][b]]
    print('launch %d (steps %d..%d): %d instances differ' % (l, l * K, l * K + K - 1, len(bad)), bad[:12])
    for b in bad[:6]:
        print('   inst %d queued: solves %d iters %d status %d err %d/%d flags %d x %.17g | per-instance: solves %d iters %d status %d err %d/%d flags %d x %.17g' % (
            b, q['solves'][b], q['iters'][b], q['status'][b], q['errlast'][b], q['err'][b], q['flags'][b], q['x'][b], r['solves'][b], r['iters'][b], r['status'][b], r['errlast'][b], r['err'][b], r['flags'][b], r['x'][b]))
