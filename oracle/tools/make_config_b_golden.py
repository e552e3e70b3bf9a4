"""Golden vectors for Config B (BASELINE.json configs[1]): the oracle run over the first RTI steps of the loop of
/root/reference/test/gait_opt_playground.cpp:113-126 (state := node 1 of the previous trajectory, time = i*dt) for all
256 seeded instances.  Stores per step the solver status, the QP sizes, the Armijo step (config_b_rti.json) and the FULL QP
minimiser of every instance and step (config_b_rti_x.npz: x[256][steps][412], zero padded beyond n) so that the GPU suite can
check status classes (solved / primal infeasible) and solutions entry-wise without re-running the oracle.
Usage: python oracle/tools/make_config_b_golden.py   (writes tests/golden/config_b_rti.json and config_b_rti_x.npz)"""
import json, os, sys
from multiprocessing import Pool
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
STEPS = 4


def run(b):
    from srbm_loader import host
    from oracle_py import OracleMPC
    import bench
    cfg = host.load_config()
    s0, ee = bench.config_b_instance(cfg, b)
    o = OracleMPC(cfg); o.set_warmstart(s0); o.initial_run(s0, ee)
    out = dict(instance=b, init_status=o.stats()['status'], steps=[])
    dt = cfg['integrator_dt']
    for i in range(STEPS):
        t = i * dt
        state = o.states()[1]
        eel = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
        o.rti(state, t, eel)
        st = o.stats(); sz = o.sizes(); x = o.qp_x()
        out['steps'].append(dict(status=st['status'], iters=st['qp_iters'], n=sz['n'], m=sz['m'], alpha=st['alpha'],
                                 cost=st['cost'], x_sum=float(np.sum(x)), x_abs_max=float(np.abs(x).max()),
                                 x_head=[float(v) for v in x[12:24]]))
        out.setdefault('_x', []).append(np.pad(x, (0, 412 - len(x))))
    return out


if __name__ == '__main__':
    with Pool(8) as p:
        res = p.map(run, range(256), chunksize=4)
    X = np.array([r.pop('_x') for r in res])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'config_b_rti_x.npz'), x=X)
    with open(os.path.join(ROOT, 'tests', 'golden', 'config_b_rti.json'), 'w') as f:
        json.dump(dict(steps=STEPS, instances=res), f)
    from collections import Counter
    for i in range(STEPS):
        print('step', i, Counter(r['steps'][i]['status'] for r in res))
