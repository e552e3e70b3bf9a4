import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
import bench
cfg = host.load_config()
gold = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_b_rti.json')))
B = 256
states, ees = zip(*[bench.config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees).reshape(B, 12)
g = host.BatchMPC(cfg, B); g.set_state_trajectory_warm_start(states); g.set_solver_tolerances(1e-13, 1e-13, 1e-10, 200)
g.create_initial_run(states, ees)
alive = np.ones(B, bool)
for i in range(gold['steps']):
    g.rti_advance(i, 1); g.synchronize()
    s, e = g.status(); st = g.stats(); x = g.raw_qp_minimiser(); sz = g.sizes()
    so = np.array([r['steps'][i]['status'] for r in gold['instances']])
    worst = 0; nbad = 0
    for b in range(B):
        if not alive[b]: continue
        r = gold['instances'][b]['steps'][i]
        cls_o = 'ok' if so[b] <= 1 else ('unconv' if so[b] == 2 else None) or ('inf' if so[b] in (3, 5) else 'other')
        cls_g = 'ok' if s[b] <= 1 else ('unconv' if s[b] == 2 else None) or ('inf' if s[b] in (3, 5) else 'other')
        if cls_o != cls_g:
            print('step', i, 'inst', b, 'status oracle', so[b], 'gpu', s[b], 'iters', r['iters'], int(st[b, 4])); nbad += 1
        if cls_o != 'ok' or cls_g != 'ok': alive[b] = False; continue
        assert sz[b, 0] == r['n'] and sz[b, 1] == r['m']
        rel = np.abs(x[b, 12:24] - np.array(r['x_head'])).max() / max(1.0, r['x_abs_max'])
        rel = max(rel, abs(x[b, :r['n']].sum() - r['x_sum']) / (r['n'] * max(1.0, r['x_abs_max'])))
        if st[b, 0] != r['alpha']: print('step', i, 'inst', b, 'alpha', st[b, 0], r['alpha'])
        if rel > 1e-4: print('step', i, 'inst', b, 'rel', rel, 'status', so[b], s[b]); nbad += 1
        worst = max(worst, rel)
    print('step', i, 'compared', int(alive.sum()), 'worst rel', worst, 'mismatches', nbad, 'gpu iters mean/max', st[:, 4].mean(), st[:, 4].max())
