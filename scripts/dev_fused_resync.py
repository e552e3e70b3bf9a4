"""worst relative primal error per step of the re-synchronised protocol through the fused launch (tests/test_gpu_resync.py), for several start_mu"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import test_gpu_resync as T
from oracle_py import load_config
from bench import config_b_instance
cfg = load_config(); B = int(os.environ.get('B', 256)); STEPS = int(os.environ.get('STEPS', 20))
states, ees = zip(*[config_b_instance(cfg, b) for b in range(B)])
states, ees = np.array(states), np.array(ees)
for mu in [float(v) for v in os.environ.get('MUS', '0.1,0').split(',')]:
    r = T.resync_protocol(cfg, states, ees, steps=STEPS, fused=True, x_tol=1.0, start_mu=mu, qp_every=1000, min_alive=0)
    print('start_mu', mu, 'counters', r['counters'], 'worst x %.3e' % r['worst']['x'])
    print('   per step (worst, instance, # > 1e-4):', [(float('%.2e' % a), b, c) for a, b, c in r['x_by_step']])
