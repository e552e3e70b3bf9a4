import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
if os.environ.get('SRBM_LIB'): host.LIB_PATH = os.path.join(ROOT, 'bilevel-gait-gen_amd', os.environ['SRBM_LIB'])
from test_gpu_dense import device_solve
for n in [32, 64, 108, 120, 128, 160]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n + 5)); M = A @ A.T + 0.1 * np.eye(n)
    x, X = device_solve([M] * 4, [rng.standard_normal(n)] * 4)
    print(n, 'trtri ticks', device_solve.ticks[:, 0], 'solve ticks', device_solve.ticks[:, 1])
