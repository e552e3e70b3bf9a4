import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
if os.environ.get('SRBM_LIB'): host.LIB_PATH = os.path.join(ROOT, 'bilevel-gait-gen_amd', os.environ['SRBM_LIB'])
from test_gpu_dense import device_cholesky
NS = [int(v) for v in os.environ['NS'].split(',')] if os.environ.get('NS') else None
for n in NS or [2, 5, 8, 12, 20, 31, 32, 33, 48, 100, 121, 124, 126, 127, 128, 129, 132, 136, 140, 144, 145, 157, 160]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n + 5)); M = A @ A.T + 0.1 * np.eye(n)
    (L,), nreg = device_cholesky([M])
    E = np.abs(L @ L.T - M)
    bad = np.argwhere(E > 1e-9 * np.abs(M).max())
    print(n, 'ticks', device_cholesky.ticks, 'err %.2e' % E.max(), 'nreg', nreg[0], 'first bad', bad[:3].tolist(), 'n bad', len(bad))
