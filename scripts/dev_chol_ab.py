"""A/B of two library builds on the dense unit-test hook: factor (bitwise comparison) and s_memtime ticks of load + factorisation.
python scripts/dev_chol_ab.py libA.so libB.so   (each library in its own process, SRBM_RTI_LIB)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
    import numpy as np
    from test_gpu_dense import device_cholesky
    out = {}
    for n in [4, 16, 17, 33, 64, 100, 104, 108, 112, 120, 148, 160]:
        rng = np.random.default_rng(n)
        A = rng.standard_normal((n, n + 5)); M = A @ A.T + 0.1 * np.eye(n)
        G = rng.standard_normal((6 * n, n)) * (rng.random((6 * n, n)) < 0.05)
        M2 = 1e-3 * np.eye(n) + (G.T * 10.0 ** rng.uniform(-4, 10, 6 * n)) @ G
        ticks = []
        for k, mat in enumerate((M, M2)):
            (L,), nreg = device_cholesky([mat])
            ticks.append(int(device_cholesky.ticks[0]))
            out['L_%d_%d' % (n, k)] = L
            out['nreg_%d_%d' % (n, k)] = nreg
            out['err_%d_%d' % (n, k)] = np.abs(L @ L.T - mat).max() / np.abs(mat).max()
        out['ticks_%d' % n] = np.array(ticks)
    np.savez(sys.argv[2], **out)
else:
    import numpy as np
    files = []
    for k, lib in enumerate(sys.argv[1:]):
        f = '/tmp/chol_ab_%d.npz' % k
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', f], env=dict(os.environ, SRBM_RTI_LIB=os.path.abspath(lib)))
        files.append(np.load(f))
    a, b = files[0], files[-1]
    for key in sorted(a.files, key=lambda s: (s.split('_')[0], [int(x) for x in s.split('_')[1:]])):
        if key.startswith('ticks'):
            print(key, 'A', a[key].tolist(), 'B', b[key].tolist())
        elif key.startswith('L_'):
            print(key, 'bitwise equal' if np.array_equal(a[key], b[key]) else 'DIFFERENT max %.3e' % np.abs(a[key] - b[key]).max(),
                  'backward err A %.2e B %.2e' % (a['err_' + key[2:]], b['err_' + key[2:]]), 'nreg', a['nreg_' + key[2:]].tolist(), b['nreg_' + key[2:]].tolist())
