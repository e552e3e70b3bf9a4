"""GPU unit tests of the dense fp64 building blocks of the IPM kernel (bilevel-gait-gen_amd/csrc/srbm_dense.hiph) against
numpy in fp64: the MFMA Cholesky of the register-resident normal matrix.  Tolerance: backward error |L L' - M| <=
1e-13 |M| (fp64 Cholesky is backward stable; the entries of L themselves are compared to 1e-9 relative on
well-conditioned inputs only)."""
import ctypes as C
import numpy as np
import pytest

from srbm_loader import host

pytestmark = pytest.mark.gpu


def pack(M):
    n = M.shape[0]
    return np.concatenate([M[i, :i + 1] for i in range(n)])


def unpack(p, n):
    L = np.zeros((n, n))
    k = 0
    for i in range(n):
        L[i, :i + 1] = p[k:k + i + 1]; k += i + 1
    return L


def device_cholesky(mats):
    lib = host.lib()
    n = mats[0].shape[0]
    inp = np.ascontiguousarray(np.stack([pack(M) for M in mats]))
    out = np.zeros_like(inp); nreg = np.zeros(len(mats), np.int32)
    rc = lib.srbm_debug_cholesky(n, len(mats), inp.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)),
                                 nreg.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0, lib.srbm_last_error().decode()
    device_cholesky.ticks = (nreg >> 8) * 16      # diagnostic: s_memtime ticks of load + factorisation
    return [unpack(o, n) for o in out], nreg & 0xff


@pytest.mark.parametrize('n', [1, 3, 4, 15, 16, 17, 63, 64, 118, 120, 157, 160])
def test_mfma_cholesky_matches_numpy(n):
    rng = np.random.default_rng(n)
    mats = []
    for _ in range(3):
        A = rng.standard_normal((n, n + 5))
        mats.append(A @ A.T + 0.1 * np.eye(n))
    Ls, nreg = device_cholesky(mats)
    assert np.all(nreg == 0)
    for M, L in zip(mats, Ls):
        assert np.abs(np.triu(L, 1)).max() == 0
        assert np.abs(L @ L.T - M).max() <= 1e-13 * np.abs(M).max()
        Lr = np.linalg.cholesky(M)
        assert np.abs(L - Lr).max() <= 1e-9 * np.abs(Lr).max()


def test_mfma_cholesky_barrier_weighted_matrix():
    """the shape the IPM produces: H with curvature 1e-3 plus G' W G with weights spread over 14 decades"""
    rng = np.random.default_rng(7)
    n, m = 120, 752
    G = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.05)
    w = 10.0 ** rng.uniform(-4, 10, m)
    M = 1e-3 * np.eye(n) + G.T @ (w[:, None] * G)
    (L,), nreg = device_cholesky([M])
    assert nreg[0] == 0
    assert np.abs(L @ L.T - M).max() <= 1e-13 * np.abs(M).max()


def test_mfma_cholesky_reports_indefinite_input():
    n = 40
    M = np.eye(n); M[17, 17] = -1.0
    (L,), nreg = device_cholesky([M])
    assert nreg[0] == 1


def device_solve(mats, rhs):
    lib = host.lib()
    n = mats[0].shape[0]
    inp = np.ascontiguousarray(np.stack([pack(M) for M in mats])); r = np.ascontiguousarray(np.stack(rhs))
    x = np.zeros_like(r); X = np.zeros_like(inp); ticks = np.zeros(2 * len(mats), np.int32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib.srbm_debug_solve(n, len(mats), dp(inp), dp(r), dp(x), dp(X), ticks.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0, lib.srbm_last_error().decode()
    device_solve.ticks = ticks.reshape(-1, 2)
    return x, [unpack(v, n) for v in X]


@pytest.mark.parametrize('n', [1, 5, 16, 17, 33, 64, 100, 120, 128, 129, 144, 157, 160])
def test_explicit_factor_inverse_and_solve(n):
    """X = L^-1 by MFMA tile products, then x = X'(X b): against numpy"""
    rng = np.random.default_rng(100 + n)
    mats, rhs = [], []
    for _ in range(2):
        A = rng.standard_normal((n, n + 3))
        mats.append(A @ A.T + 0.5 * np.eye(n)); rhs.append(rng.standard_normal(n))
    x, Xs = device_solve(mats, rhs)
    for M, b, xv, X in zip(mats, rhs, x, Xs):
        L = np.linalg.cholesky(M)
        assert np.abs(X @ L - np.eye(n)).max() <= 1e-10 * np.linalg.cond(L)
        xr = np.linalg.solve(M, b)
        assert np.abs(xv - xr).max() <= 1e-9 * max(1.0, np.abs(xr).max()) * np.linalg.cond(M) ** 0.5


@pytest.mark.parametrize('n,nfix', [(20, 3), (120, 8), (120, 12), (120, 16), (137, 9), (148, 12), (160, 8), (160, 1), (33, 30)])
def test_dense_phase_on_a_subset_of_the_columns(n, nfix):
    """the IPM's dense phase leaves the pinned / substituted position variables (identity rows of the normal matrix) out: tiles loaded
    through an increasing column map, factor and inverse of the free block, solves that gather and scatter through the map.  Against a
    numpy solve of the FULL system with the identity rows in place; entries outside the map keep the right-hand side's value."""
    lib = host.lib()
    rng = np.random.default_rng(7000 + 31 * n + nfix)
    fixed = np.sort(rng.choice(n, nfix, replace=False))
    free = np.setdiff1d(np.arange(n), fixed).astype(np.int32)
    mats, rhs = [], []
    for _ in range(3):
        G = rng.standard_normal((2 * n, n)); w = 10.0 ** rng.uniform(-3, 8, 2 * n)
        M = 1e-2 * np.eye(n) + G.T @ (w[:, None] * G)
        M[fixed, :] = 0; M[:, fixed] = 0; M[fixed, fixed] = 1.0
        b = rng.standard_normal(n); b[fixed] = rng.standard_normal(nfix) * (rng.random(nfix) < 0.5)      # the IPM's are zero there; any value must survive
        mats.append(M); rhs.append(b)
    inp = np.ascontiguousarray(np.stack([pack(M) for M in mats])); r = np.ascontiguousarray(np.stack(rhs))
    x = np.zeros_like(r); nreg = np.zeros(len(mats), np.int32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    rc = lib.srbm_debug_solve_mapped(n, len(free), ip(free), len(mats), dp(inp), dp(r), dp(x), ip(nreg))
    assert rc == 0, lib.srbm_last_error().decode()
    assert np.all(nreg == 0)
    for M, b, xv in zip(mats, rhs, x):
        xr = np.linalg.solve(M, b)
        assert np.array_equal(xv[fixed], b[fixed])
        Mf = M[np.ix_(free, free)]
        assert np.abs(xv - xr).max() <= 1e-9 * max(1.0, np.abs(xr).max()) * np.linalg.cond(Mf) ** 0.5
    # a map that is not increasing is refused
    bad = free.copy(); bad[[0, 1]] = bad[[1, 0]]
    if len(bad) > 1: assert lib.srbm_debug_solve_mapped(n, len(bad), ip(bad), 1, dp(inp), dp(r), dp(x), ip(nreg)) != 0
