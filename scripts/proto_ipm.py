"""numpy prototype of the device IPM (non-homogeneous Mehrotra, full space with equality rows) to study robustness."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from srbm_loader import host
from oracle_py import OracleMPC
import bench

def get_qp(b, step=0):
    cfg = host.load_config()
    s0, ee = bench.config_b_instance(cfg, b)
    o = OracleMPC(cfg); o.set_warmstart(s0); o.initial_run(s0, ee)
    state = o.states()[1]; dt = cfg['integrator_dt']
    for i in range(step + 1):
        t = i * dt
        eel = np.array([[o.ee_value(e, 1, c, t) for c in range(3)] for e in range(4)])
        st = o.rti(state, t, eel); state = o.states()[1]
    A, bb, P, q = o.qp_dense()
    sz = o.sizes(); N = cfg['num_nodes']
    nx = (N + 1) * 12; ns = sz['n_force_box'] // 2 if 'n_force_box' in sz else None
    iseq = np.zeros(len(bb), bool)
    iseq[:nx] = True
    iseq[len(bb) - sz['n_td'] - 8:] = True
    return P, q, A, bb, iseq, o.stats(), o.qp_x()

def mehrotra(P, q, A, b, iseq, mode='plain', maxit=100, verbose=True, reg=0.0):
    n = len(q); E = A[iseq]; be = b[iseq]; G = A[~iseq]; h = b[~iseq]; me = len(be); mi = len(h)
    def kkt_solve(w, r1, re, r2):
        # [P E' G'; E 0 0; G 0 -1/w] [dx dy dl] = [r1 re r2]
        K = np.zeros((n + me + mi, n + me + mi))
        K[:n, :n] = P; K[:n, n:n + me] = E.T; K[:n, n + me:] = G.T
        K[n:n + me, :n] = E; K[n + me:, :n] = G
        K[n + me:, n + me:] = -np.diag(1.0 / w)
        K[n:n + me, n:n + me] = -1e-12 * np.eye(me)
        sol = np.linalg.solve(K, np.concatenate([r1, re, r2]))
        return sol[:n], sol[n:n + me], sol[n + me:]
    # start: w = 1
    x, y, l = kkt_solve(np.ones(mi), -q, be, h)
    # here G x - (l') = h  with l' = -(s): s = h - G x = -l'
    s = h - G @ x
    lam = -s
    smin, zmin = s.min(), lam.min()
    if mode == 'plain':
        tgs = max(1.0, 0.1 * np.maximum(s, 0).sum() / mi); tgz = max(1.0, 0.1 * np.maximum(lam, 0).sum() / mi)
        shs = (-smin + tgs) if smin <= 0 else (tgs - smin if smin < tgs else 0.0)
        shz = (-zmin + tgz) if zmin <= 0 else (tgz - zmin if zmin < tgz else 0.0)
        s = s + shs; lam = lam + shz
    elif mode == 'clarabel':
        # Clarabel: if s violates -> shift s by (1 - min) ... (same for z); unit shift
        if smin <= 0: s = s + (1.0 - smin)
        if zmin <= 0: lam = lam + (1.0 - zmin)
    for it in range(maxit):
        rp = G @ x + s - h; re = E @ x - be; rd = P @ x + q + E.T @ y + G.T @ lam
        mu = s @ lam / mi
        pc = 0.5 * x @ P @ x + q @ x; dc = -0.5 * x @ P @ x - be @ y - h @ lam
        gap = abs(pc - dc) / max(1, min(abs(pc), abs(dc)))
        w = lam / s
        # affine
        dx, dy, dl = kkt_solve(w, -rd, -re, -rp + s)
        ds = -rp - G @ dx
        def amax(ds, dl):
            a = 1.0
            m1 = ds < 0; m2 = dl < 0
            if m1.any(): a = min(a, (-s[m1] / ds[m1]).min())
            if m2.any(): a = min(a, (-lam[m2] / dl[m2]).min())
            return a
        aa = amax(ds, dl); sigma = (1 - aa) ** 3
        rcl = (s * lam + ds * dl - sigma * mu) / lam
        c1 = 1 - sigma
        dx, dy, dl2 = kkt_solve(w, -c1 * rd, -c1 * re, -c1 * rp + rcl)
        ds2 = -c1 * rp - G @ dx
        a = 0.99 * amax(ds2, dl2)
        if verbose: print(it, 'mu %.3e sigma %.3e alpha %.3e gap %.3e rp %.2e rd %.2e re %.2e' % (mu, sigma, a, gap, abs(rp).max(), abs(rd).max(), abs(re).max()))
        if gap < 1e-13 and abs(rp).max() < 1e-9 and abs(rd).max() < 1e-8: return x, it
        x = x + a * dx; y = y + a * dy; lam = lam + a * dl2; s = s + a * ds2
        if a < 1e-10: break
    return x, it

def main1():
    b = int(sys.argv[1]); step = int(sys.argv[2]); mode = sys.argv[3] if len(sys.argv) > 3 else 'plain'
    P, q, A, bb, iseq, st, xo = get_qp(b, step)
    print('oracle', st['status'], st['qp_iters'])
    x, it = mehrotra(P, q, A, bb, iseq, mode)
    print('rel err vs oracle', abs(x - xo).max() / max(1, abs(xo).max()))


def equilibrate(P, q, A, b, iters=10, lo=1e-4, hi=1e4):
    P = P.copy(); q = q.copy(); A = A.copy(); b = b.copy()
    n = len(q); m = len(b); D = np.ones(n); E = np.ones(m); c = 1.0
    def lim(v):
        v = np.where(v == 0, 1.0, v); v = np.clip(v, lo, hi); return 1 / np.sqrt(v)
    for it in range(iters):
        dw = np.maximum(abs(P).max(0), abs(A).max(0)); ew = abs(A).max(1)
        dw = lim(dw); ew = lim(ew)
        P = P * dw[:, None] * dw[None, :]; A = A * ew[:, None] * dw[None, :]
        q *= dw; D *= dw; b *= ew; E *= ew
        sc = max(abs(P).max(0).mean(), abs(q).max())
        if sc == 0: sc = 1.0
        sc = min(max(sc, lo), hi)
        P /= sc; q /= sc; c /= sc
    return P, q, A, b, D, E, c


def hsde(P, q, A, b, iseq, equil=True, maxit=100, verbose=True, tol=1e-13, hs=True):
    n = len(q); m = len(b)
    P0, q0, A0, b0 = P, q, A, b
    if equil: P, q, A, b, D, E, c = equilibrate(P, q, A, b)
    else: D = np.ones(n); E = np.ones(m); c = 1.0
    nn = ~iseq; deg = nn.sum()
    def kkt(Wd):
        K = np.zeros((n + m, n + m)); K[:n, :n] = P; K[:n, n:] = A.T; K[n:, :n] = A
        K[n:, n:] = -np.diag(Wd + 1e-13)
        import scipy.linalg as sl
        lu = sl.lu_factor(K)
        return lambda r: sl.lu_solve(lu, r)
    sol = kkt(np.ones(m))(np.concatenate([-q, b]))
    x = sol[:n]; z = sol[n:].copy(); s = -sol[n:].copy()
    def shift(v, primal):
        mn = v[nn].min(); pos = np.maximum(v[nn], 0).sum(); tg = max(1.0, 0.1 * pos / deg)
        sh = (-mn + tg) if mn <= 0 else (tg - mn if mn < tg else 0.0)
        v[nn] += sh
        if primal: v[iseq] = 0
    shift(s, True); shift(z, False)
    tau = 1.0; kap = 1.0
    for it in range(maxit):
        Px = P @ x; xPx = x @ Px
        rx = -Px - q * tau - A.T @ z; rz = A @ x + s - b * tau
        rtau = q @ x + b @ z + kap + xPx / tau
        mu = (s[nn] @ z[nn] + tau * kap) / (deg + 1)
        ti = 1 / tau
        pc = (q @ x * ti + xPx * ti * ti / 2) / c; dc = (-b @ z * ti - xPx * ti * ti / 2) / c
        gap = abs(pc - dc); gap_rel = gap / max(1, min(abs(pc), abs(dc)))
        resp = abs(rz / E).max() * ti; resd = abs(rx / D).max() * ti / c
        Wd = np.where(nn, s / np.where(nn, z, 1), 0.0)
        solve = kkt(Wd)
        s2 = solve(np.concatenate([-q, b])); x2 = s2[:n]; z2 = s2[n:]
        xi = x * ti; Pxi = P @ xi
        tau_den = kap / tau + xi @ Pxi - b @ z2 - (q + 2 * Pxi) @ x2
        def step(dx_, dz_, dt_, dk_, dsc):
            r = solve(np.concatenate([dx_, dsc - dz_])); x1 = r[:n]; z1 = r[n:]
            num = dt_ - dk_ / tau + b @ z1 + (q + 2 * Pxi) @ x1
            ot = num / tau_den
            if not hs: ot = 0.0
            ox = x1 + ot * x2; oz = z1 + ot * z2
            os_ = np.where(nn, -(dsc + Wd * oz), 0.0)
            ok = -(dk_ + kap * ot) / tau
            return ox, oz, os_, ot, ok
        def slen(dz, ds, dt_, dk_):
            a = 1.0
            if dt_ < 0: a = min(a, -tau / dt_)
            if dk_ < 0 and hs: a = min(a, -kap / dk_)
            m1 = nn & (dz < 0); m2 = nn & (ds < 0)
            if m1.any(): a = min(a, (-z[m1] / dz[m1]).min())
            if m2.any(): a = min(a, (-s[m2] / ds[m2]).min())
            return a
        dsc = np.where(nn, s, 0.0)
        dxa, dza, dsa, dta, dka = step(rx, rz, rtau, kap * tau, dsc)
        aa = slen(dza, dsa, dta, dka); sigma = (1 - aa) ** 3
        dsc = np.where(nn, (s * z + dsa * dza - sigma * mu) / np.where(nn, z, 1), 0.0)
        dx, dz, ds, dt_, dk_ = step((1 - sigma) * rx, (1 - sigma) * rz, (1 - sigma) * rtau, kap * tau + dka * dta - sigma * mu, dsc)
        a = 0.99 * slen(dz, ds, dt_, dk_)
        if verbose: print(it, 'mu %.3e sigma %.3e alpha %.3e gap %.3e rp %.2e rd %.2e tau %.3e kap %.3e' % (mu, sigma, a, gap_rel, resp, resd, tau, kap))
        if (gap < tol or gap_rel < tol) and resp < 1e-10 and resd < 1e-10: break
        if a < 1e-4: break
        x = x + a * dx; z = z + a * dz; s = s + a * ds; tau += a * dt_; kap += a * dk_
    return D * x / tau, it


if __name__ == '__main__' and len(sys.argv) > 3 and sys.argv[3].startswith('hsde'):
    P, q, A, bb, iseq, st, xo = get_qp(int(sys.argv[1]), int(sys.argv[2]))
    x, it = hsde(P, q, A, bb, iseq, equil='eq' in sys.argv[3], hs='nohs' not in sys.argv[3])
    print('rel err vs oracle', abs(x - xo).max() / max(1, abs(xo).max()))

if __name__ == '__main__' and not (len(sys.argv) > 3 and (sys.argv[3].startswith('hsde') or sys.argv[3].startswith('cond'))):
    main1()


def condensed(P, q, A, b, iseq, nx=252):
    """device-like reduced problem: x = xp + Z u with exact equalities"""
    import scipy.linalg as sl
    E = A[iseq]; be = b[iseq]
    Ess = E[:nx, :nx]; Esu = E[:nx, nx:]
    Ep = E[nx:, nx:]; bp = be[nx:]
    assert abs(E[nx:, :nx]).max() == 0
    Zu = sl.null_space(Ep)
    up = np.linalg.lstsq(Ep, bp, rcond=None)[0]
    S = -np.linalg.solve(Ess, Esu)
    Z = np.vstack([S @ Zu, Zu])
    xp = np.concatenate([np.linalg.solve(Ess, be[:nx] - Esu @ up), up])
    G = A[~iseq]; h = b[~iseq]
    Gc = G @ Z; hc = h - G @ xp
    Hc = Z.T @ P @ Z; gc = Z.T @ (P @ xp + q)
    return Hc, gc, Gc, hc, Z, xp


def mehrotra_c(H, g, G, h, e=None, c=1.0, maxit=80, verbose=True):
    """device algorithm on the condensed QP with row scaling e / cost scale c used for the starting point only"""
    n = len(g); mi = len(h)
    if e is None: e = np.ones(mi)
    def solve(w, r1, r2):
        M = H + G.T @ (w[:, None] * G)
        du = np.linalg.solve(M, r1 + G.T @ (w * r2))
        return du, w * (G @ du - r2)
    w0 = e * e / c
    u = np.linalg.solve(H + G.T @ (w0[:, None] * G), -g + G.T @ (w0 * h))
    s = h - G @ u; lam = -w0 * s
    sh = e * s; zh = c * lam / e
    def shift(v):
        mn = v.min(); pos = np.maximum(v, 0).sum(); tg = max(1.0, 0.1 * pos / mi)
        return v + ((-mn + tg) if mn <= 0 else (tg - mn if mn < tg else 0.0))
    s = shift(sh) / e; lam = e * shift(zh) / c
    for it in range(maxit):
        rp = G @ u + s - h; rd = H @ u + g + G.T @ lam
        mu = s @ lam / mi
        pc = 0.5 * u @ H @ u + g @ u; dc = -0.5 * u @ H @ u - h @ lam
        gap = abs(pc - dc) / max(1, min(abs(pc), abs(dc)))
        w = lam / s
        du, dl = solve(w, -rd, -rp + s)
        ds = -rp - G @ du
        def amax(ds, dl):
            a = 1.0
            m1 = ds < 0; m2 = dl < 0
            if m1.any(): a = min(a, (-s[m1] / ds[m1]).min())
            if m2.any(): a = min(a, (-lam[m2] / dl[m2]).min())
            return a
        aa = amax(ds, dl); sigma = (1 - aa) ** 3; c1 = 1 - sigma
        rcl = (s * lam + ds * dl - sigma * mu) / lam
        du, dl2 = solve(w, -c1 * rd, -c1 * rp + rcl)
        ds2 = -c1 * rp - G @ du
        a = 0.99 * amax(ds2, dl2)
        hl = h @ lam; gtl = abs(G.T @ lam).max()
        if verbose: print(it, 'mu %.3e sigma %.3e alpha %.3e gap %.3e rp %.2e rd %.2e | h.lam %.3e |Gtlam| %.3e ratio %.2e lam_inf %.2e' % (mu, sigma, a, gap, abs(rp).max(), abs(rd).max(), hl, gtl, gtl / max(1e-300, -hl) / max(1, abs(u).max() + abs(lam).max()), abs(lam).max()))
        if gap < 1e-13 and abs(rp).max() < 1e-9: return u, it
        if a < 1e-8: return u, 999
        u = u + a * du; lam = lam + a * dl2; s = s + a * ds2
    return u, it


def scales(H, g, G, mode):
    rn = np.sqrt((G * G).sum(1)) if '2' in mode else abs(G).max(1)
    e = 1.0 / np.clip(np.where(rn == 0, 1.0, rn), 1e-4, 1e4)
    sc = max(abs(H).max(0).mean(), abs(g).max()); sc = min(max(sc, 1e-4), 1e4)
    if 'noc' in mode: sc = 1.0
    if 'noe' in mode: e = np.ones(len(e))
    return e, 1.0 / (1.0 / sc) if False else (e, sc)[0], sc


if __name__ == '__main__' and len(sys.argv) > 3 and sys.argv[3].startswith('cond'):
    insts = [int(v) for v in sys.argv[1].split(',')]
    for b_ in insts:
        P, q, A, bb, iseq, st, xo = get_qp(b_, int(sys.argv[2]))
        H, g, G, h, Z, xp = condensed(P, q, A, bb, iseq)
        mode = sys.argv[3]
        rn = np.sqrt((G * G).sum(1)) if '2' in mode else abs(G).max(1)
        e = 1.0 / np.clip(np.where(rn == 0, 1.0, rn), 1e-4, 1e4)
        sc = max(abs(H).max(0).mean(), abs(g).max()); sc = min(max(sc, 1e-4), 1e4)
        if 'noc' in mode: sc = 1.0
        if 'noe' in mode: e = np.ones(len(e))
        if '_c' in mode: sc = float(mode.split('_c')[1])
        u, it = mehrotra_c(H, g, G, h, e, 1.0 / sc, verbose=len(insts) == 1)
        x = xp + Z @ u
        print(mode, 'inst', b_, 'oracle status', st['status'], 'iters', st['qp_iters'], 'gap %.1e' % st['gap_rel'], 'proto iters', it, 'rel err vs oracle %.2e' % (abs(x - xo).max() / max(1, abs(xo).max())), 'cost scale', sc, 'row norms', rn[rn > 0].min(), rn.max())
