"""Developer lab (CPU only): multilevel variants of the device preconditioner on a numpy mirror.
Level l: SIMPLE block-Jacobi sweep + aggregation coarse space; the coarse saddle-point problem is either
inverted densely (the shipped two-level method, aggregates grown until kc <= coarse_max) or handed to the
same construction one level down (aggregates stay small).
python tools/precond_lab3.py N nu [variant ...]"""
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from optconpy_amd import _lib, problems as pb  # noqa: E402
from check_precond import lists  # noqa: E402
from precond_lab import gmres_right  # noqa: E402


def agg(graph, size):
    g = graph.tocsr()
    g.sort_indices()
    return _lib.host_aggregate(g, size)


def bj_matrix(Mat, blk, nb):
    order, ptr = lists(blk, nb)
    Mp = Mat[order][:, order].tocsr()
    invs = []
    for b in range(nb):
        Bm = Mp[ptr[b]:ptr[b + 1], ptr[b]:ptr[b + 1]].toarray()
        invs.append(np.linalg.inv(Bm))
    n = len(order)
    Pm = sps.csr_matrix((np.ones(n), (np.arange(n), order)), shape=(n, n))
    return (Pm.T @ sps.block_diag(invs) @ Pm).tocsr()


def level(Ap, J, gv, depth, bs=32, av=16, ap=24, coarse_max=4096, gamma=1, av_c=None, ap_c=None, stats=None,
          omega=1.0, post_coarse=False, inner=0):
    """Returns P(r) ~ S^-1 r for S = [[Ap, J^T], [J, 0]]."""
    nv, npp = Ap.shape[0], J.shape[0]
    S = sps.bmat([[Ap, J.T], [J, None]], format="csr")
    pat = abs(Ap).tocsr()
    blk, nb = agg(pat, bs)
    pp = (abs(J) @ abs(J).T).tocsr()
    pblk, npb = agg(pp, bs)
    AinvM = bj_matrix(Ap, blk, nb)
    Sh = (J @ AinvM @ J.T).tocsr()
    SinvM = bj_matrix(Sh, pblk, npb)

    def P1(r):
        zv = AinvM @ r[:nv]
        zp = SinvM @ (J @ zv - r[nv:])
        zv = zv - AinvM @ (J.T @ zp)
        return np.vstack([zv, zp])

    a_v, a_p = av, ap
    while True:
        if a_p == bs:
            a_p += a_p // 2
        va, kv = agg(gv, a_v)
        pa, kp = agg(pp, a_p)
        if depth > 0 or kv + kp <= coarse_max:
            break
        a_v *= 2
        a_p *= 2
    Yv = sps.csr_matrix((np.ones(nv), (np.arange(nv), va)), shape=(nv, kv))
    Yp = sps.csr_matrix((np.ones(npp), (np.arange(npp), pa)), shape=(npp, kp))
    Y = sps.block_diag([Yv, Yp]).tocsr()
    if stats is not None:
        stats.append((nv + npp, kv + kp, a_v, a_p, S.nnz))
    if depth > 0 and kv + kp > coarse_max:
        A1 = (Yv.T @ Ap @ Yv).tocsr()
        J1 = (Yp.T @ J @ Yv).tocsr()
        g1 = (Yv.T @ abs(gv) @ Yv).tocsr()
        S1 = sps.bmat([[A1, J1.T], [J1, None]], format="csr")
        Pc = level(A1, J1, g1, depth - 1, bs=bs, av=av_c or av, ap=ap_c or ap, coarse_max=coarse_max, gamma=gamma,
                   av_c=av_c, ap_c=ap_c, stats=stats, omega=omega, post_coarse=post_coarse, inner=inner)

        def C(rc):
            if inner > 0:       # fixed number of right-preconditioned GMRES steps (outer must be flexible)
                xc, _ = gmres_right(S1, Pc, rc.ravel(), tol=1e-14, restart=inner, maxit=inner)
                return xc.reshape(-1, 1)
            zc = Pc(rc)
            for _ in range(gamma - 1):
                zc = zc + Pc(rc - S1 @ zc)
            return zc
    else:
        Einv = np.linalg.inv((Y.T @ S @ Y).toarray())

        def C(rc):
            return Einv @ rc

    def P(r):
        z = Y @ C(Y.T @ r)
        z = z + omega * P1(r - S @ z)
        if post_coarse:
            z = z + Y @ C(Y.T @ (r - S @ z))
        return z
    return P


if __name__ == "__main__":
    N = int(sys.argv[1])
    nu = float(sys.argv[2])
    pr = pb.ricc_problem(N, nu)
    calA = (-pr.A - pr.Nc).T.tocsr()
    calE = pr.M.T.tocsr()
    J = pr.J
    n = pr.NV + pr.NP
    rng = np.random.default_rng(1)
    b = np.r_[rng.standard_normal(pr.NV), np.zeros(pr.NP)]
    cm = int(4096 * (n / 5.0e5)) if len(sys.argv) > 3 and sys.argv[3] == "scaled" else 4096
    variants = {
        "two-level (shipped)": dict(depth=0),
        "two-level, exact fine coarse": dict(depth=0, coarse_max=10 ** 9),
        "3-level": dict(depth=1),
        "3-level W": dict(depth=1, gamma=2),
        "3-level, coarse 8/12": dict(depth=1, av_c=8, ap_c=12),
        "3-level W post": dict(depth=1, gamma=2, post_coarse=True),
        "4-level W": dict(depth=2, gamma=2),
        "3-level K2": dict(depth=1, inner=2),
        "3-level K3": dict(depth=1, inner=3),
        "3-level K4": dict(depth=1, inner=4),
        "3-level K6": dict(depth=1, inner=6),
        "3-level K10": dict(depth=1, inner=10),
        "4-level K3": dict(depth=2, inner=3),
        "3-level c2": dict(depth=1, av_c=2, ap_c=3),
        "3-level K2 c2": dict(depth=1, inner=2, av_c=2, ap_c=3),
        "3-level K3 c2": dict(depth=1, inner=3, av_c=2, ap_c=3),
        "3-level K4 c2": dict(depth=1, inner=4, av_c=2, ap_c=3),
        "3-level c4": dict(depth=1, av_c=4, ap_c=6),
        "3-level K2 c4": dict(depth=1, inner=2, av_c=4, ap_c=6),
        "3-level K4 c4": dict(depth=1, inner=4, av_c=4, ap_c=6),
        "3-level c8": dict(depth=1, av_c=8, ap_c=12),
        "3-level c16p1": dict(depth=1, av_c=16, ap_c=1),
        "cm1000 3-level c2p2": dict(depth=1, av_c=2, ap_c=2, coarse_max=1000),
        "cm1000 5-level c2p2": dict(depth=3, av_c=2, ap_c=2, coarse_max=1000),
        "cm1000 5-level c2p1": dict(depth=3, av_c=2, ap_c=1, coarse_max=1000),
        "cm1000 two-level": dict(depth=0, coarse_max=1000),
        "cm1000 3-level K2": dict(depth=1, av_c=2, ap_c=2, coarse_max=1000, inner=2),
        "cm1000 3-level K3": dict(depth=1, av_c=2, ap_c=2, coarse_max=1000, inner=3),
        "cm1000 3-level K4": dict(depth=1, av_c=2, ap_c=2, coarse_max=1000, inner=4),
        "cm1000 3-level K8": dict(depth=1, av_c=2, ap_c=2, coarse_max=1000, inner=8),
        "cm1000 3-level K2 p3": dict(depth=1, av_c=2, ap_c=3, coarse_max=1000, inner=2),
        "cm1000 3-level K4 p3": dict(depth=1, av_c=2, ap_c=3, coarse_max=1000, inner=4),
        "cm1000 4-level c3p3": dict(depth=2, av_c=3, ap_c=3, coarse_max=1000),
        "3-level c8p4": dict(depth=1, av_c=8, ap_c=4),
        "3-level K2 c8p4": dict(depth=1, inner=2, av_c=8, ap_c=4),
        "4-level c4p2": dict(depth=2, av_c=4, ap_c=2),
        "4-level c4p3": dict(depth=2, av_c=4, ap_c=3),
        "4-level K2 c4p2": dict(depth=2, av_c=4, ap_c=2, inner=2),
        "two-level av64 ap6": dict(depth=0, av=64, ap=6),
        "two-level av48 ap12": dict(depth=0, av=48, ap=12),
        "two-level av256 ap24": dict(depth=0, av=256, ap=24),
        "two-level av192 ap48": dict(depth=0, av=192, ap=48),
        "3-level K2 c16p1": dict(depth=1, inner=2, av_c=16, ap_c=1),
        "3-level K4 c16p1": dict(depth=1, inner=4, av_c=16, ap_c=1),
        "3-level c2p1": dict(depth=1, av_c=2, ap_c=1),
        "3-level K2 c2p1": dict(depth=1, inner=2, av_c=2, ap_c=1),
        "3-level K4 c2p1": dict(depth=1, inner=4, av_c=2, ap_c=1),
        "3-level K2 c8": dict(depth=1, inner=2, av_c=8, ap_c=12),
        "3-level K3 c8": dict(depth=1, inner=3, av_c=8, ap_c=12),
        "3-level K4 c8": dict(depth=1, inner=4, av_c=8, ap_c=12),
        "3-level K6 c8": dict(depth=1, inner=6, av_c=8, ap_c=12),
    }
    sel = [a for a in sys.argv[3:] if a != "scaled"] or list(variants)
    print("n = %d, coarse_max = %d" % (n, cm))
    for name in sel:
        kw = dict(variants[name])
        kw.setdefault("coarse_max", cm)
        row = []
        t0 = time.time()
        st = []
        for p in (-1.0, -30.0, -1000.0):
            Ap = (calA + p * calE).tocsr()
            st = []
            P = level(Ap, J, calE, stats=st, **kw)
            S = sps.bmat([[Ap, J.T], [J, None]], format="csr")
            x, its = gmres_right(S, P, b)
            row.append(its)
        print("%-30s its p=-1/-30/-1000: %4d %4d %4d   levels (n, kc, av, ap): %s  (%.0fs)"
              % (name, row[0], row[1], row[2], [s[:4] for s in st], time.time() - t0), flush=True)
