"""Developer check: device preconditioner vs a numpy mirror built from the same
host aggregation.  python tools/check_precond.py N nu"""
import sys

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, ".")
from optconpy_amd import _lib, problems as pb  # noqa: E402


def lists(blk, nb):
    order = np.argsort(blk, kind="stable")
    cnt = np.bincount(blk, minlength=nb)
    ptr = np.r_[0, np.cumsum(cnt)]
    return order, ptr


def mirror(calA, calE, J, alpha, beta, bs=32, av=16, ap=24, coarse_max=4096, use_coarse=True):
    nv, npp = calA.shape[0], J.shape[0]
    pat = (abs(calA) + abs(calE)).tocsr()
    pat.sort_indices()
    blk, nb = _lib.host_aggregate(pat, bs)
    pp = (abs(J) @ abs(J).T).tocsr()
    pp.sort_indices()
    pblk, npb = _lib.host_aggregate(pp, bs)
    Ap = (beta * calA + alpha * calE).tocsr()
    D = Ap.diagonal()
    S = sps.bmat([[Ap, J.T], [J, None]], format="csr")

    def bj(Mat, blk, nb):
        order, ptr = lists(blk, nb)
        Mp = Mat[order][:, order].tocsr()
        invs = [np.linalg.inv(Mp[ptr[b]:ptr[b + 1], ptr[b]:ptr[b + 1]].toarray()) for b in range(nb)]

        def app(r):
            rp = r[order]
            z = np.empty_like(rp)
            for b in range(nb):
                z[ptr[b]:ptr[b + 1]] = invs[b] @ rp[ptr[b]:ptr[b + 1]]
            out = np.empty_like(z)
            out[order] = z
            return out
        return app, order, invs
    Ainv, aorder, ainvs = bj(Ap, blk, nb)
    # consistent SIMPLE: Schur complement and velocity update with the block-Jacobi inverse
    Pm = sps.csr_matrix((np.ones(nv), (np.arange(nv), aorder)), shape=(nv, nv))
    AinvM = (Pm.T @ sps.block_diag(ainvs) @ Pm).tocsr()
    Sh = (J @ AinvM @ J.T).tocsr()
    Sinv, _, _ = bj(Sh, pblk, npb)
    Y = None
    if use_coarse:
        g = calE if calE.nnz > 2 * nv else pat
        while True:
            va, kv = _lib.host_aggregate(g, av)
            pa, kp = _lib.host_aggregate(pp, ap)
            if kv + kp <= max(16, coarse_max):
                break
            av *= 2
            ap *= 2
        Y = sps.block_diag([sps.csr_matrix((np.ones(nv), (np.arange(nv), va)), shape=(nv, kv)),
                            sps.csr_matrix((np.ones(npp), (np.arange(npp), pa)), shape=(npp, kp))]).tocsr()
        Einv = np.linalg.inv((Y.T @ S @ Y).toarray())

    def P1(r):
        zv = Ainv(r[:nv])
        zp = Sinv(J @ zv - r[nv:])
        zv = zv - Ainv(J.T @ zp)
        return np.vstack([zv, zp])

    def P(r):
        if Y is None:
            return P1(r)
        z = Y @ (Einv @ (Y.T @ r))
        return z + P1(r - S @ z)
    return P, S


if __name__ == "__main__":
    N = int(sys.argv[1])
    nu = float(sys.argv[2])
    pr = pb.ricc_problem(N, nu)
    calA = (-pr.A - pr.Nc).T.tocsr()
    calE = pr.M.T.tocsr()
    J = pr.J
    n = pr.NV + pr.NP
    rng = np.random.default_rng(1)
    R = rng.standard_normal((n, 16))
    for use_coarse in (0, 1):
        ctx = _lib.Context(0, verbose=1, use_coarse=use_coarse)
        ctx.set_operator(calA, calE, J)
        for (al, be) in ((-1.0, 1.0), (-100.0, 1.0), (1.0, 0.0)):
            P, S = mirror(calA, calE, J, al, be, use_coarse=bool(use_coarse))
            Zm = P(R)
            Zd = ctx.precond_apply(al, be, R)
            e = np.linalg.norm(Zd - Zm) / np.linalg.norm(Zm)
            ev = np.linalg.norm(Zd[:pr.NV] - Zm[:pr.NV]) / np.linalg.norm(Zm[:pr.NV])
            ep = np.linalg.norm(Zd[pr.NV:] - Zm[pr.NV:]) / np.linalg.norm(Zm[pr.NV:])
            print("coarse=%d (alpha,beta)=(%g,%g): rel diff %.2e (v %.2e p %.2e) |Z| %.3e" %
                  (use_coarse, al, be, e, ev, ep, np.linalg.norm(Zm)), flush=True)
        ctx.close()
