"""Developer lab (CPU only): GMRES iteration counts of preconditioner variants on the numpy
mirror of the device preconditioner (tools/check_precond.py).  python tools/precond_lab.py N nu"""
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spsla

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from optconpy_amd import _lib, problems as pb  # noqa: E402
from check_precond import lists  # noqa: E402


def gmres_right(S, P, b, tol=1e-10, restart=30, maxit=600):
    n = b.size
    x = np.zeros(n)
    bn = np.linalg.norm(b)
    its = 0
    while its < maxit:
        r = b - S @ x
        beta = np.linalg.norm(r)
        if beta <= tol * bn:
            break
        V = np.zeros((n, restart + 1))
        Zs = np.zeros((n, restart))
        H = np.zeros((restart + 1, restart))
        V[:, 0] = r / beta
        g = np.zeros(restart + 1)
        g[0] = beta
        k = 0
        for j in range(restart):
            Zs[:, j] = P(V[:, j:j + 1]).ravel()
            w = S @ Zs[:, j]
            for _ in range(2):
                h = V[:, :j + 1].T @ w
                w -= V[:, :j + 1] @ h
                H[:j + 1, j] += h
            H[j + 1, j] = np.linalg.norm(w)
            V[:, j + 1] = w / H[j + 1, j]
            its += 1
            k = j + 1
            y, res, *_ = np.linalg.lstsq(H[:k + 1, :k], g[:k + 1], rcond=None)
            rn = np.linalg.norm(H[:k + 1, :k] @ y - g[:k + 1])
            if rn <= tol * bn or its >= maxit:
                break
        x = x + Zs[:, :k] @ y
    return x, its


def build(calA, calE, J, alpha, beta, bs=32, av=16, ap=24, smooth=0.0, smooth_p=0.0, nsm=1, two_sweeps=False,
          omega_bj=1.0, coarse="agg", vcycle=False, post_only=False, bs_p=None, einv_dtype=None, additive=False,
          binv_dtype=None):
    nv, npp = calA.shape[0], J.shape[0]
    pat = (abs(calA) + abs(calE)).tocsr()
    pat.sort_indices()
    blk, nb = _lib.host_aggregate(pat, bs)
    pp = (abs(J) @ abs(J).T).tocsr()
    pp.sort_indices()
    pblk, npb = _lib.host_aggregate(pp, bs_p or bs)
    Ap = (beta * calA + alpha * calE).tocsr()
    S = sps.bmat([[Ap, J.T], [J, None]], format="csr")

    def bj(Mat, blk, nb):
        order, ptr = lists(blk, nb)
        Mp = Mat[order][:, order].tocsr()
        invs = [np.linalg.inv(Mp[ptr[b]:ptr[b + 1], ptr[b]:ptr[b + 1]].toarray()) for b in range(nb)]
        Pm = sps.csr_matrix((np.ones(len(order)), (np.arange(len(order)), order)), shape=(len(order),) * 2)
        return (Pm.T @ sps.block_diag(invs) @ Pm).tocsr()
    AinvM = bj(Ap, blk, nb)
    Sh = (J @ AinvM @ J.T).tocsr()
    SinvM = bj(Sh, pblk, npb)
    g = calE if calE.nnz > 2 * nv else pat
    va, kv = _lib.host_aggregate(g, av)
    pa, kp = _lib.host_aggregate(pp, ap)
    Yv = sps.csr_matrix((np.ones(nv), (np.arange(nv), va)), shape=(nv, kv))
    Yp = sps.csr_matrix((np.ones(npp), (np.arange(npp), pa)), shape=(npp, kp))
    if smooth > 0.0:
        # smoothed aggregation on the velocity block: P = (I - w D^-1 K) P0 with the symmetric part
        K = (0.5 * (Ap + Ap.T)).tocsr()
        Dinv = sps.diags(1.0 / K.diagonal())
        for _ in range(nsm):
            Yv = (Yv - smooth * (Dinv @ (K @ Yv))).tocsr()
    if smooth_p > 0.0:
        Kp = Sh
        Dinv = sps.diags(1.0 / Kp.diagonal())
        Yp = (Yp - smooth_p * (Dinv @ (Kp @ Yp))).tocsr()
    Y = sps.block_diag([Yv, Yp]).tocsr()
    Ec = (Y.T @ S @ Y).toarray()
    Einv = np.linalg.inv(Ec)
    if einv_dtype == "f16":
        sc = np.abs(Einv).max() / 6e4
        Einv = (Einv / sc).astype(np.float16).astype(np.float64) * sc
    elif einv_dtype == "f16row":
        sc = np.abs(Einv).max(axis=1, keepdims=True) / 6e4
        Einv = (Einv / sc).astype(np.float16).astype(np.float64) * sc
    elif einv_dtype == "bf16":
        import torch
        Einv = torch.from_numpy(Einv).to(torch.bfloat16).to(torch.float64).numpy()
    elif einv_dtype == "f32":
        Einv = Einv.astype(np.float32).astype(np.float64)
    if binv_dtype == "f16":
        def q(Mx):
            Mx = Mx.tocsr(copy=True)
            sc = np.abs(Mx.data).max() / 6e4
            Mx.data = (Mx.data / sc).astype(np.float16).astype(np.float64) * sc
            return Mx
        AinvM, SinvM = q(AinvM), q(SinvM)
    info = dict(kc=Y.shape[1], nnzY=Y.nnz / Y.shape[0], nnzSY=(S @ Y).nnz / S.shape[0])

    def P1(r):
        zv = AinvM @ r[:nv]
        zp = SinvM @ (J @ zv - r[nv:])
        zv = zv - AinvM @ (J.T @ zp)
        return np.vstack([zv, zp])

    def P(r):
        if vcycle:
            z = omega_bj * P1(r)
            z = z + Y @ (Einv @ (Y.T @ (r - S @ z)))
            return z + omega_bj * P1(r - S @ z)
        if additive:
            return Y @ (Einv @ (Y.T @ r)) + omega_bj * P1(r)
        if post_only:
            z = P1(r)
            return z + Y @ (Einv @ (Y.T @ (r - S @ z)))
        z = Y @ (Einv @ (Y.T @ r))
        z = z + omega_bj * P1(r - S @ z)
        if two_sweeps:
            z = z + Y @ (Einv @ (Y.T @ (r - S @ z)))
        return z
    return P, S, info


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
    variants = {
        "base": dict(),
        "SA w=0.5": dict(smooth=0.5),
        "SA w=0.67": dict(smooth=0.67),
        "SA w=0.67 x2": dict(smooth=0.67, nsm=2),
        "SA v0.67 p0.5": dict(smooth=0.67, smooth_p=0.5),
        "coarse twice (V-cycle like)": dict(two_sweeps=True),
        "SA0.67 + coarse twice": dict(smooth=0.67, two_sweeps=True),
        "av=8,ap=12": dict(av=8, ap=12),
        "Einv f32": dict(einv_dtype="f32"),
        "Einv f16": dict(einv_dtype="f16"),
        "Einv f16 rowscale": dict(einv_dtype="f16row"),
        "Einv bf16": dict(einv_dtype="bf16"),
        "Einv+blocks f16": dict(einv_dtype="f16row", binv_dtype="f16"),
        "additive": dict(additive=True),
        "additive w=0.7": dict(additive=True, omega_bj=0.7),
        "base w=0.8": dict(omega_bj=0.8),
        "base w=0.7": dict(omega_bj=0.7),
        "base w=0.6": dict(omega_bj=0.6),
        "V(1,1) w=0.6": dict(vcycle=True, omega_bj=0.6),
        "V(1,1) w=0.8": dict(vcycle=True, omega_bj=0.8),
        "V(1,1) w=0.7 bs64": dict(vcycle=True, omega_bj=0.7, bs=64),
        "V(1,1) w=0.7 SA.5": dict(vcycle=True, omega_bj=0.7, smooth=0.5),
        "V(1,1)": dict(vcycle=True),
        "V(1,1) w=0.7": dict(vcycle=True, omega_bj=0.7),
        "smooth then coarse": dict(post_only=True),
        "bs=64": dict(bs=64),
        "bs=16": dict(bs=16),
        "bs=32,bs_p=64": dict(bs_p=64),
        "V(1,1)+SA": dict(vcycle=True, smooth=0.67),
    }
    sel = sys.argv[3:] or list(variants)
    for name in sel:
        kw = variants[name]
        row = []
        t0 = time.time()
        for p in (-1.0, -30.0, -1000.0):
            P, S, info = build(calA, calE, J, p, 1.0, **kw)
            x, its = gmres_right(S, P, b)
            row.append(its)
        print("%-28s its p=-1/-30/-1000: %4d %4d %4d   kc=%d nnz/row Y %.1f SY %.1f  (%.0fs)"
              % (name, row[0], row[1], row[2], info["kc"], info["nnzY"], info["nnzSY"], time.time() - t0), flush=True)
