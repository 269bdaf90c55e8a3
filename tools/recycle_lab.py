"""Developer lab (CPU only): subspace recycling (GCRO-DR style: k harmonic Ritz vectors of the last cycle carried over,
the space re-compressed to the k slowest directions) across a sequence of solves with ONE shifted saddle matrix, on the
scipy mirror of the device cycle (tools/schur_lab.py, smoothed prolongation).  Result: 62 -> 57
iterations at p = 1 with k = 10, 58 -> 53 with k = 20 and restart 60, nothing at p = 300 -- the spectrum of the
preconditioned operator has no few outliers to deflate.
python tools/recycle_lab.py N nu p k [restart]"""
import os, sys, time
import numpy as np, scipy.sparse as sps
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tools"))
import schur_lab as sl
from optconpy_amd import problems as pb

def arnoldi_solve(Aop, b, x0, C=None, U=None, tol=1e-10, restart=30, maxit=2000):
    """GMRES on Aop (already preconditioned operator u -> S P u), optionally deflated by (U, C): C = Aop U, C^T C = I.
    Returns u (solution of Aop u = b), iterations, and (V, Hbar) of the LAST cycle."""
    bn = np.linalg.norm(b)
    x = x0.copy()
    r = b - Aop(x)
    if C is not None:
        y = C.T @ r
        x = x + U @ y
        r = r - C @ y
    its = 0
    last = None
    while its < maxit:
        beta = np.linalg.norm(r)
        if beta <= tol * bn:
            break
        V = np.zeros((len(b), restart + 1)); H = np.zeros((restart + 1, restart)); B = None
        if C is not None:
            B = np.zeros((C.shape[1], restart))
        V[:, 0] = r / beta
        k = 0
        for j in range(restart):
            w = Aop(V[:, j])
            if C is not None:
                B[:, j] = C.T @ w
                w = w - C @ B[:, j]
            for _ in range(2):
                h = V[:, :j + 1].T @ w
                H[:j + 1, j] += h
                w = w - V[:, :j + 1] @ h
            H[j + 1, j] = np.linalg.norm(w)
            V[:, j + 1] = w / H[j + 1, j]
            its += 1; k = j + 1
            e1 = np.zeros(k + 1); e1[0] = beta
            y, *_ = np.linalg.lstsq(H[:k + 1, :k], e1, rcond=None)
            rn = np.linalg.norm(H[:k + 1, :k] @ y - e1)
            if rn <= tol * bn or its >= maxit:
                break
        dx = V[:, :k] @ y
        if C is not None:
            x = x + dx - U @ (B[:, :k] @ y)
        else:
            x = x + dx
        r = b - Aop(x)
        if C is not None:           # keep r orthogonal to C (it is, up to rounding)
            pass
        last = (V[:, :k + 1].copy(), H[:k + 1, :k].copy())
    return x, its, last

def harmonic_ritz(V, Hb, k):
    m = Hb.shape[1]
    Hm = Hb[:m, :m]
    em = np.zeros(m); em[-1] = 1.0
    f = np.linalg.solve(Hm.T, em)
    M = Hm + (Hb[m, m - 1] ** 2) * np.outer(f, em)
    th, G = np.linalg.eig(M)
    idx = np.argsort(abs(th))[:k]
    Gs = G[:, idx]
    # real basis of the span
    Gr = np.column_stack([Gs.real, Gs.imag])
    Q, R = np.linalg.qr(Gr)
    keep = abs(np.diag(R)) > 1e-10 * abs(R).max()
    Q = Q[:, keep][:, :k]
    return V[:, :m] @ Q

if __name__ == "__main__":
    N = int(sys.argv[1]); nu = float(sys.argv[2]); p = float(sys.argv[3]); k = int(sys.argv[4])
    restart = int(sys.argv[5]) if len(sys.argv) > 5 else 30
    pr = pb.ricc_problem(N, nu)
    MT = pr.M.T.tocsr(); calA = (-pr.A - pr.Nc).T.tocsr()
    Ap = (calA - p * MT).tocsr()
    S, P, kk = sl.make_precond(Ap, pr.J, MT, "base", sa=calA)
    Aop = lambda u: S @ P(u)
    rng = np.random.default_rng(3)
    nrhs = 6
    # a sequence of right-hand sides; "corr": each the previous solution pushed through M (as ADI's residual factors evolve)
    U = C = None
    print("N=%d p=%g k=%d restart=%d" % (N, p, k, restart))
    for t in range(nrhs):
        b = np.r_[rng.standard_normal(pr.NV), np.zeros(pr.NP)]
        x0 = np.zeros_like(b)
        _, it0, _ = arnoldi_solve(Aop, b, x0, restart=restart)
        u, it1, last = arnoldi_solve(Aop, b, x0, C=C, U=U, restart=restart)
        print("  rhs %d: plain %d   recycled %d" % (t, it0, it1), flush=True)
        # update the recycle space from the last cycle's harmonic Ritz vectors (plus the old space)
        V, Hb = last
        Y = harmonic_ritz(V, Hb, k)
        if U is not None:
            Y = np.column_stack([U, Y])       # old space + new vectors, then compress to k by QR-SVD of A Y
        AY = np.column_stack([Aop(Y[:, i]) for i in range(Y.shape[1])])
        Q, R = np.linalg.qr(AY)
        if Y.shape[1] > k:
            # keep the k directions of smallest singular value of R (slow modes of the operator on this space)
            uu, ss, vt = np.linalg.svd(R)
            sel = vt[-k:, :].T
            Y = Y @ sel
            AY = AY @ sel
            Q, R = np.linalg.qr(AY)
        C = Q; U = Y @ np.linalg.inv(R)
