"""Developer lab (CPU only): BLOCK GMRES on the scipy mirror of the device cycle -- the m columns of a panel share ONE
Krylov space (block Arnoldi: the new block is orthogonalised against all stored blocks and factorised W = V R), against
the product's m independent per-column GMRES processes.  Counts operator applications per column (= lockstep
iterations of the device batch) to a relative residual of 1e-10 in EVERY column.
python tools/block_lab.py N nu p m [restart_blocks] [rhs: random|adi]"""
import os
import sys

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R)
sys.path.insert(0, os.path.join(_R, "tools"))
import schur_lab as sl  # noqa: E402
from optconpy_amd import problems as pb  # noqa: E402


def column_gmres(Aop, B, tol=1e-10, restart=30, maxit=3000):
    its = []
    for c in range(B.shape[1]):
        b = B[:, c]
        bn = np.linalg.norm(b)
        x = np.zeros_like(b)
        n_it = 0
        while n_it < maxit:
            r = b - Aop(x[:, None])[:, 0]
            beta = np.linalg.norm(r)
            if beta <= tol * bn:
                break
            V = np.zeros((len(b), restart + 1))
            H = np.zeros((restart + 1, restart))
            V[:, 0] = r / beta
            k = 0
            for j in range(restart):
                w = Aop(V[:, j:j + 1])[:, 0]
                for _ in range(2):
                    h = V[:, :j + 1].T @ w
                    H[:j + 1, j] += h
                    w = w - V[:, :j + 1] @ h
                H[j + 1, j] = np.linalg.norm(w)
                V[:, j + 1] = w / H[j + 1, j]
                n_it += 1
                k = j + 1
                e1 = np.zeros(k + 1)
                e1[0] = beta
                y, *_ = np.linalg.lstsq(H[:k + 1, :k], e1, rcond=None)
                if np.linalg.norm(H[:k + 1, :k] @ y - e1) <= tol * bn or n_it >= maxit:
                    break
            x = x + V[:, :k] @ y
        its.append(n_it)
    return its


def block_gmres(Aop, B, tol=1e-10, restart=30, maxit=3000):
    """Returns the number of block iterations (one operator application per column each)."""
    n, m = B.shape
    bn = np.linalg.norm(B, axis=0)
    X = np.zeros_like(B)
    n_it = 0
    while n_it < maxit:
        R = B - Aop(X)
        if np.all(np.linalg.norm(R, axis=0) <= tol * bn):
            break
        V0, R0 = np.linalg.qr(R)
        Vs = [V0]
        H = np.zeros(((restart + 1) * m, restart * m))
        k = 0
        for j in range(restart):
            W = Aop(Vs[j])
            for _ in range(2):
                for i in range(j + 1):
                    h = Vs[i].T @ W
                    H[i * m:(i + 1) * m, j * m:(j + 1) * m] += h
                    W = W - Vs[i] @ h
            Q, Rw = np.linalg.qr(W)
            H[(j + 1) * m:(j + 2) * m, j * m:(j + 1) * m] = Rw
            Vs.append(Q)
            n_it += 1
            k = j + 1
            E1 = np.zeros(((k + 1) * m, m))
            E1[:m, :] = R0
            Y, *_ = np.linalg.lstsq(H[:(k + 1) * m, :k * m], E1, rcond=None)
            res = np.linalg.norm(H[:(k + 1) * m, :k * m] @ Y - E1, axis=0)
            if np.all(res <= tol * bn) or n_it >= maxit:
                break
        X = X + np.column_stack(Vs[:k]) @ Y
    return n_it


if __name__ == "__main__":
    N = int(sys.argv[1]); nu = float(sys.argv[2]); p = float(sys.argv[3]); m = int(sys.argv[4])
    restart = int(sys.argv[5]) if len(sys.argv) > 5 else 30
    kind = sys.argv[6] if len(sys.argv) > 6 else "random"
    pr = pb.ricc_problem(N, nu)
    MT = pr.M.T.tocsr()
    calA = (-pr.A - pr.Nc).T.tocsr()
    Ap = (calA - p * MT).tocsr()
    S, P, kk = sl.make_precond(Ap, pr.J, MT, "base", sa=calA)

    def Aop(U):
        return np.column_stack([S @ P(U[:, c]) for c in range(U.shape[1])])

    rng = np.random.default_rng(3)
    if kind == "adi":
        # right-hand sides shaped like the ADI's: the projected output operator C~^T (smooth, few columns) padded with
        # its images under the mass matrix -- strongly correlated columns, as the residual factors W are
        C = np.asarray(pr.mc_mat.T.todense() if hasattr(pr.mc_mat, "todense") else pr.mc_mat.T)[:, :max(1, m // 4)]
        cols = [C]
        while sum(c.shape[1] for c in cols) < m:
            cols.append(MT @ cols[-1])
        Bv = np.column_stack(cols)[:, :m]
    else:
        Bv = rng.standard_normal((pr.NV, m))
    B = np.vstack([Bv, np.zeros((pr.NP, m))])
    print("N=%d n=%d p=%g m=%d restart=%d rhs=%s" % (N, pr.NV + pr.NP, p, m, restart, kind), flush=True)
    ci = column_gmres(Aop, B[:, :min(m, 4)], restart=restart)
    print("  per-column GMRES (first %d columns): iterations %s" % (len(ci), ci), flush=True)
    bi = block_gmres(Aop, B, restart=restart)
    print("  block GMRES: %d block iterations (lockstep operator applications per column)" % bi, flush=True)
