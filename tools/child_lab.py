"""Developer lab (CPU only): what stands in for the dense coarse inverse on a THIRD level.  Parent level exactly as
tools/schur_lab.py (smoothed prolongation, SIMPLE sweep with 32-row blocks); the coarse saddle matrix
E = P^T S P (kv + kp unknowns) is
  dense   inverted (two levels, what the product does up to coarse_max),
  simple  one cycle of a child level: pairwise aggregation of the coarse velocity unknowns (pressure 1:1), dense inverse
          of the child's Galerkin matrix, then ONE SIMPLE sweep on E (block-Jacobi of E_A, Schur blocks) -- the
          product's child level,
  vanka   the same coarse correction, then ONE additive Vanka sweep: per coarse pressure unknown the patch of itself and
          the velocity unknowns its row of E_J touches, local saddle systems solved exactly, overlapping velocity
          corrections averaged (damping om),
  vankam  the Vanka sweep multiplicative (patch after patch on the updated residual).
Counts outer GMRES iterations to 1e-10.
python tools/child_lab.py N nu p [variants...]"""
import os
import sys

import numpy as np
import scipy.sparse as sps

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R)
sys.path.insert(0, os.path.join(_R, "tools"))
import schur_lab as sl  # noqa: E402
from optconpy_amd import problems as pb  # noqa: E402


def build(Ap, J, gv, calA, av, ap, bs=32):
    nv, npp = Ap.shape[0], J.shape[0]
    S = sps.bmat([[Ap, J.T], [J, None]], format="csr")
    blk, nb = sl.agg(abs(Ap), bs)
    pp = (abs(J) @ abs(J).T).tocsr()
    pblk, npb = sl.agg(pp, bs)
    Ainv = sl.bj_inverse(Ap, blk, nb)
    Sh = (J @ Ainv @ J.T).tocsr()
    Sinv = sl.bj_inverse(Sh, pblk, npb)
    va, kv = sl.agg(gv, av)
    pa, kp = sl.agg(pp, ap)
    Yv = sps.csr_matrix((np.ones(nv), (np.arange(nv), va)), shape=(nv, kv))
    Yp = sps.csr_matrix((np.ones(npp), (np.arange(npp), pa)), shape=(npp, kp))
    K0 = (0.5 * (calA + calA.T)).tocsr()
    dinv = 1.0 / calA.diagonal()
    Yv = (Yv - 0.5 * (sps.diags(dinv) @ (K0 @ Yv))).tocsr()
    Y = sps.block_diag([Yv, Yp]).tocsr()
    E = (Y.T @ S @ Y).toarray()

    def simple(r):
        zv = Ainv @ r[:nv]
        zp = Sinv @ (J @ zv - r[nv:])
        zv = zv - Ainv @ (J.T @ zp)
        return np.r_[zv, zp]
    return S, Y, E, kv, kp, simple


def child_cycle(E, kv, kp, kind, om=0.7, pair=2, cbs=32):
    k = kv + kp
    EA, EJt, EJ = E[:kv, :kv], E[:kv, kv:], E[kv:, :kv]
    # child coarse space: aggregates of `pair` coarse velocity unknowns on the graph of E_A, pressure 1:1
    g = sps.csr_matrix(abs(EA) > 1e-12 * abs(EA).max())
    va2, kv2 = sl.agg(g.astype(float), pair)
    Y2 = sps.block_diag([sps.csr_matrix((np.ones(kv), (np.arange(kv), va2)), shape=(kv, kv2)), sps.identity(kp)]).tocsr()
    E2inv = np.linalg.inv(Y2.T @ E @ Y2)

    def coarse(r):
        return Y2 @ (E2inv @ (Y2.T @ r))

    if kind == "simple":
        blk, nb = sl.agg(g.astype(float), cbs)
        Ainv = sl.bj_inverse(sps.csr_matrix(EA), blk, nb).toarray()
        Sh = EJ @ Ainv @ EJt
        pg = sps.csr_matrix(abs(Sh) > 1e-12 * abs(Sh).max()).astype(float)
        pblk, npb = sl.agg(pg, cbs)
        Sinv = sl.bj_inverse(sps.csr_matrix(Sh), pblk, npb).toarray()

        def sweep(r):
            zv = Ainv @ r[:kv]
            zp = Sinv @ (EJ @ zv - r[kv:])
            zv = zv - Ainv @ (EJt @ zp)
            return np.r_[zv, zp]
    else:
        patches = []
        for i in range(kp):
            vs = np.nonzero(abs(EJ[i]) > 1e-12 * abs(EJ).max())[0]
            idx = np.r_[vs, kv + i]
            patches.append((idx, np.linalg.inv(E[np.ix_(idx, idx)])))
        cnt = np.zeros(k)
        for idx, _ in patches:
            cnt[idx] += 1
        # velocity unknowns no pressure row touches: plain Jacobi on E_A
        lone = np.nonzero(cnt[:kv] == 0)[0]
        dl = 1.0 / np.diag(EA)[lone] if len(lone) else None
        print("    vanka patches: %d, mean size %.1f, max %d, velocity overlap %.2f, untouched velocity unknowns %d" % (
            len(patches), np.mean([len(p[0]) for p in patches]), max(len(p[0]) for p in patches),
            cnt[:kv][cnt[:kv] > 0].mean(), len(lone)), flush=True)

        # restricted additive (RAS): every velocity unknown takes the correction of ONE owner patch (the pressure row
        # with the largest |E_J| entry for it)
        owner = np.full(kv, -1)
        best = np.zeros(kv)
        for i in range(kp):
            w = abs(EJ[i])
            take = w > best
            owner[take] = i
            best[take] = w[take]
        if kind == "vankar":
            def sweep(r):
                z = np.zeros(k)
                for i, (idx, inv) in enumerate(patches):
                    d = inv @ r[idx]
                    vs = idx[:-1]
                    mine = owner[vs] == i
                    z[vs[mine]] = om * d[:-1][mine]
                    z[kv + i] = om * d[-1]
                if len(lone):
                    z[lone] = om * dl * r[lone]
                return z
        elif kind == "vankatwo":          # two damped additive sweeps
            def one(r):
                z = np.zeros(k)
                for idx, inv in patches:
                    z[idx] += inv @ r[idx]
                z[:kv] = om * z[:kv] / np.maximum(cnt[:kv], 1)
                z[kv:] = om * z[kv:]
                if len(lone):
                    z[lone] = om * dl * r[lone]
                return z

            def sweep(r):
                z = one(r)
                return z + one(r - E @ z)
        elif kind == "vankac":            # coloured: patches of one colour share no unknown and are applied together
            sets = [set(idx.tolist()) for idx, _ in patches]
            colour = [-1] * len(patches)
            for i in range(len(patches)):
                used = {colour[j] for j in range(i) if sets[i] & sets[j]}
                c = 0
                while c in used:
                    c += 1
                colour[i] = c
            ncol = max(colour) + 1
            groups = [[i for i in range(len(patches)) if colour[i] == c] for c in range(ncol)]
            print("    coloured Vanka: %d colours, patches per colour %s" % (ncol, [len(g_) for g_ in groups]), flush=True)

            def sweep(r):
                z = np.zeros(k)
                res = r.copy()
                for g_ in groups:
                    d = np.zeros(k)
                    for i in g_:
                        idx, inv = patches[i]
                        d[idx] = om * (inv @ res[idx])
                    z += d
                    res -= E @ d
                if len(lone):
                    z[lone] += om * dl * res[lone]
                return z
        elif kind == "vanka":
            def sweep(r):
                z = np.zeros(k)
                for idx, inv in patches:
                    z[idx] += inv @ r[idx]
                z[:kv] = om * z[:kv] / np.maximum(cnt[:kv], 1)
                z[kv:] = om * z[kv:]
                if len(lone):
                    z[lone] = om * dl * r[lone]
                return z
        else:
            def sweep(r):
                z = np.zeros(k)
                res = r.copy()
                for idx, inv in patches:
                    d = om * (inv @ res[idx])
                    z[idx] += d
                    res -= E[:, idx] @ d
                if len(lone):
                    z[lone] += om * dl * res[lone]
                return z

    def cycle(r):
        z = coarse(r)
        return z + sweep(r - E @ z)
    return cycle, kv2


if __name__ == "__main__":
    N = int(sys.argv[1]); nu = float(sys.argv[2]); p = float(sys.argv[3])
    variants = sys.argv[4:] or ["dense", "simple", "vanka", "vankam"]
    av = int(os.environ.get("AV", "16")); ap = int(os.environ.get("AP", "24"))
    pr = pb.ricc_problem(N, nu)
    MT = pr.M.T.tocsr()
    calA = (-pr.A - pr.Nc).T.tocsr()
    if os.environ.get("DRE") == "1":      # operator of a DRE time step (schur_lab.py): -(M^T / 2 + tau (A + N)^T)
        tau = float(np.diff(pb.get_tint(0.0, 1.0, 16, True)).max())
        calA = (-(0.5 * MT + tau * (pr.A.T + pr.Nc.T))).tocsr()
    Ap = (calA - p * MT).tocsr()
    S, Y, E, kv, kp, simple = build(Ap, pr.J, MT, calA, av, ap)
    b = np.r_[np.random.default_rng(1).standard_normal(pr.NV), np.zeros(pr.NP)]
    print("N=%d n=%d p=%g aggregates (%d, %d): coarse (kv, kp) = (%d, %d)" % (N, pr.NV + pr.NP, p, av, ap, kv, kp), flush=True)
    Einv = np.linalg.inv(E)
    for v in variants:
        kind = v.rstrip("0123456789.")
        om = float(v[len(kind):]) if len(v) > len(kind) else 0.7
        if kind == "dense":
            capply = lambda r: Einv @ r  # noqa: E731
            extra = ""
        else:
            capply, kv2 = child_cycle(E, kv, kp, kind, om=om, pair=int(os.environ.get("PAIR", "2")))
            extra = " child coarse %d" % (kv2 + kp)

        def P(r):
            z = Y @ capply(Y.T @ r)
            return z + simple(r - S @ z)
        x, its = sl.gmres_right(S, P, b)
        print("  %-10s outer iterations %d%s" % (v, its, extra), flush=True)
