"""Developer lab (CPU only): the device preconditioner cycle on a scipy mirror, with variants of the PRESSURE step.
  base:   z = Y E^-1 Y^T r ;  z += SIMPLE(r - S z)      (plain aggregation (16, 24), 32-blocks; as shipped without smoothing)
  schur2: the SIMPLE sweep's Schur solve  z_p = B_S t  is two-level: coarse Galerkin of S^ = J A^^-1 J^T on the pressure
          aggregates first, then the block-Jacobi sweep on the updated t
python tools/schur_lab.py N nu [dre]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optconpy_amd import _lib, problems as pb  # noqa: E402


def agg(graph, size):
    g = sps.csr_matrix(graph)
    g.sort_indices()
    blk, nb = _lib.host_aggregate(g, size)
    return np.asarray(blk), nb


def bj_inverse(Mat, blk, nb):
    """Block-diagonal inverse of Mat for the partition blk (sparse)."""
    order = np.argsort(blk, kind="stable")
    cnt = np.bincount(blk, minlength=nb)
    ptr = np.r_[0, np.cumsum(cnt)]
    Mp = sps.csr_matrix(Mat)[order][:, order].tocsr()
    invs = [np.linalg.inv(Mp[ptr[b]:ptr[b + 1], ptr[b]:ptr[b + 1]].toarray()) for b in range(nb)]
    n = len(order)
    Pm = sps.csr_matrix((np.ones(n), (np.arange(n), order)), shape=(n, n))
    return (Pm.T @ sps.block_diag(invs) @ Pm).tocsr()


def gmres_right(S, P, b, tol=1e-10, restart=30, maxit=1500):
    """Right-preconditioned restarted GMRES (MGS), iterations to relative residual tol."""
    x = np.zeros_like(b)
    bn = np.linalg.norm(b)
    its = 0
    while its < maxit:
        r = b - S @ x
        beta = np.linalg.norm(r)
        if beta <= tol * bn:
            break
        V = [r / beta]
        Z = []
        H = np.zeros((restart + 1, restart))
        k = 0
        for j in range(restart):
            z = P(V[j])
            w = S @ z
            for i in range(j + 1):
                H[i, j] = V[i] @ w
                w = w - H[i, j] * V[i]
            for i in range(j + 1):
                h = V[i] @ w
                H[i, j] += h
                w = w - h * V[i]
            H[j + 1, j] = np.linalg.norm(w)
            V.append(w / H[j + 1, j])
            Z.append(z)
            its += 1
            k = j + 1
            e1 = np.zeros(k + 1)
            e1[0] = beta
            y, res, _, _ = np.linalg.lstsq(H[:k + 1, :k], e1, rcond=None)
            rn = np.linalg.norm(H[:k + 1, :k] @ y - e1)
            if rn <= tol * bn or its >= maxit:
                break
        x = x + np.column_stack(Z[:k]) @ y
    return x, its


def make_precond(Ap, J, gv, variant, bs=32, av=16, ap=24, cc=None, pbs=None, sa=None, sa_steps=1, sa_shifted=False):
    nv, npp = Ap.shape[0], J.shape[0]
    S = sps.bmat([[Ap, J.T], [J, None]], format="csr")
    blk, nb = agg(abs(Ap), bs)
    pp = (abs(J) @ abs(J).T).tocsr()
    pblk, npb = agg(pp, pbs or bs)      # pbs: pressure (Schur) blocks of their own size
    Ainv = bj_inverse(Ap, blk, nb)
    Sh = (J @ Ainv @ J.T).tocsr()
    Sinv = bj_inverse(Sh, pblk, npb)
    va, kv = agg(gv, av)
    pa, kp = agg(pp, ap)
    Yv = sps.csr_matrix((np.ones(nv), (np.arange(nv), va)), shape=(nv, kv))
    Yp = sps.csr_matrix((np.ones(npp), (np.arange(npp), pa)), shape=(npp, kp))
    if sa is not None:
        # smoothed aggregation of the velocity prolongation as the product builds it (ricadi_host.cpp:build_setup):
        # P_v = (I - omega D^-1 sym(cal A)) Y_v, omega = 0.5 (scaled down where rho(D^-1 K0) > 2), shift independent
        K0 = (0.5 * (sa + sa.T)).tocsr()
        dinv = 1.0 / sa.diagonal()
        x = np.random.default_rng(5).standard_normal(nv)
        for _ in range(20):
            y = dinv * (K0 @ x)
            rho = np.linalg.norm(y) / np.linalg.norm(x)
            x = y / np.linalg.norm(y)
        om = 0.5 * (2.0 / rho if rho > 2.0 else 1.0)
        if sa_shifted:             # smooth with the SHIFTED operator's symmetric part (per-shift prolongation)
            K0 = (0.5 * (Ap + Ap.T)).tocsr()
            dinv = 1.0 / Ap.diagonal()
        for _ in range(sa_steps):
            Yv = (Yv - om * (sps.diags(dinv) @ (K0 @ Yv))).tocsr()
    Y = sps.block_diag([Yv, Yp]).tocsr()
    Einv = np.linalg.inv((Y.T @ S @ Y).toarray())
    if variant.startswith("schur2"):
        fine = variant.endswith("f")       # finer pressure aggregates for the Schur coarse space
        if fine:
            pa2, kp2 = agg(pp, 8)
            Yq = sps.csr_matrix((np.ones(npp), (np.arange(npp), pa2)), shape=(npp, kp2))
        else:
            Yq = Yp
        Scinv = np.linalg.inv((Yq.T @ Sh @ Yq).toarray())

    if variant.startswith("cc") and cc is not None:
        # Cahouet-Chabard form of the Schur inverse: -(c nu Mp^-1 + p [J Mb^-1 J^T]_b^-1), Mp ~ h^2 I (lumped P1 mass)
        MTm, pshift, nu_, h_ = cc
        cfac = float(variant[2:] or 1.0)
        Minv_b = bj_inverse(MTm, blk, nb)
        Lb_inv = bj_inverse((J @ Minv_b @ J.T).tocsr(), pblk, npb)
        Sinv = (-(cfac * nu_ / h_ ** 2) * sps.identity(npp) - pshift * Lb_inv).tocsr()
    om_p = float(variant[3:]) if variant.startswith("omp") else 1.0     # "omp0.8": damped pressure update
    om_v = float(variant[3:]) if variant.startswith("omv") else 1.0     # "omv0.8": damped velocity correction
    if variant.startswith("oms"):                                       # "oms0.9": the whole sweep damped
        om_s = float(variant[3:])
    else:
        om_s = 1.0

    def schur_solve(t):
        if om_p != 1.0:
            return om_p * (Sinv @ t)
        if variant.startswith("schur2"):
            zc = Yq @ (Scinv @ (Yq.T @ t))
            return zc + Sinv @ (t - Sh @ zc)
        return Sinv @ t

    import re as _re
    _m = _re.search(r"oma([0-9.]+)", variant)        # "oma0.7", "post2oma0.7": damped velocity predictor
    om_a = float(_m.group(1)) if _m else 1.0

    _m2 = _re.search(r"bsa([0-9.]+)", variant)       # "bsa1.5": Braess-Sarazin scaling alpha of the block inverse
    bs_a = float(_m2.group(1)) if _m2 else 1.0

    toks = variant.split("+")
    if "b16" in toks or "m32" in toks or "mb2" in toks:
        # block operands as the device stores them (BF16: 8 mantissa bits, round to nearest even)
        def bf16(a):
            u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
            u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
            return u.view(np.float32)
        Ainv = sps.csr_matrix((bf16(Ainv.data).astype(np.float64), Ainv.indices, Ainv.indptr), shape=Ainv.shape)
        Sinv = sps.csr_matrix((bf16(Sinv.data).astype(np.float64), Sinv.indices, Sinv.indptr), shape=Sinv.shape)
        AinvJT = (Ainv @ J.T).tocsr()      # the last sweep's per-block rectangles, BF16-stored too
        AinvJT = sps.csr_matrix((bf16(AinvJT.data).astype(np.float64), AinvJT.indices, AinvJT.indptr), shape=AinvJT.shape)
        A32, S32, R32 = Ainv.astype(np.float32), Sinv.astype(np.float32), AinvJT.astype(np.float32)

        def mm(M64, M32, x):
            """block product as the matrix cores would do it: m32 = FP32 operands and accumulation,
            mb2 = panel side rounded to two BF16 pieces (16 mantissa bits), FP32 accumulation"""
            if "m32" in toks:
                return (M32 @ x.astype(np.float32)).astype(np.float64)
            if "mb2" in toks:
                hi = bf16(x)
                lo = bf16(x - hi.astype(np.float64))
                return (M32 @ hi).astype(np.float64) + (M32 @ lo).astype(np.float64)
            return M64 @ x

        def simple(r):        # noqa: F811 -- the cycle on the rounded operands
            zv = mm(Ainv, A32, r[:nv])
            zp = mm(Sinv, S32, J @ zv - r[nv:])
            zv = zv - mm(AinvJT, R32, zp)
            return np.r_[zv, zp]
        _simple_rounded = simple

    def simple(r):
        if "b16" in toks or "m32" in toks or "mb2" in toks:
            return _simple_rounded(r)
        if bs_a != 1.0:            # A^ -> alpha A^ in predictor, Schur complement and correction
            zv = (Ainv @ r[:nv]) / bs_a
            zp = bs_a * schur_solve(J @ zv - r[nv:])
            zv = zv - (Ainv @ (J.T @ zp)) / bs_a
            return np.r_[zv, zp]
        zv = om_a * (Ainv @ r[:nv])
        zp = schur_solve(J @ zv - r[nv:])
        zv = zv - om_v * (Ainv @ (J.T @ zp))
        return om_s * np.r_[zv, zp]

    keep_coarse_form = any(t in ("b16", "m32", "mb2") for t in variant.split("+"))
    c32 = "c32" in variant.split("+")            # coarse apply in FP32 ARITHMETIC on the FP32-stored inverse
    c32h = "c32h" in variant.split("+")          # ... head of the coarse residual only (no tail)
    if c32 or c32h:
        Einv32 = Einv.astype(np.float32)
        Yt = Y.T.tocsr()

        def coarse(r):
            rc = Yt @ r
            hd = rc.astype(np.float32)
            e = (Einv32 @ hd).astype(np.float64)
            if c32:
                e = e + (Einv32 @ (rc - hd).astype(np.float32)).astype(np.float64)
            return Y @ e
    else:
        E32 = Einv.astype(np.float32).astype(np.float64)     # the product stores the inverse in FP32

        def coarse(r):
            return Y @ (E32 @ (Y.T @ r))

    def P(r):
        if "c32" in variant or "c64" in variant or keep_coarse_form:
            z = coarse(r)
            return z + simple(r - S @ z)
        if variant == "nocoarse":
            return simple(r)
        if variant.startswith("pre"):               # sweep first, coarse correction on its residual
            z = simple(r)
            return z + Y @ (Einv @ (Y.T @ (r - S @ z)))
        if variant.startswith("v11"):               # pre-sweep, coarse correction, post-sweep
            z = simple(r)
            r1 = r - S @ z
            z = z + Y @ (Einv @ (Y.T @ r1))
            return z + simple(r - S @ z)
        if variant.startswith("post2"):             # coarse correction, two post-sweeps
            z = Y @ (Einv @ (Y.T @ r))
            z = z + simple(r - S @ z)
            return z + simple(r - S @ z)
        z = Y @ (Einv @ (Y.T @ r))
        return z + simple(r - S @ z)
    return S, P, (kv, kp)


if __name__ == "__main__":
    N = int(sys.argv[1])
    nu = float(sys.argv[2])
    dre = len(sys.argv) > 3 and sys.argv[3] == "dre"
    pr = pb.ricc_problem(N, nu)
    MT = pr.M.T.tocsr()
    if dre:
        tau = float(np.diff(pb.get_tint(0.0, 1.0, 16, True)).max())
        calA = (-(0.5 * MT + tau * (pr.A.T + pr.Nc.T))).tocsr()
    else:
        calA = (-pr.A - pr.Nc).T.tocsr()
    rng = np.random.default_rng(1)
    b = np.r_[rng.standard_normal(pr.NV), np.zeros(pr.NP)]
    print("N = %d, nu = %g%s, n = %d" % (N, nu, " (DRE operator)" if dre else "", pr.NV + pr.NP))
    plist = [float(t) for t in os.environ.get("SHIFTS", "1,30,300,3000").split(",")]
    cases = [("mass only", 1.0, 0.0)] + [("p = %g" % p, -p, 1.0) for p in plist]
    sel = [a for a in sys.argv[3:] if a != "dre"] or ["base", "schur2", "schur2f"]
    for variant in sel:
        row = []
        t0 = time.time()
        for name, al, be in cases:
            Ap = (be * calA + al * MT).tocsr()
            kw = {}
            if variant.startswith("ap"):          # "ap8": base cycle with pressure aggregates of 8
                kw = dict(ap=int(variant[2:]))
            if variant.startswith("bs") and variant[2:].isdigit():          # "bs64": base cycle with 64-row smoother blocks
                kw = dict(bs=int(variant[2:]))
            if variant.startswith("pb"):          # "pb128": Schur blocks of 128 pressure rows, velocity blocks of 32
                kw = dict(pbs=int(variant[2:]))
            if variant.startswith("av"):          # "av8": base cycle with velocity aggregates of 8
                kw = dict(av=int(variant[2:]))
            if variant.startswith("cc"):
                kw = dict(cc=(MT, -al if be else 0.0, nu if be else 0.0, 1.0 / N))
                if not be:      # mass only: Ap = M, Schur inverse = [J Mb^-1 J^T]_b^-1 (as base)
                    kw = {}
            if "+" in variant or variant in ("sa", "sa2", "sa3", "sas"):  # "sa+pb256": tokens sa (smoothed prolongation), pbN, bsN, avN, apN
                kw = {}
                for tok in variant.split("+"):
                    if tok in ("sa", "sa2", "sa3", "sas"):
                        kw["sa"] = calA
                        if tok[2:].isdigit():
                            kw["sa_steps"] = int(tok[2:])
                        if tok == "sas":
                            kw["sa_shifted"] = True
                    elif tok in ("c32", "c32h", "c64", "b16", "m32", "mb2"):
                        pass
                    elif tok[:2] in ("pb", "bs", "av", "ap"):
                        kw[{"pb": "pbs", "bs": "bs", "av": "av", "ap": "ap"}[tok[:2]]] = int(tok[2:])
            keepv = any(t in ("c32", "c32h", "c64", "b16", "m32", "mb2") for t in variant.split("+"))
            S, P, kk = make_precond(Ap, pr.J, MT, variant if keepv else
                                    "base" if (kw and "cc" not in kw) or (variant.startswith("cc") and not kw) else variant, **kw)
            x, its = gmres_right(S, P, b)
            row.append(its)
        print("%-10s %s   coarse (kv, kp) = %s  (%.0f s)" % (variant, "  ".join("%s: %d" % (c[0], i) for c, i in zip(cases, row)), kk,
                                                          time.time() - t0), flush=True)
