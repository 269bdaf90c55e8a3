"""The five identities the reference's one unit test pins
(/root/reference/tests/test_units_compfacres_compress.py:85-106), stated once in this
repo's own terms so that the CPU oracle and the HIP drop-in are held to the same checks.

With X = Z Z^T, P the discrete Leray projector of (M, J) and the projected residual
R(X) = P^T (F^T X M + M^T X F + W W^T) P:

  (1) ||M^T X M||_F can be had from the small Gram matrix (M^T Z)^T (M^T Z);
  (2) the factored residual norm `comp_proj_lyap_res_norm` equals ||R(X)||_F formed densely;
  (3) a compressed factor stays in the projected space: M^T Xc M = P^T (M^T Xc M) P;
  (4) compression keeps ||M^T X M||_F;
  (5) compression keeps the residual norm.

Added here (the reference test does not check it): the ADI run has actually converged.
"""
import numpy as np
import scipy.sparse.linalg as spsla


def leray_projector(M, J):
    """Dense P = I - M^-1 J^T (J M^-1 J^T)^-1 J (small N only)."""
    NV = M.shape[0]
    lu = spsla.splu(M.tocsc())
    MinvJt = lu.solve(J.T.toarray())
    schur = J @ MinvJt
    return np.eye(NV) - MinvJt @ np.linalg.solve(schur, J.toarray())


def dense_projected_residual(Z, F, M, W, P):
    """||P^T (F^T X M + M^T X F + W W^T) P||_F and ||P^T W W^T P||_F, dense."""
    L = F.T @ (Z @ (Z.T @ M.toarray()))          # F^T X M
    Wp = P.T @ W
    R = P.T @ (L + L.T) @ P + Wp @ Wp.T
    return np.linalg.norm(R), np.linalg.norm(Wp @ Wp.T)


def check_reference_identities(pru, M, J, F, W, adi_dict, thresh=1e-6):
    """Runs `pru.solve_proj_lyap_stein` + `pru.compress_Zsvd` + `pru.comp_proj_lyap_res_norm`
    (any module with the reference's names: the oracle or the HIP drop-in) and asserts the
    identities above.  Returns (Z, Z_compressed)."""
    Z = pru.solve_proj_lyap_stein(amat=F, mmat=M, jmat=J, wmat=W, adi_dict=adi_dict)["zfac"]
    P = leray_projector(M, J)
    dense_res, rhs_norm = dense_projected_residual(Z, F, M, W, P)
    factored_res = np.sqrt(abs(pru.comp_proj_lyap_res_norm(Z, F, M, W, J)))
    MtZ = M.T @ Z
    gram_norm = np.linalg.norm(MtZ.T @ MtZ)
    assert np.isclose(np.linalg.norm(MtZ @ MtZ.T), gram_norm, rtol=1e-5, atol=1e-8)       # (1)
    assert abs(dense_res - factored_res) <= 1e-5 * rhs_norm                                # (2)
    assert dense_res < 1e-6 * rhs_norm                                                     # converged
    Zc = pru.compress_Zsvd(Z, k=None, thresh=thresh, shplot=True)
    MtZc = M.T @ Zc
    Xc = MtZc @ MtZc.T
    assert np.allclose(Xc, P.T @ Xc @ P)                                                   # (3)
    assert np.isclose(np.linalg.norm(MtZc.T @ MtZc), gram_norm, rtol=1e-5, atol=1e-8)      # (4)
    factored_res_c = np.sqrt(abs(pru.comp_proj_lyap_res_norm(Zc, F, M, W, J)))
    assert abs(factored_res_c - dense_res) <= 1e-5 * rhs_norm                              # (5)
    assert Zc.shape[1] < Z.shape[1]
    return Z, Zc


def dense_projected_are(calA, calE, J, B, W):
    """Independent dense solution of the projected algebraic Riccati equation

        cal_A X cal_E^T + cal_E X cal_A^T - cal_E X B B^T X cal_E^T + W W^T = 0,   X = Theta Xh Theta^T,

    on the divergence-free space ker(J) = range(Theta): the equation is restricted to an
    orthonormal basis Theta of ker(J) and handed to ``scipy.linalg.solve_continuous_are`` --
    no ADI, no Newton-Kleinman, no saddle-point solve, nothing shared with oracle/ or the HIP
    path.  Small N only.  Returns the NV x NV matrix X."""
    import scipy.linalg as sla
    Th = sla.null_space(J.toarray() if hasattr(J, "toarray") else np.asarray(J))
    dn = lambda a: a.toarray() if hasattr(a, "toarray") else np.asarray(a)
    Ah = Th.T @ dn(calA) @ Th
    Eh = Th.T @ dn(calE) @ Th
    Bh = Th.T @ dn(B)
    Wh = Th.T @ dn(W)
    # standard form in Y = Eh Xh Eh^T:  a^T Y + Y a - Y b b^T Y + q = 0,  a = (Ah Eh^-1)^T, b = Eh^-T Bh
    # (scipy's generalised-pencil path rejects these pencils after balancing)
    a = np.linalg.solve(Eh.T, Ah.T)
    b = np.linalg.solve(Eh.T, Bh)
    q = Wh @ Wh.T
    Y = sla.solve_continuous_are(a, b, q, np.eye(b.shape[1]))
    for _ in range(2):      # polish the Schur-method result: dense Newton steps (LAPACK Lyapunov solver)
        ac = a - b @ (b.T @ Y)
        Y = sla.solve_continuous_lyapunov(ac.T, -(q + Y @ b @ b.T @ Y))
        Y = 0.5 * (Y + Y.T)
    res = a.T @ Y + Y @ a - Y @ b @ b.T @ Y + q
    assert np.linalg.norm(res) < 1e-12 * max(np.linalg.norm(q), np.linalg.norm(a.T @ Y))
    Xh = np.linalg.solve(Eh, np.linalg.solve(Eh, Y).T).T
    return Th @ (0.5 * (Xh + Xh.T)) @ Th.T


def dre_step_inputs(pr, tau=0.05, seed=0, with_old=False):
    """Arguments of the reference's per-time-step call (/root/reference/solve_dae_ric.py:147-159)
    built from a small problem: ft_mat = -(M^T/2 + tau (A^T + N^T)), w_mat = [M^T Zc, sqrt(tau) C~^T],
    bmat = sqrt(tau) B~, z0 = Zc, mtxoldb = sqrt(tau) * (gain accumulated by earlier outer steps).
    Returns (kwargs for proj_alg_ric_newtonadi, dict of the dense pieces for the checker)."""
    import scipy.sparse as sps
    from oracle import lin_alg_utils as olau
    MT = pr.M.T.tocsr()
    ft = (-(0.5 * MT + tau * (pr.A.T + pr.Nc.T))).tocsr()
    mct = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tb = olau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
    tct = olau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
    Zc = np.sqrt(0.1) * olau.apply_massinv(pr.M, tct)             # terminal value (:100)
    wmat = np.hstack([MT @ Zc, np.sqrt(tau) * tct])
    kw = dict(mmat=MT, amat=ft, transposed=True, jmat=pr.J, bmat=np.sqrt(tau) * sps.csr_matrix(tb),
              wmat=wmat, z0=Zc)
    old = None
    if with_old:
        rng = np.random.default_rng(seed)
        # a gain-like NV x NU matrix in range(P^T): -M^T Y Y^T B~ for a random projected Y
        Y = 0.3 * olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=rng.standard_normal((pr.NV, 3)))
        old = -(MT @ (Y @ (Y.T @ tb)))
        kw["mtxoldb"] = np.sqrt(tau) * old
    return kw, dict(MT=MT, ft=ft, tb=tb, wmat=wmat, tau=tau, old=old)


def dense_dre_sweep_gains(pr, kw, tmesh):
    """Independent dense restatement of the backward implicit-Euler sweep of the differential Riccati equation
    (/root/reference/solve_dae_ric.py:100,122-189): per time step the projected algebraic Riccati equation of the
    step -- cal A = -(M^T/2 + tau (A + N(t))^T), cal E = M^T, B = sqrt(tau) B~, W W^T = M^T X_{k+1} M + tau C~^T C~ --
    is solved by :func:`dense_projected_are` (scipy's Schur method on ker J), starting from the terminal value
    X(T) = gamma M^-1 C~^T C~ M^-T.  Nothing is compressed, nothing is iterated.  Returns {t: -M^T X(t) B~}."""
    import scipy.linalg as sla
    from oracle import lin_alg_utils as olau
    MT = pr.M.T.tocsr()
    tct = olau.apply_invsqrt_fromright(kw["vmat"], kw["mcmat"].T, output="dense")
    tb = olau.apply_invsqrt_fromright(kw["rmat"], kw["bmat"], output="dense")
    Zc = np.sqrt(kw["gamma"]) * olau.apply_massinv(pr.M, tct)
    X = Zc @ Zc.T
    gains = {tmesh[-1]: -(MT @ (X @ tb))}
    for tk in range(len(tmesh) - 2, -1, -1):
        t = tmesh[tk]
        tau = tmesh[tk + 1] - t
        nmat, _ = kw["get_tdpart"](time=t)
        ft = -(0.5 * MT + tau * (pr.A.T + nmat.T))
        # factor of M^T X M + tau C~^T C~ (any factor does: only W W^T enters)
        Q = MT @ X @ MT.T + tau * (tct @ tct.T)
        ev, U = sla.eigh(0.5 * (Q + Q.T))
        keep = ev > 1e-15 * ev.max()
        W = U[:, keep] * np.sqrt(ev[keep])
        X = dense_projected_are(ft.toarray(), MT, pr.J, np.sqrt(tau) * tb, W)
        gains[t] = -(MT @ (X @ tb))
    return gains
