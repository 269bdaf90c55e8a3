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
