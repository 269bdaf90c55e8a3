"""Matrix generator (stand-in for dolfin_navier_scipy, optcont_main.py:322-334)."""
import numpy as np
import pytest

from optconpy_amd import problems as pb


@pytest.mark.parametrize("N", [4, 15])
def test_sizes_and_symmetry(N):
    sm = pb.stokes_system(N, nu=0.5)
    NV, NP = pb.drivcav_sizes(N)
    assert sm["M"].shape == (NV, NV) and sm["J"].shape == (NP, NV)
    assert abs(sm["M"] - sm["M"].T).max() < 1e-15
    assert abs(sm["A"] - sm["A"].T).max() < 1e-13
    # NV = 2(2N-1)^2, NP = (N+1)^2 - 1
    assert NV == 2 * (2 * N - 1) ** 2 and NP == (N + 1) ** 2 - 1


def test_appendix_a_nnz():
    sm = pb.stokes_system(15, nu=1.0)
    Nc = pb.convection_matrix(15)
    assert sm["M"].nnz == 17926          # SURVEY.md Appendix A, row N=15
    assert (sm["A"] + Nc).nnz == 35852


def test_oseen_part_is_skew():
    No = pb.convection_matrix(8, newton_term=False)
    # skew up to the quadrature error of the (non-polynomial) vortex field
    assert abs(No + No.T).max() < 1e-5 * abs(No).max()


def test_orderings_are_permutations():
    a = pb.stokes_system(5, nu=1.0, ordering="component")
    b = pb.stokes_system(5, nu=1.0, ordering="interleaved")
    ni = a["NV"] // 2
    perm = np.r_[2 * np.arange(ni), 2 * np.arange(ni) + 1]   # component -> interleaved index
    assert abs(b["M"][perm][:, perm] - a["M"]).max() < 1e-15
    assert abs(b["J"][:, perm] - a["J"]).max() < 1e-15


def test_time_mesh_matches_reference_formula():
    t = pb.get_tint(0.0, 2.0, 8, True)     # optcont_main.py:141-150
    assert t[0] == 0.0 and abs(t[-1] - 2.0) < 1e-15 and np.all(np.diff(t) > 0)
    assert abs(t[4] - 1.0) < 1e-15
