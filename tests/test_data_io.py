"""On-disk formats (reference: dou.save_npa/load_npa/save_spa/load_spa call sites)."""
import numpy as np
import pytest

from optconpy_amd import data_io as dio, problems as pb


def test_roundtrips_are_bit_exact(tmp_path):
    pr = pb.ricc_problem(5, 0.1, NU=2, NY=2)
    f = str(tmp_path / "mat")
    dio.save_spa(pr.M.tocsc(), f)                     # csc in, csr out, same values
    M2 = dio.load_spa(f)
    assert (M2 != pr.M).nnz == 0 and M2.indices.dtype == np.int32
    Z = np.random.default_rng(0).standard_normal((pr.NV, 7))
    dio.save_npa(Z, f + "__Z")
    assert np.array_equal(dio.load_npa(f + "__Z"), Z)
    with pytest.raises(IOError):                      # the callers' "not computed yet" signal
        dio.load_npa(str(tmp_path / "missing__Z"))
    with pytest.raises(IOError):
        dio.load_spa(str(tmp_path / "missing"))


def test_problem_bundle(tmp_path):
    pr = pb.ricc_problem(4, 0.2, NU=2, NY=2)
    path = str(tmp_path / "prob.npz")
    dio.save_problem(pr, path)
    q = dio.load_problem(path)
    for k in ("M", "A", "J", "Nc", "b_mat", "mc_mat", "rmat"):
        assert (q[k] != pr[k]).nnz == 0, k
    assert (q.N, q.NV, q.NP) == (pr.N, pr.NV, pr.NP) and q.nu == pr.nu


def test_datastr_is_a_stable_key():
    a = dio.get_datastr(time=0.25, meshp=15, nu=0.1, Nts=8, data_prfx="x_")
    assert a == dio.get_datastr(time=0.25, meshp=15, nu=0.1, Nts=8, data_prfx="x_")
    assert a != dio.get_datastr(time=0.5, meshp=15, nu=0.1, Nts=8, data_prfx="x_")
