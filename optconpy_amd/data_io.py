"""On-disk formats on either side of the hot path.

The reference caches every expensive product through ``dolfin_navier_scipy``'s
``data_output_utils`` (``dou.save_npa`` / ``dou.load_npa`` for dense arrays,
``dou.save_spa`` / ``dou.load_spa`` for sparse ones: ``optcont_main.py:373-391``,
``solve_dae_ric.py:104-109,143-145,168-170``) under names built by
``get_datastr`` (``optcont_main.py:153-157``).  These helpers keep that calling
convention -- a missing file raises ``IOError``, which the callers use as the
"not computed yet" signal -- with plain ``.npy`` / ``.npz`` files, and add a
one-file bundle for a whole problem so that the CPU oracle and the MI355X path
read identical bytes.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sps

__all__ = ["save_npa", "load_npa", "save_spa", "load_spa", "get_datastr",
           "save_problem", "load_problem"]


def save_npa(v, fstring="notspecified"):
    np.save(fstring + ".npy", np.asarray(v))


def load_npa(fstring):
    path = fstring if fstring.endswith(".npy") else fstring + ".npy"
    if not os.path.exists(path):
        raise IOError("no data file " + path)
    return np.load(path)


def save_spa(sparray, fstring="notspecified"):
    m = sps.csr_matrix(sparray)
    m.sort_indices()
    np.savez(fstring + ".npz", indptr=m.indptr.astype(np.int32), indices=m.indices.astype(np.int32),
             data=m.data.astype(np.float64), shape=np.array(m.shape, dtype=np.int64))


def load_spa(fstring):
    path = fstring if fstring.endswith(".npz") else fstring + ".npz"
    if not os.path.exists(path):
        raise IOError("no data file " + path)
    z = np.load(path)
    return sps.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))


def get_datastr(time=None, meshp=None, nu=None, Nts=None, data_prfx="", **kw):
    """Cache-key string in the spirit of ``optcont_main.py:153-157``."""
    return (data_prfx + "time{0}_nu{1}_mesh{2}_Nts{3}".format(time, nu, meshp, Nts))


_SPARSE = ("M", "A", "J", "Nc", "b_mat", "mc_mat", "u_masmat", "y_masmat", "rmat")


def save_problem(pr, path):
    """One ``.npz`` with every matrix of a :func:`optconpy_amd.problems.ricc_problem` dict."""
    out = {}
    for k in _SPARSE:
        m = sps.csr_matrix(pr[k])
        m.sort_indices()
        out[k + "__indptr"] = m.indptr.astype(np.int32)
        out[k + "__indices"] = m.indices.astype(np.int32)
        out[k + "__data"] = m.data.astype(np.float64)
        out[k + "__shape"] = np.array(m.shape, dtype=np.int64)
    out["meta"] = np.array([pr["N"], pr["nu"], pr["NV"], pr["NP"]], dtype=np.float64)
    np.savez_compressed(path, **out)


def load_problem(path):
    from .problems import RicProblem
    if not os.path.exists(path):
        raise IOError("no data file " + path)
    z = np.load(path)
    pr = RicProblem()
    for k in _SPARSE:
        pr[k] = sps.csr_matrix((z[k + "__data"], z[k + "__indices"], z[k + "__indptr"]),
                               shape=tuple(z[k + "__shape"]))
    N, nu, NV, NP = z["meta"]
    pr.update(N=int(N), nu=float(nu), NV=int(NV), NP=int(NP))
    return pr
