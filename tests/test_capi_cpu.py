"""C-ABI library: loads without a GPU, exports every declared symbol, host logic."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sps

from optconpy_amd import _lib, problems as pb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ricadi.h")).read()
    declared = set(re.findall(r"\b(ricadi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ricadi_ctx", "ricadi_opts", "ricadi_adi_params"}
    assert len(declared) >= 25
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "symbol missing from libricadi_hip.so: " + name
        assert name in _lib.SIGNATURES, "ctypes signature missing: " + name
    assert set(_lib.SIGNATURES) <= declared


def test_defaults_follow_reference():
    p = _lib.adi_params(None)          # optcont_main.py:122-131
    assert (p.adi_max_steps, p.nwtn_max_steps) == (200, 16)
    assert (p.adi_newZ_reltol, p.nwtn_upd_reltol, p.nwtn_upd_abstol) == (1e-8, 5e-8, 1e-7)
    p = _lib.adi_params(dict(adi_max_steps=7, unknown_key=1, verbose=True, ms=[-1.0]))
    assert p.adi_max_steps == 7 and p.verbose == 1


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        _lib.Context(0)


def test_host_aggregate_partition():
    sm = pb.stokes_system(6, nu=1.0)
    M = sm["M"]
    blk, nb = _lib.host_aggregate(M, 16)
    assert blk.min() == 0 and blk.max() == nb - 1
    cnt = np.bincount(blk)
    assert cnt.max() <= 16 and cnt.min() >= 1 and cnt.sum() == M.shape[0]
    # aggregates are connected in the graph of M and never mix the two
    # velocity components (M is block diagonal over components)
    half = M.shape[0] // 2
    for b in range(nb):
        idx = np.flatnonzero(blk == b)
        assert (idx < half).all() or (idx >= half).all()
        sub = M[idx][:, idx]
        ncomp, _ = sps.csgraph.connected_components(sub, directed=False)
        assert ncomp == 1
    # bsize 1 -> singletons; huge bsize -> one block per connected component
    assert _lib.host_aggregate(M, 1)[1] == M.shape[0]
    assert _lib.host_aggregate(M, 10 ** 6)[1] == 2


def test_host_cauchy():
    ps = np.array([-0.5, -2.0, -9.0, -40.0])
    rinv, c1 = _lib.host_cauchy(ps)
    C = -1.0 / (ps[:, None] + ps[None, :])
    assert np.allclose(rinv, np.triu(rinv))
    assert np.allclose(rinv @ rinv.T, np.linalg.inv(C), rtol=1e-9)
    assert np.allclose(C @ c1, np.ones(4), rtol=1e-9)
    with pytest.raises(RuntimeError):
        _lib.host_cauchy([-1.0, -1.0])      # repeated shift: C singular


def test_input_normalisation():
    a = sps.csc_matrix(np.array([[1.0, 0, 2], [0, 3, 0], [4, 0, 5]]))
    rp, ci, v, sh = _lib.as_csr(2.0 * a.T)      # csc, scaled, transposed (solve_dae_ric.py:85,156)
    assert rp.dtype == np.int32 and ci.dtype == np.int32 and v.dtype == np.float64
    assert np.allclose(sps.csr_matrix((v, ci, rp), shape=sh).toarray(), 2.0 * a.T.toarray())
    p = _lib.as_panel(np.arange(3.0))
    assert p.shape == (3, 1) and p.flags["C_CONTIGUOUS"]
    with pytest.raises(ValueError):
        _lib.as_panel(np.zeros((4, 2)), nrows=3)


def test_dropin_package_names():
    import sadptprj_riclyap_adi.lin_alg_utils as lau
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    for name in ("proj_alg_ric_newtonadi", "solve_proj_lyap_stein", "compress_Zsvd",
                 "get_mTzzTtb", "comp_proj_lyap_res_norm"):
        assert callable(getattr(pru, name))
    for name in ("solve_sadpnt_smw", "app_prj_via_sadpnt", "apply_massinv",
                 "apply_invsqrt_fromright", "apply_sqrt_fromright", "app_luinv_to_spmat",
                 "mm_dnssps"):
        assert callable(getattr(lau, name))
    # host-only helpers work without a GPU
    Ms = sps.csr_matrix(np.diag([4.0, 9.0]))
    assert np.allclose(lau.apply_invsqrt_fromright(Ms, np.eye(2)), np.diag([0.5, 1 / 3.0]))


def test_mirror_glue_functions_against_scipy():
    """The MIRROR's mm_dnssps / app_luinv_to_spmat (host glue of the drop-in, not the oracle's copies):
    `lau.mm_dnssps` for every mix of dense / sparse factors (optcont_main.py:232-236) and `lau.app_luinv_to_spmat`
    with a SuperLU `factorized` handle applied to a sparse J^T, as the reference's test builds its projector
    (tests/test_units_compfacres_compress.py:70-71)."""
    import scipy.sparse.linalg as spsla
    import sadptprj_riclyap_adi.lin_alg_utils as lau
    import optconpy_amd.lin_alg_utils as glau
    assert lau.mm_dnssps is glau.mm_dnssps and lau.app_luinv_to_spmat is glau.app_luinv_to_spmat
    rng = np.random.default_rng(5)
    A = rng.standard_normal((6, 4))
    Bs = sps.random(4, 5, density=0.5, random_state=1, format="csc")
    As = sps.csr_matrix(A * (np.abs(A) > 0.5))
    for X, Y in ((A, Bs.toarray()), (As, Bs), (A, Bs), (As, Bs.toarray())):
        Z = lau.mm_dnssps(X, Y)
        assert isinstance(Z, np.ndarray) and not sps.issparse(Z)
        Xd = X.toarray() if sps.issparse(X) else X
        Yd = Y.toarray() if sps.issparse(Y) else Y
        assert np.allclose(Z, Xd @ Yd, rtol=1e-14, atol=1e-14)
    M = (sps.random(9, 9, density=0.3, random_state=2) + 5.0 * sps.eye(9)).tocsc()
    J = sps.random(3, 9, density=0.5, random_state=3, format="csr")
    Mlu = spsla.factorized(M)
    MinvJt = lau.app_luinv_to_spmat(Mlu, J.T)
    assert MinvJt.shape == (9, 3) and isinstance(MinvJt, np.ndarray)
    assert np.allclose(M @ MinvJt, J.T.toarray(), atol=1e-12)
    assert np.allclose(lau.app_luinv_to_spmat(Mlu, J.T.toarray()), MinvJt)         # dense input as well


def test_operator_cache_key_is_exact():
    """The resident operator / the remembered CSR conversion are reused only for IDENTICAL matrices: the key is a
    128-bit hash of the index and value bytes (backend.content_hash), not sums -- an in-place edit that
    preserves every sum (two values swapped, a sign-symmetric change) is seen."""
    from optconpy_amd import backend, proj_ric_utils as gpru
    A = sps.random(40, 40, density=0.2, random_state=7, format="csr") + sps.eye(40, format="csr")
    M = sps.eye(40, format="csr") * 2.0
    k0 = backend._fingerprint(A)
    assert backend._fingerprint(A.copy()) == k0 and backend._fingerprint(A.tocsc()) == k0
    a1, e1 = gpru._orient(A, M, False)
    a2, e2 = gpru._orient(A, M, False)
    assert a2 is a1 and e2 is e1                                    # remembered for the same operands
    i, j = 3, 11
    A.data[i], A.data[j] = A.data[j], A.data[i]                     # sum, |sum|, nnz, indices all unchanged
    assert A.data[i] != A.data[j]
    assert backend._fingerprint(A) != k0
    a3, e3 = gpru._orient(A, M, False)
    assert a3 is not a1
    assert np.allclose(a3.toarray(), A.T.toarray())
    B = A.copy()
    B.indices[[0, 1]] = B.indices[[1, 0]]                           # same multiset of column indices per row
    assert backend._fingerprint(B) != backend._fingerprint(A)


def test_cauchy_data_closed_form_and_conditioning_bound():
    """ricadi_host_cauchy: R^-1 and C^-1 1 of a sweep's Cauchy matrix in closed form (partial fractions of the ADI
    steps' rational functions).  Checked against the defining identities in extended precision on the sets the
    workloads use -- also the worst admissible one (8 neighbours of the 32-shift list, cond(C) = 6e9) --, and the
    conditioning bound: 16 neighbours of that list (cond(C) = 4e13, where round 3's numerically factorised data were
    wrong by 1e-6 and the cfg3 gain by 1.3e-5) are refused, so that the drivers halve the sweep."""
    from optconpy_amd import problems as pb
    ld = np.longdouble
    for ps in (pb.logshifts(1.0, 3e3, 16), pb.logshifts(1.0, 3e3, 32)[:8], pb.logshifts(1.0, 3e3, 32)[8:16],
               pb.logshifts(1.0, 3e3, 128, interleave=True)[:16], [-5.0, -3.0, -2.0, -1.5, -1.3, -1.1, -1.0][:4]):
        g = len(ps)
        rinv, c1 = _lib.host_cauchy(ps)
        Ri = np.asarray(rinv, dtype=ld).reshape(g, g)
        assert np.allclose(np.tril(np.asarray(rinv).reshape(g, g), -1), 0.0)
        p = np.asarray(ps, dtype=ld)
        Cm = -1.0 / (p[:, None] + p[None, :])
        # R^-T C R^-1 = I and C (C^-1 1) = 1, evaluated in extended precision
        E = Ri.T @ Cm @ Ri - np.eye(g, dtype=ld)
        scale = float(np.abs(Ri).max()) ** 2 * float(np.abs(Cm).max())
        assert float(np.abs(E).max()) < 1e-13 * max(scale, 1.0)
        one = Cm @ np.asarray(c1, dtype=ld)
        assert float(np.abs(one - 1.0).max()) < 1e-13 * float(np.abs(np.asarray(c1)).max()) * float(np.abs(Cm).max())
    with pytest.raises((RuntimeError, ValueError)):
        _lib.host_cauchy(pb.logshifts(1.0, 3e3, 32)[:16])
    with pytest.raises((RuntimeError, ValueError)):
        _lib.host_cauchy([-1.0, -2.0, -1.0])


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof of the two structs that cross the C-ABI, three ways: a C probe
    compiled from include/ricadi.h, the library's own ricadi_sizeof_*(), and the ctypes
    mirrors in optconpy_amd/_lib.py.  (A mirror shorter than the library's struct makes the
    library read past the caller's buffer: the round-1 abort)"""
    import ctypes as C
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "ricadi.h")).read()

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return re.findall(r"\b(?:int|double)\s+(\w+)\s*;", body)

    names = {"ricadi_opts": fields("ricadi_opts"), "ricadi_adi_params": fields("ricadi_adi_params")}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ricadi.h"', 'int main(void){']
    for st, fl in names.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fl:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    lines.append('return 0;}')
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    probe = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    lib = _lib.load()
    mirrors = {"ricadi_opts": _lib.RicadiOpts, "ricadi_adi_params": _lib.RicadiAdiParams}
    sizes = {"ricadi_opts": lib.ricadi_sizeof_opts(), "ricadi_adi_params": lib.ricadi_sizeof_adi_params()}
    sig = dict(part.split(":") for part in lib.ricadi_struct_signature().decode().split(";"))
    for st in names:      # the library's signature string follows the header's declarations
        decl = re.search(r"typedef struct %s \{(.*?)\} %s;" % (st, st), hdr, re.S).group(1)
        decl = re.sub(r"/\*.*?\*/", "", decl, flags=re.S)
        assert sig[st] == "".join(t[0] for t in re.findall(r"\b(int|double)\s+\w+\s*;", decl)), st
    for st, fl in names.items():
        mir = mirrors[st]
        assert [f for f, _ in mir._fields_] == fl, st           # same fields, same order
        assert int(probe[st]) == C.sizeof(mir) == sizes[st], st
        for f in fl:
            assert int(probe[st + "." + f]) == getattr(mir, f).offset, (st, f)


def test_load_refuses_mismatched_struct(monkeypatch):
    """The handshake itself: a mirror of another size must not get past load()."""
    import ctypes as C

    class Short(C.Structure):
        _fields_ = _lib.RicadiAdiParams._fields_[:-1]
    full = _lib.RicadiAdiParams
    monkeypatch.setattr(_lib, "RicadiAdiParams", Short)
    monkeypatch.setattr(_lib, "_lib", None)
    # dropping the trailing int leaves sizeof unchanged (tail padding): the signature sees it
    assert C.sizeof(Short) == C.sizeof(full)
    with pytest.raises(RuntimeError, match="ricadi_adi_params"):
        _lib.load()
    monkeypatch.undo()
    _lib._lib = None
    _lib.load()


def test_load_refuses_other_abi_version(monkeypatch):
    """The stats arrays' lengths are part of the ABI but not of the struct handshake: a library of another
    ricadi_version() must not get past load() either."""
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.load()
    monkeypatch.undo()
    _lib._lib = None
    _lib.load()


def test_host_deal_shift_owners():
    """Dealing of an ADI shift list to ranks (the shift-parallel sweeps): deterministic, every
    rank used, per-rank batches bounded, and the slow (small |p|) shifts are not stacked on one rank."""
    ms = -np.logspace(0.0, np.log10(3e3), 16)
    assert (_lib.host_deal(ms, 1) == 0).all()
    for world in (2, 4, 8, 16):
        own = _lib.host_deal(ms, world)
        assert (own == _lib.host_deal(ms, world)).all()
        cnt = np.bincount(own, minlength=world)
        assert cnt.min() >= 1 and cnt.max() <= -(-16 // world) + 1
        # the `world` slowest shifts (smallest |p|) all have different owners
        slow = np.argsort(-ms)[:world]
        assert len(set(own[slow])) == world
    # the order of the list does not matter for who shares a rank with whom
    perm = np.random.default_rng(0).permutation(16)
    a, b = _lib.host_deal(ms, 4), _lib.host_deal(ms[perm], 4)
    groups = lambda o, idx: sorted(tuple(sorted(np.round(idx[o == r], 9))) for r in range(4))
    assert groups(a, ms) == groups(b, ms[perm])
    with pytest.raises(ValueError):
        _lib.host_deal([1.0, -1.0], 2)           # positive shift


def test_host_sa_criterion_on_the_operator_kinds():
    """ricadi_host_sa_criterion (host logic of the round-3 smoothed aggregation): on for the diffusion-dominated NSE
    operator and the DRE operator of solve_dae_ric.py:147, off for a mass matrix (lau.app_prj_via_sadpnt hands one over
    as the operator, optcont_main.py:405-408) and for a convection-dominated operator."""
    from optconpy_amd import _lib, problems as pb
    pr = pb.ricc_problem(12, 0.1)
    calA = (-pr.A - pr.Nc).T.tocsr()
    on, rs, sk = _lib.host_sa_criterion(calA)
    assert on and 0.0 <= rs < 0.15 and 0.0 <= sk < 0.7, (on, rs, sk)
    dre = (-(0.5 * pr.M.T + 0.1 * (pr.A + pr.Nc).T)).tocsr()
    assert _lib.host_sa_criterion(dre)[0]
    on_m, rs_m, sk_m = _lib.host_sa_criterion(pr.M.T.tocsr())
    assert not on_m and rs_m > 1.0 and sk_m == -1.0
    pc = pb.ricc_problem(16, 0.01)                       # cell Peclet number ~ 1.6: the second test decides
    on_c, rs_c, sk_c = _lib.host_sa_criterion((-pc.A - pc.Nc).T.tocsr())
    assert not on_c and rs_c < 0.15 and sk_c > 0.7, (on_c, rs_c, sk_c)
    # the skew ratio is that of the matrix: recomputed with scipy
    K = (0.5 * (calA + calA.T)).tocsr()
    S = (0.5 * (calA - calA.T)).tocsr()
    import scipy.sparse as sps
    off = K - sps.diags(K.diagonal())
    assert abs(sk - np.abs(S.data).sum() / np.abs(off.data).sum()) < 1e-12


def test_host_plan_levels_keeps_two_levels_for_stiffness_dominated_operators():
    """ricadi_host_plan_levels (the hierarchy rule of ricadi_set_operator, host side): an operator the smoothed
    prolongation is made for keeps two levels up to 1.5 x coarse_max and grows its aggregates for it; a
    convection-dominated one goes to a child level as soon as the gentle child fits coarse_max (measured on the
    MI355X at n = 2e5 / 5e5: DESIGN.md section 3)."""
    pr = pb.ricc_problem(40, 0.05)
    calA, calE = (-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr()
    assert _lib.host_sa_criterion(calA)[0]
    full = _lib.host_plan_levels(calA, calE, pr.J)
    assert full["levels"] == 2 and full["smoothed"] and full["kc"] == full["kcv"] + full["kcp"] <= 4096
    # coarse_max below the base aggregates' coarse dimension: two levels, k within 1.5 x the cap, coarser aggregates
    for cm in (400, 150, 60):
        p = _lib.host_plan_levels(calA, calE, pr.J, coarse_max=cm)
        assert p["levels"] == 2 and p["smoothed"] and p["kc"] <= cm + cm // 2 and p["kc"] < full["kc"], (cm, p)
    # the same mesh, convection dominated: three levels at the aggregates whose gentle child fits
    pc = pb.ricc_problem(40, 0.0005)
    calAc = (-pc.A - pc.Nc).T.tocsr()
    assert not _lib.host_sa_criterion(calAc)[0]
    p = _lib.host_plan_levels(calAc, calE, pc.J, coarse_max=400)
    assert p["levels"] == 3 and not p["smoothed"] and p["kc"] > 400 and 0.55 * p["kcv"] + p["kcp"] <= 400, p
    # ... unless the caller asks for two levels only
    p2 = _lib.host_plan_levels(calAc, calE, pc.J, coarse_max=400, max_levels=2)
    assert p2["levels"] == 2 and p2["kc"] <= 400, p2
    # no coarse level at all
    assert _lib.host_plan_levels(calA, calE, pr.J, use_coarse=0)["levels"] == 1
