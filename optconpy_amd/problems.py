"""Dolfin-free generator of driven-cavity-shaped saddle-point systems.

The reference obtains ``M, A, J`` and the convection linearisation from FEniCS
through ``dolfin_navier_scipy`` (``/root/reference/optcont_main.py:322-334``,
``:185-198``, ``:556-568``), which is not available offline.  This module
assembles the same kind of matrices -- P2/P1 Taylor-Hood on the unit square,
``N x N`` squares cut by right diagonals, homogeneous Dirichlet velocity dofs
condensed (``optcont_main.py:332-334``), the last pressure dof removed
(``optcont_main.py:327-329``) -- with numpy/scipy only, so that the oracle and
the HIP path consume identical CSR bytes.

Sizes: ``NV = 2 (2N-1)^2``, ``NP = (N+1)^2 - 1``.

Also here: distributed control / observation operators in the spirit of
``distr_control_fenics.cont_obs_utils`` (``optcont_main.py:372-391``), the time
mesh of ``optcont_main.py:141-150`` and the default solver parameters of
``optcont_main.py:122-135``.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps

__all__ = [
    "drivcav_sizes", "stokes_system", "convection_matrix", "vortex_field",
    "convection_from_vector", "nodal_interpolant", "convection_term",
    "control_observation", "get_tint", "default_nwtn_adi_dict", "RicProblem",
    "ricc_problem", "logshifts",
]


# ---------------------------------------------------------------------------
# reference element: vertices (0,0),(1,0),(0,1); P2 local order
# [v1, v2, v3, m23, m13, m12]
# ---------------------------------------------------------------------------
def _duffy_rule(npt=5):
    """Gauss rule on the reference triangle, exact to degree 2*npt-2."""
    g, w = np.polynomial.legendre.leggauss(npt)
    g = 0.5 * (g + 1.0)
    w = 0.5 * w
    u, v = np.meshgrid(g, g, indexing="ij")
    wu, wv = np.meshgrid(w, w, indexing="ij")
    x = u.ravel()
    y = (v * (1.0 - u)).ravel()
    wt = (wu * wv * (1.0 - u)).ravel()
    return x, y, wt


def _p1_ref(x, y):
    lam = np.stack([1.0 - x - y, x, y])                       # (3, q)
    dlam = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])   # (3, 2)
    return lam, dlam


def _p2_ref(x, y):
    lam, dlam = _p1_ref(x, y)
    l1, l2, l3 = lam
    phi = np.stack([l1 * (2 * l1 - 1), l2 * (2 * l2 - 1), l3 * (2 * l3 - 1),
                    4 * l2 * l3, 4 * l1 * l3, 4 * l1 * l2])    # (6, q)
    q = x.size
    dphi = np.empty((6, 2, q))
    for d in range(2):
        dphi[0, d] = (4 * l1 - 1) * dlam[0, d]
        dphi[1, d] = (4 * l2 - 1) * dlam[1, d]
        dphi[2, d] = (4 * l3 - 1) * dlam[2, d]
        dphi[3, d] = 4 * (l2 * dlam[2, d] + l3 * dlam[1, d])
        dphi[4, d] = 4 * (l1 * dlam[2, d] + l3 * dlam[0, d])
        dphi[5, d] = 4 * (l1 * dlam[1, d] + l2 * dlam[0, d])
    return phi, dphi


def drivcav_sizes(N):
    """(NV, NP) of the condensed system, SURVEY.md Appendix A."""
    return 2 * (2 * N - 1) ** 2, (N + 1) ** 2 - 1


class _Mesh:
    """Index bookkeeping for the structured Taylor-Hood mesh."""

    def __init__(self, N):
        self.N = N
        self.h = 1.0 / N
        nf = 2 * N + 1                      # fine (P2) nodes per direction
        self.nf = nf
        ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
        ii = ii.ravel()
        jj = jj.ravel()

        def fine(a, b):
            return a * nf + b

        def vert(i, j):
            return i * (N + 1) + j

        # vertices of the two triangle families, as (i, j) vertex indices
        fam = {
            "lo": ((ii, jj), (ii + 1, jj), (ii + 1, jj + 1)),
            "up": ((ii, jj), (ii + 1, jj + 1), (ii, jj + 1)),
        }
        self.p2 = {}
        self.p1 = {}
        self.orig = {}
        for key, (v1, v2, v3) in fam.items():
            a = [2 * v[0] for v in (v1, v2, v3)]
            b = [2 * v[1] for v in (v1, v2, v3)]
            loc = [fine(a[0], b[0]), fine(a[1], b[1]), fine(a[2], b[2]),
                   fine((a[1] + a[2]) // 2, (b[1] + b[2]) // 2),
                   fine((a[0] + a[2]) // 2, (b[0] + b[2]) // 2),
                   fine((a[0] + a[1]) // 2, (b[0] + b[1]) // 2)]
            self.p2[key] = np.stack(loc, axis=1)                      # (ne, 6)
            self.p1[key] = np.stack([vert(*v1), vert(*v2), vert(*v3)], axis=1)
            self.orig[key] = np.stack([v1[0] * self.h, v1[1] * self.h], axis=1)
        h = self.h
        # Jacobians  B = [p2-p1, p3-p1]
        self.B = {"lo": np.array([[h, h], [0.0, h]]),
                  "up": np.array([[h, 0.0], [h, h]])}
        # interior P2 nodes (homogeneous Dirichlet on the whole boundary)
        a, b = np.meshgrid(np.arange(nf), np.arange(nf), indexing="ij")
        inner = (a > 0) & (a < nf - 1) & (b > 0) & (b < nf - 1)
        self.inner_nodes = np.flatnonzero(inner.ravel())
        self.fine_xy = np.stack([a.ravel() * h / 2, b.ravel() * h / 2], axis=1)
        self.n_inner = self.inner_nodes.size
        glob2inner = -np.ones(nf * nf, dtype=np.int64)
        glob2inner[self.inner_nodes] = np.arange(self.n_inner)
        self.glob2inner = glob2inner


def _coo_to_csr(rows, cols, vals, shape):
    keep = (rows >= 0) & (cols >= 0)
    m = sps.coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=shape)
    m = m.tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return m


def _vel_index(mesh, comp, nodes, ordering):
    """Map (component, fine node) -> condensed velocity dof (or -1)."""
    inner = mesh.glob2inner[nodes]
    if ordering == "component":
        idx = comp * mesh.n_inner + inner
    elif ordering == "interleaved":
        idx = 2 * inner + comp
    else:
        raise ValueError("ordering must be 'component' or 'interleaved'")
    return np.where(inner >= 0, idx, -1)


def stokes_system(N, nu=1.0, ordering="component"):
    """Assemble ``M, A, J`` (condensed, pressure pinned).

    Counterpart of ``dts.get_stokessysmats`` + ``dts.condense_sysmatsbybcs``
    as used at ``optcont_main.py:322-334`` and
    ``tests/test_units_compfacres_compress.py:31-46``.  ``A`` already carries
    the factor ``nu``.
    """
    mesh = _Mesh(N)
    NV, NP = drivcav_sizes(N)
    x, y, wt = _duffy_rule(5)
    phi, dphi = _p2_ref(x, y)
    psi, _ = _p1_ref(x, y)
    rM, cM, vM = [], [], []
    rA, cA, vA = [], [], []
    rJ, cJ, vJ = [], [], []
    for key in ("lo", "up"):
        B = mesh.B[key]
        det = abs(np.linalg.det(B))
        Binv = np.linalg.inv(B)
        # physical gradients: grad = B^{-T} gradhat
        g = np.einsum("dk,ikq->idq", Binv.T, dphi)               # (6, 2, q)
        Mloc = np.einsum("iq,jq,q->ij", phi, phi, wt) * det
        Aloc = np.einsum("idq,jdq,q->ij", g, g, wt) * det * nu
        # J[k, (c, j)] = int psi_k d(phi_j)/dx_c   (div(u) * q)
        Jloc = np.einsum("kq,jcq,q->ckj", psi, g, wt) * det       # (2, 3, 6)
        p2 = mesh.p2[key]
        p1 = mesh.p1[key]
        ne = p2.shape[0]
        for c in range(2):
            dof = _vel_index(mesh, c, p2, ordering)               # (ne, 6)
            rr = np.repeat(dof[:, :, None], 6, axis=2).ravel()
            cc = np.repeat(dof[:, None, :], 6, axis=1).ravel()
            rM.append(rr); cM.append(cc); vM.append(np.tile(Mloc.ravel(), ne))
            rA.append(rr); cA.append(cc); vA.append(np.tile(Aloc.ravel(), ne))
            pr = np.repeat(p1[:, :, None], 6, axis=2).ravel()
            pc = np.repeat(dof[:, None, :], 3, axis=1).ravel()
            rJ.append(pr); cJ.append(pc); vJ.append(np.tile(Jloc[c].ravel(), ne))
    M = _coo_to_csr(np.concatenate(rM), np.concatenate(cM), np.concatenate(vM), (NV, NV))
    A = _coo_to_csr(np.concatenate(rA), np.concatenate(cA), np.concatenate(vA), (NV, NV))
    J = _coo_to_csr(np.concatenate(rJ), np.concatenate(cJ), np.concatenate(vJ), (NP + 1, NV))
    J = J[:-1, :].tocsr()       # remove the freedom in the pressure
    J.sort_indices()
    return dict(M=M, A=A, J=J, NV=NV, NP=NP, N=N, nu=nu, ordering=ordering)


def vortex_field(xy, amp=1.0):
    """Lid-driven-like vortex ``v = curl(psi)``, ``psi = sin^2(pi x) sin^2(pi y)``.

    Returns ``v`` (2, n) and ``grad v`` (2, 2, n) with ``gv[c, d] = dv_c/dx_d``.
    SURVEY.md section 8(d), "Common" inputs.
    """
    x, y = xy[..., 0], xy[..., 1]
    sx, cx = np.sin(np.pi * x), np.cos(np.pi * x)
    sy, cy = np.sin(np.pi * y), np.cos(np.pi * y)
    pi = np.pi
    # psi_y = 2 pi sx^2 sy cy ; psi_x = 2 pi sx cx sy^2
    v1 = amp * 2 * pi * sx ** 2 * sy * cy
    v2 = -amp * 2 * pi * sx * cx * sy ** 2
    c2x, c2y = cx ** 2 - sx ** 2, cy ** 2 - sy ** 2
    dv1dx = amp * 4 * pi ** 2 * sx * cx * sy * cy
    dv1dy = amp * 2 * pi ** 2 * sx ** 2 * c2y
    dv2dx = -amp * 2 * pi ** 2 * c2x * sy ** 2
    dv2dy = -dv1dx
    v = np.stack([v1, v2])
    gv = np.stack([np.stack([dv1dx, dv1dy]), np.stack([dv2dx, dv2dy])])
    return v, gv


def _analytic_fields(mesh, field, amp):
    """Per triangle family: the field and its gradient at the quadrature points, from a function."""
    x, y, _ = _duffy_rule(5)
    ref = np.stack([x, y], axis=0)                                # (2, q)

    def at(key):
        pts = mesh.orig[key][:, None, :] + (mesh.B[key] @ ref).T[None, :, :]   # (ne, q, 2)
        return field(pts, amp)                                    # (2,ne,q), (2,2,ne,q)
    return at


def _discrete_fields(mesh, vvec, ordering):
    """The same for a DISCRETE velocity: the P2 finite element function of the condensed dof vector
    ``vvec`` (boundary values zero: homogeneous Dirichlet dofs are condensed, ``optcont_main.py:332-334``)
    and its gradient, interpolated element by element at the quadrature points."""
    x, y, _ = _duffy_rule(5)
    phi, dphi = _p2_ref(x, y)
    vvec = np.asarray(vvec, dtype=float).ravel()
    vpad = np.concatenate([vvec, [0.0]])                          # index -1 -> boundary value 0

    def at(key):
        Binv = np.linalg.inv(mesh.B[key])
        g = np.einsum("dk,ikq->idq", Binv.T, dphi)                # (6, 2, q) physical gradients
        p2 = mesh.p2[key]
        v, gv = [], []
        for c in range(2):
            loc = vpad[_vel_index(mesh, c, p2, ordering)]         # (ne, 6) nodal values of component c
            v.append(loc @ phi)                                   # (ne, q)
            gv.append(np.einsum("ej,jdq->deq", loc, g))           # (2, ne, q): d v_c / d x_d
        return np.stack(v), np.stack(gv)
    return at


def _assemble_convection(mesh, fields, ordering, newton_term):
    NV, _ = drivcav_sizes(mesh.N)
    x, y, wt = _duffy_rule(5)
    phi, dphi = _p2_ref(x, y)
    rows, cols, vals = [], [], []
    for key in ("lo", "up"):
        B = mesh.B[key]
        det = abs(np.linalg.det(B))
        Binv = np.linalg.inv(B)
        g = np.einsum("dk,ikq->idq", Binv.T, dphi)                # (6, 2, q)
        v, gv = fields(key)                                       # (2,ne,q), (2,2,ne,q)
        # (v . grad phi_j) phi_i
        vgrad = np.einsum("ceq,jcq->ejq", v, g)                   # (ne, 6, q)
        adv = np.einsum("ejq,iq,q->eij", vgrad, phi, wt) * det    # (ne, 6, 6)
        p2 = mesh.p2[key]
        for ci in range(2):           # test component (row)
            for cj in range(2):       # trial component (col)
                loc = np.zeros_like(adv)
                if ci == cj:
                    loc = loc + adv
                if newton_term:
                    # (u . grad) v, component ci: sum_d u_d dv_ci/dx_d ; u_d = phi_j e_cj
                    loc = loc + np.einsum("eq,jq,iq,q->eij", gv[ci, cj], phi, phi, wt) * det
                elif ci != cj:
                    continue
                di = _vel_index(mesh, ci, p2, ordering)
                dj = _vel_index(mesh, cj, p2, ordering)
                rows.append(np.repeat(di[:, :, None], 6, axis=2).ravel())
                cols.append(np.repeat(dj[:, None, :], 6, axis=1).ravel())
                vals.append(loc.ravel())
    return _coo_to_csr(np.concatenate(rows), np.concatenate(cols),
                       np.concatenate(vals), (NV, NV))


def convection_matrix(N, field=vortex_field, amp=1.0, ordering="component",
                      newton_term=True):
    """Linearised convection ``N(v) u = (v.grad) u + (u.grad) v`` about an ANALYTIC field.

    Stand-in for ``snu.get_v_conv_conts`` (``optcont_main.py:185-198,
    456-462``); ``newton_term=False`` gives the Oseen (Picard) part only.
    """
    mesh = _Mesh(N)
    return _assemble_convection(mesh, _analytic_fields(mesh, field, amp), ordering, newton_term)


def convection_term(N, vvec, ordering="component"):
    """The nonlinear term itself, tested: ``H(v)_i = int ((v.grad) v) . phi_i`` for the discrete
    velocity ``vvec`` (NV x 1).  ``snu.get_v_conv_conts`` returns it as ``rhs_con``
    (``optcont_main.py:194-198``): Newton's linearisation ``(u.grad)u ~ N(v) u - H(v)``."""
    mesh = _Mesh(N)
    NV, _ = drivcav_sizes(N)
    x, y, wt = _duffy_rule(5)
    phi, _ = _p2_ref(x, y)
    fields = _discrete_fields(mesh, vvec, ordering)
    out = np.zeros(NV + 1)
    for key in ("lo", "up"):
        det = abs(np.linalg.det(mesh.B[key]))
        v, gv = fields(key)
        for c in range(2):
            adv = v[0] * gv[c, 0] + v[1] * gv[c, 1]                # (ne, q): (v . grad) v_c
            loc = np.einsum("eq,iq,q->ei", adv, phi, wt) * det     # (ne, 6)
            np.add.at(out, _vel_index(mesh, c, mesh.p2[key], ordering).ravel(), loc.ravel())
    return out[:NV].reshape(-1, 1)


def convection_from_vector(N, vvec, ordering="component", newton_term=True):
    """Convection linearisation about a DISCRETE velocity -- ``snu.get_v_conv_conts(prev_v=...)`` as
    ``get_convmats_rhs`` / ``get_tdpart`` call it with a stored velocity per time step
    (``optcont_main.py:185-198,556-568``).  Returns ``(convc_mat, rhs_con)``:

        convc_mat = N(v):  u -> (v.grad) u + (u.grad) v     (NV x NV, same element loops and quadrature as
                                                             :func:`convection_matrix`)
        rhs_con   = H(v) = N(v) v / 2                        (the constant of Newton's linearisation)

    The Dirichlet part ``rhsv_conbc`` of the reference vanishes: the boundary values are zero."""
    mesh = _Mesh(N)
    mat = _assemble_convection(mesh, _discrete_fields(mesh, vvec, ordering), ordering, newton_term)
    return mat, convection_term(N, vvec, ordering)


def nodal_interpolant(N, field=vortex_field, amp=1.0, ordering="component"):
    """Condensed dof vector (NV x 1) of the P2 nodal interpolant of an analytic field."""
    mesh = _Mesh(N)
    NV, _ = drivcav_sizes(N)
    v, _ = field(mesh.fine_xy[mesh.inner_nodes], amp)              # (2, n_inner)
    out = np.zeros(NV)
    for c in range(2):
        out[_vel_index(mesh, c, mesh.inner_nodes, ordering)] = v[c]
    return out.reshape(-1, 1)


def _hat_family(t, n):
    """n hat functions on [0,1] (partition of unity), evaluated at t."""
    if n == 1:
        return np.ones((1,) + t.shape)
    nodes = np.linspace(0.0, 1.0, n)
    hw = nodes[1] - nodes[0]
    out = np.maximum(0.0, 1.0 - np.abs(t[None] - nodes.reshape((-1,) + (1,) * t.ndim)) / hw)
    return out


def control_observation(N, M, NU=4, NY=4, ordering="component",
                        cdom=(0.4, 0.6, 0.2, 0.3), odom=(0.45, 0.55, 0.5, 0.7)):
    """Distributed control ``B`` and observation ``M_y C`` operators.

    Shapes follow ``cou.get_inp_opa`` / ``cou.get_mout_opa`` as consumed at
    ``optcont_main.py:372-391``: ``b_mat`` is NV x 2NU (sparse), ``mc_mat`` is
    2NY x NV (sparse), ``u_masmat``/``y_masmat`` are the 2NU / 2NY mass
    matrices of the 1D hat families (block diagonal over components).
    Default domains are the driven-cavity ones of ``optcont_main.py:30-33``.
    """
    mesh = _Mesh(N)
    xy = mesh.fine_xy[mesh.inner_nodes]

    def family(dom, n, along):
        x0, x1, y0, y1 = dom
        inside = (xy[:, 0] >= x0) & (xy[:, 0] <= x1) & (xy[:, 1] >= y0) & (xy[:, 1] <= y1)
        t = (xy[:, 0] - x0) / (x1 - x0) if along == 0 else (xy[:, 1] - y0) / (y1 - y0)
        f = _hat_family(np.clip(t, 0, 1), n) * inside[None]
        return f                                                  # (n, n_inner)

    def lift(f):
        n = f.shape[0]
        full = np.zeros((2 * n, 2 * mesh.n_inner))
        for c in range(2):
            idx = _vel_index(mesh, c, mesh.inner_nodes, ordering)
            full[c * n:(c + 1) * n, idx] = f
        return full

    def mass1d(n, length):
        if n == 1:
            return np.array([[length]])
        hw = length / (n - 1)
        m = np.zeros((n, n))
        for k in range(n - 1):
            m[k:k + 2, k:k + 2] += hw / 6.0 * np.array([[2.0, 1.0], [1.0, 2.0]])
        return m

    fb = lift(family(cdom, NU, 0))
    fc = lift(family(odom, NY, 1))
    b_mat = sps.csr_matrix(M @ fb.T)                  # NV x 2NU
    mc_mat = sps.csr_matrix((M @ fc.T).T)             # 2NY x NV
    b_mat.eliminate_zeros()
    mc_mat.eliminate_zeros()
    lu = (cdom[3] - cdom[2])
    ly = (odom[1] - odom[0])
    um = mass1d(NU, cdom[1] - cdom[0]) * lu
    ym = mass1d(NY, odom[3] - odom[2]) * ly
    u_masmat = sps.block_diag([um, um]).tocsr()
    y_masmat = sps.block_diag([ym, ym]).tocsr()
    return dict(b_mat=b_mat, mc_mat=mc_mat, u_masmat=u_masmat, y_masmat=y_masmat)


def get_tint(t0, tE, Nts, sqzmesh=True):
    """Time mesh of ``optcont_main.py:141-150``."""
    if sqzmesh:
        taux = np.linspace(-0.5 * np.pi, 0.5 * np.pi, int(Nts) + 1)
        taux = (np.sin(taux) + 1) * 0.5
        return (t0 + (tE - t0) * taux).flatten()
    return np.linspace(t0, tE, int(Nts) + 1).flatten()


def default_nwtn_adi_dict():
    """Defaults of ``optcont_main.py:122-131``."""
    return dict(adi_max_steps=200, adi_newZ_reltol=1e-8, nwtn_max_steps=16,
                nwtn_upd_reltol=5e-8, nwtn_upd_abstol=1e-7, verbose=False,
                full_upd_norm_check=False, check_lyap_res=False)


def logshifts(pmin, pmax, s, interleave=False):
    """``ms = -logspace(log10 pmin, log10 pmax, s)``.

    ``interleave``: the same values in the order ``0, k, 2k, ..., 1, k+1, ...`` with ``k = ceil(s / 16)``, so that
    any 16 consecutive entries of the list are spread over the whole range.  The ADI applies the shifts in list order
    (the reference takes ``ms`` as given, ``run_optcont.py:18-19``: an unsorted list); after every complete pass over
    the list the iterate is the same for every order, but the sweep form of the ADI -- G consecutive shifts solved
    against one residual factor, recombined with their G x G Cauchy matrix -- needs the shifts of a sweep well
    separated: 16 neighbours of a 128-shift list over 3.5 decades have a Cauchy matrix of condition > 1e13 (the
    library then shrinks the sweeps to 2 shifts), 16 interleaved ones are as far apart as the 16 shifts of cfg2."""
    ms = (-np.logspace(np.log10(pmin), np.log10(pmax), int(s))).tolist()
    if interleave and len(ms) > 16:
        k = -(-len(ms) // 16)
        ms = [ms[i] for r in range(k) for i in range(r, len(ms), k)]
    return ms


class RicProblem(dict):
    """Plain dict with attribute access; what the benchmarks and tests pass around."""
    __getattr__ = dict.__getitem__


def ricc_problem(N, nu, NU=4, NY=4, alphau=1e-2, conv_amp=1.0, ordering="component",
                 with_convection=True):
    """Everything the steady-state branch needs before the Newton-ADI call.

    Mirrors the preparation at ``optcont_main.py:322-425`` up to the projected,
    weighted operators; the projection / square-root steps themselves are the
    job of ``lin_alg_utils`` and are *not* done here (the callers do them through
    whichever implementation they test).
    """
    sm = stokes_system(N, nu=nu, ordering=ordering)
    M, A, J = sm["M"], sm["A"], sm["J"]
    if with_convection:
        Nc = convection_matrix(N, amp=conv_amp, ordering=ordering)
    else:
        Nc = sps.csr_matrix(M.shape)
    co = control_observation(N, M, NU=NU, NY=NY, ordering=ordering)
    return RicProblem(M=M, A=A, J=J, Nc=Nc, NV=sm["NV"], NP=sm["NP"], N=N, nu=nu,
                      rmat=alphau * co["u_masmat"], **co)
