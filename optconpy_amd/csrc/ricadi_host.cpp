// ricadi_host.cpp -- host-side setup logic of libricadi_hip.so (no device code).
//
// Builds, once per operator, everything of the two-level preconditioner that
// does not depend on the ADI shift: the unified saddle-point sparsity pattern,
// the block-Jacobi partitions (greedy graph aggregation), the dense diagonal
// blocks of cal A and cal E, the aggregation coarse space and its Galerkin
// matrices.  The per-shift parts are linear combinations formed on the device.
// Nothing here follows reference code: the reference solves these systems with
// SuperLU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "ricadi_internal.h"

namespace ricadi {

HostCsr make_csr(int nrows, int ncols, const int32_t* rp, const int32_t* ci, const double* v) {
  HostCsr a;
  a.nrows = nrows;
  a.ncols = ncols;
  a.rp.assign(rp, rp + nrows + 1);
  const int nnz = rp[nrows];
  a.ci.assign(ci, ci + nnz);
  a.v.assign(v, v + nnz);
  sort_rows(a);
  return a;
}

void sort_rows(HostCsr& a) {
  std::vector<std::pair<int, double>> tmp;
  for (int i = 0; i < a.nrows; ++i) {
    const int b = a.rp[i], e = a.rp[i + 1];
    bool sorted = true;
    for (int k = b + 1; k < e; ++k)
      if (a.ci[k] < a.ci[k - 1]) { sorted = false; break; }
    if (sorted) continue;
    tmp.clear();
    for (int k = b; k < e; ++k) tmp.emplace_back(a.ci[k], a.v[k]);
    std::sort(tmp.begin(), tmp.end(),
              [](const std::pair<int, double>& x, const std::pair<int, double>& y) {
                return x.first < y.first;
              });
    for (int k = b; k < e; ++k) {
      a.ci[k] = tmp[k - b].first;
      a.v[k] = tmp[k - b].second;
    }
  }
}

HostCsr transpose(const HostCsr& a) {
  HostCsr t;
  t.nrows = a.ncols;
  t.ncols = a.nrows;
  t.rp.assign(t.nrows + 1, 0);
  for (size_t k = 0; k < a.nnz(); ++k) t.rp[a.ci[k] + 1]++;
  for (int i = 0; i < t.nrows; ++i) t.rp[i + 1] += t.rp[i];
  t.ci.resize(a.nnz());
  t.v.resize(a.nnz());
  std::vector<int> pos(t.rp.begin(), t.rp.end() - 1);
  for (int i = 0; i < a.nrows; ++i)
    for (int k = a.rp[i]; k < a.rp[i + 1]; ++k) {
      const int d = pos[a.ci[k]]++;
      t.ci[d] = i;
      t.v[d] = a.v[k];
    }
  return t;
}

// Greedy BFS aggregation: grow a block from each still-free seed until it holds
// bsize rows.  Deterministic (seeds in index order, neighbours in CSR order).
int aggregate(int n, const int* rp, const int* ci, int bsize, int* blk) {
  if (bsize < 1) bsize = 1;
  std::fill(blk, blk + n, -1);
  std::vector<int> members;
  members.reserve(bsize);
  int nb = 0;
  for (int seed = 0; seed < n; ++seed) {
    if (blk[seed] >= 0) continue;
    members.clear();
    members.push_back(seed);
    blk[seed] = nb;
    size_t head = 0;
    while (head < members.size() && (int)members.size() < bsize) {
      const int u = members[head++];
      for (int k = rp[u]; k < rp[u + 1] && (int)members.size() < bsize; ++k) {
        const int v = ci[k];
        if (v >= 0 && v < n && blk[v] < 0) {
          blk[v] = nb;
          members.push_back(v);
        }
      }
    }
    ++nb;
  }
  return nb;
}

static void lists_from_blocks(int n, const int* blk, int nb, std::vector<int>& ptr,
                              std::vector<int>& rows) {
  ptr.assign(nb + 1, 0);
  for (int i = 0; i < n; ++i) ptr[blk[i] + 1]++;
  for (int b = 0; b < nb; ++b) ptr[b + 1] += ptr[b];
  rows.resize(n);
  std::vector<int> pos(ptr.begin(), ptr.end() - 1);
  for (int i = 0; i < n; ++i) rows[pos[blk[i]]++] = i;
}

// Galerkin product R^T M C for aggregation maps: entry (i, j) of M goes to (rowmap[i], colmap[j]).
// rows_ptr / rows_list: the fine rows of every coarse row.
static HostCsr galerkin(const HostCsr& M, int nrow_c, const int* rows_ptr, const int* rows_list, int row_off,
                        const int* colmap, int ncol_c) {
  HostCsr out;
  out.nrows = nrow_c;
  out.ncols = ncol_c;
  out.rp.assign(1, 0);
  std::vector<int> where(ncol_c, -1);
  for (int a = 0; a < nrow_c; ++a) {
    const int r0 = (int)out.ci.size();
    for (int q = rows_ptr[a]; q < rows_ptr[a + 1]; ++q) {
      const int i = rows_list[q] - row_off;
      for (int k = M.rp[i]; k < M.rp[i + 1]; ++k) {
        const int cj = colmap[M.ci[k]];
        int at = where[cj];
        if (at < r0) {
          at = (int)out.ci.size();
          where[cj] = at;
          out.ci.push_back(cj);
          out.v.push_back(0.0);
        }
        out.v[at] += M.v[k];
      }
    }
    out.rp.push_back((int)out.ci.size());
  }
  sort_rows(out);
  return out;
}

// Is smoothing the velocity aggregates with a Jacobi step on sym(A) appropriate for this operator?
//  - only for a stiffness-like A -- constants nearly in the kernel of its symmetric part: row sums small against
//    the diagonal (NSE operator 0.01-0.06, DRE operator at n = 3e4 0.03; a mass matrix 1.6).  Smoothing the
//    aggregates with a mass-like matrix makes the cycle WORSE (numpy mirror, [[M, J^T],[J, 0]]: 54 -> 122
//    iterations at omega = 0.5, no convergence at 0.67);
//  - and only while the symmetric part dominates: the smoother says nothing about a convection-dominated operator.
//    gamma = sum |skew part| / sum |off-diagonal symmetric part| (~ 1.2 x the cell Peclet number): measured
//    slowest-shift iterations with / without smoothing at gamma = 0.10: 73 / 100, 0.23: 59 / 67, 0.52: 82 / 106 and
//    93 / 106, but 1.07: 120 / 107, 2.0: 262 / 217 and 291 / 226.
// rs / gamma return the two ratios (gamma = -1 when the first test already failed).
bool sa_criterion(const HostCsr& A, double& rs_out, double& gamma_out) {
  const int nv = A.nrows;
  double srs = 0.0, sdg = 0.0;
  std::vector<double> colsum(nv, 0.0);
  for (int i = 0; i < nv; ++i)
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) colsum[A.ci[k]] += A.v[k];
  for (int i = 0; i < nv; ++i) {
    double rsum = 0.0, dg = 0.0;
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) {
      rsum += A.v[k];
      if (A.ci[k] == i) dg += A.v[k];
    }
    srs += std::fabs(0.5 * (rsum + colsum[i]));
    sdg += std::fabs(dg);
  }
  rs_out = sdg > 0.0 ? srs / sdg : -1.0;
  gamma_out = -1.0;
  if (!(sdg > 0.0) || srs > 0.15 * sdg) return false;
  const HostCsr At0 = transpose(A);
  std::vector<double> w1(nv, 0.0), w2(nv, 0.0);
  double sk = 0.0, sy = 0.0;
  for (int i = 0; i < nv; ++i) {
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) w1[A.ci[k]] += 0.5 * A.v[k];
    for (int k = At0.rp[i]; k < At0.rp[i + 1]; ++k) w2[At0.ci[k]] += 0.5 * At0.v[k];
    auto visit = [&](int j) {
      if (w1[j] == 0.0 && w2[j] == 0.0) return;
      sk += std::fabs(w1[j] - w2[j]);
      if (j != i) sy += std::fabs(w1[j] + w2[j]);
      w1[j] = w2[j] = 0.0;
    };
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) visit(A.ci[k]);
    for (int k = At0.rp[i]; k < At0.rp[i + 1]; ++k) visit(At0.ci[k]);
  }
  gamma_out = sy > 0.0 ? sk / sy : 1e30;
  return gamma_out <= 0.7;
}

void build_setup(const HostCsr& A, const HostCsr& E, const HostCsr& J, const ricadi_opts& o,
                 HostSetup& hs, int max_levels, double sa_omega) {
  const int nv = A.nrows, np = J.nrows, n = nv + np;
  hs.nv = nv;
  hs.np = np;
  hs.n = n;
  HostCsr JT = transpose(J);

  // ---- unified saddle pattern -------------------------------------------
  hs.s_rp.assign(n + 1, 0);
  hs.s_ci.clear();
  hs.s_srcA.clear();
  hs.s_srcE.clear();
  hs.s_srcJ.clear();
  hs.dA.assign(nv, 0.0);
  hs.dE.assign(nv, 0.0);
  // velocity-velocity pattern kept aside for the graph work
  std::vector<int> vv_rp(nv + 1, 0), vv_ci;
  for (int i = 0; i < nv; ++i) {
    int a = A.rp[i], ae = A.rp[i + 1], e = E.rp[i], ee = E.rp[i + 1];
    while (a < ae || e < ee) {
      const int ca = a < ae ? A.ci[a] : INT32_MAX;
      const int ce = e < ee ? E.ci[e] : INT32_MAX;
      const int c = std::min(ca, ce);
      double va = 0.0, ve = 0.0;
      if (ca == c) va = A.v[a++];
      if (ce == c) ve = E.v[e++];
      hs.s_ci.push_back(c);
      hs.s_srcA.push_back(va);
      hs.s_srcE.push_back(ve);
      hs.s_srcJ.push_back(0.0);
      vv_ci.push_back(c);
      if (c == i) {
        hs.dA[i] = va;
        hs.dE[i] = ve;
      }
    }
    vv_rp[i + 1] = (int)vv_ci.size();
    for (int k = JT.rp[i]; k < JT.rp[i + 1]; ++k) {
      hs.s_ci.push_back(nv + JT.ci[k]);
      hs.s_srcA.push_back(0.0);
      hs.s_srcE.push_back(0.0);
      hs.s_srcJ.push_back(JT.v[k]);
    }
    hs.s_rp[i + 1] = (int)hs.s_ci.size();
  }
  for (int k = 0; k < np; ++k) {
    for (int q = J.rp[k]; q < J.rp[k + 1]; ++q) {
      hs.s_ci.push_back(J.ci[q]);
      hs.s_srcA.push_back(0.0);
      hs.s_srcE.push_back(0.0);
      hs.s_srcJ.push_back(J.v[q]);
    }
    hs.s_rp[nv + k + 1] = (int)hs.s_ci.size();
  }

  // ---- block-Jacobi partition of the velocity block -----------------------
  const int bs = (o.bj_block <= 16) ? 16 : (o.bj_block <= 32 ? 32 : 64);
  hs.bs = bs;
  std::vector<int> blk(nv);
  hs.nbv = aggregate(nv, vv_rp.data(), vv_ci.data(), bs, blk.data());
  lists_from_blocks(nv, blk.data(), hs.nbv, hs.bv_ptr, hs.bv_rows);
  std::vector<int> local(nv);
  for (int b = 0; b < hs.nbv; ++b)
    for (int k = hs.bv_ptr[b]; k < hs.bv_ptr[b + 1]; ++k) local[hs.bv_rows[k]] = k - hs.bv_ptr[b];
  hs.bv_A.assign((size_t)hs.nbv * bs * bs, 0.0);
  hs.bv_E.assign((size_t)hs.nbv * bs * bs, 0.0);
  for (int i = 0; i < nv; ++i) {
    const int b = blk[i];
    double* Ba = hs.bv_A.data() + (size_t)b * bs * bs + (size_t)local[i] * bs;
    double* Be = hs.bv_E.data() + (size_t)b * bs * bs + (size_t)local[i] * bs;
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
      if (blk[A.ci[k]] == b) Ba[local[A.ci[k]]] += A.v[k];
    for (int k = E.rp[i]; k < E.rp[i + 1]; ++k)
      if (blk[E.ci[k]] == b) Be[local[E.ci[k]]] += E.v[k];
  }

  // ---- pressure graph (pattern of J J^T) and its block partition ----------
  std::vector<int> pp_rp(np + 1, 0), pp_ci;
  {
    std::vector<int> mark(np, -1);
    for (int k = 0; k < np; ++k) {
      for (int q = J.rp[k]; q < J.rp[k + 1]; ++q) {
        const int j = J.ci[q];
        for (int t = JT.rp[j]; t < JT.rp[j + 1]; ++t) {
          const int k2 = JT.ci[t];
          if (mark[k2] != k) {
            mark[k2] = k;
            pp_ci.push_back(k2);
          }
        }
      }
      std::sort(pp_ci.begin() + pp_rp[k], pp_ci.end());
      pp_rp[k + 1] = (int)pp_ci.size();
    }
  }
  std::vector<int> pblk(std::max(np, 1));
  hs.nbp = np > 0 ? aggregate(np, pp_rp.data(), pp_ci.data(), bs, pblk.data()) : 0;
  lists_from_blocks(np, pblk.data(), hs.nbp, hs.bp_ptr, hs.bp_rows);

  // ---- dense J sub-blocks for the consistent SIMPLE Schur complement ---------
  // S_bb = sum_beta J_{b,beta} Ahat_beta^-1 J_{b,beta}^T needs, per pressure block b,
  // the bs x bs slices of J against every velocity block beta it touches.  J does
  // not depend on the shift, so the slices are extracted once.
  {
    hs.jd_ptr.assign(1, 0);
    hs.jd_vblk.clear();
    hs.jd_val.clear();
    std::vector<int> slot(std::max(hs.nbv, 1), -1), touched;
    for (int b = 0; b < hs.nbp; ++b) {
      touched.clear();
      const size_t base = hs.jd_vblk.size();
      for (int q = hs.bp_ptr[b]; q < hs.bp_ptr[b + 1]; ++q) {
        const int k = hs.bp_rows[q], il = q - hs.bp_ptr[b];
        for (int e = J.rp[k]; e < J.rp[k + 1]; ++e) {
          const int j = J.ci[e], vb = blk[j];
          if (slot[vb] < 0) {
            slot[vb] = (int)touched.size();
            touched.push_back(vb);
            hs.jd_vblk.push_back(vb);
            hs.jd_val.resize(hs.jd_val.size() + (size_t)bs * bs, 0.0);
          }
          hs.jd_val[(base + slot[vb]) * (size_t)bs * bs + (size_t)il * bs + local[j]] += J.v[e];
        }
      }
      for (int vb : touched) slot[vb] = -1;
      hs.jd_ptr.push_back((int)hs.jd_vblk.size());
    }
  }

  // ---- row blocks of the LDS-tiled SpMM ------------------------------------
  // Rows are visited aggregate by aggregate (compact mesh patches) and packed
  // greedily into blocks of <= 32 rows whose set of distinct columns stays
  // within kSbMaxCols, so that the x tile of a block fits the LDS budget.
  {
    int kSbMaxRows = 32, kSbMaxCols = 152;   // 152 x 16 x 8 B tiles: 8 workgroups per CU fit the 160 KB LDS
    std::vector<int> order;
    order.reserve(n);
    for (int q = 0; q < nv; ++q) order.push_back(hs.bv_rows[q]);
    for (int q = 0; q < np; ++q) order.push_back(nv + hs.bp_rows[q]);
    hs.sb_rowptr.assign(1, 0);
    hs.sb_rows.clear();
    hs.sb_rp.assign(1, 0);
    hs.sb_cptr.assign(1, 0);
    hs.sb_cols.clear();
    hs.sb_perm.clear();
    hs.sb_lidx.clear();
    hs.sb_max_cols = hs.sb_max_nnz = 0;
    std::vector<int> stamp(n, -1), pos(n, -1), cols, brows;
    int bid = 0;
    size_t at = 0;
    while (at < order.size()) {
      cols.clear();
      brows.clear();
      // a velocity block never continues into the pressure rows
      const bool vel = order[at] < nv;
      while (at < order.size() && (int)brows.size() < kSbMaxRows && (order[at] < nv) == vel) {
        const int row = order[at];
        int fresh = 0;
        for (int k = hs.s_rp[row]; k < hs.s_rp[row + 1]; ++k)
          if (stamp[hs.s_ci[k]] != bid) ++fresh;
        if (!brows.empty() && (int)cols.size() + fresh > kSbMaxCols) break;
        for (int k = hs.s_rp[row]; k < hs.s_rp[row + 1]; ++k) {
          const int c = hs.s_ci[k];
          if (stamp[c] != bid) {
            stamp[c] = bid;
            cols.push_back(c);
          }
        }
        brows.push_back(row);
        ++at;
      }
      std::sort(cols.begin(), cols.end());
      for (size_t j = 0; j < cols.size(); ++j) pos[cols[j]] = (int)j;
      // Rows of similar length next to each other: a wave of the kernel steps through the 16-entry chunks of FOUR
      // consecutive local rows together (DPP broadcasts need all lanes), i.e. through the LONGEST of the four;
      // sorted by (half-)chunk count the four rows of a wave-pass need the same number of steps almost everywhere
      // (P2 vertex / edge-midpoint rows differ by a factor two in their entry counts).  RICADI_SB_SORT=0: visit order.
      const bool sort_rows_by_len = true;
      if (sort_rows_by_len)
        std::stable_sort(brows.begin(), brows.end(), [&](int a, int b) {
          return (hs.s_rp[a + 1] - hs.s_rp[a] + 7) / 8 > (hs.s_rp[b + 1] - hs.s_rp[b] + 7) / 8;   // half chunks
        });
      const size_t nnz0 = hs.sb_perm.size();
      // Entry order within a row: the kernel's 32-lane halves pair the local rows (2j, 2j+1), and
      // their two ds_read_b64 of a step (16 columns = 128 B each) are conflict free iff the two tile
      // rows have opposite parity (LDS bank = (byte / 4) mod 64).  Even local rows therefore list
      // their even tile rows first, odd local rows their odd ones: the parities differ wherever both
      // rows are in their first or both in their second part.
      const bool parity_order = true;
      int ql = 0;
      for (int row : brows) {
        for (int pass = 0; pass < 2; ++pass)
          for (int k = hs.s_rp[row]; k < hs.s_rp[row + 1]; ++k) {
            const int l = pos[hs.s_ci[k]];
            const bool first = !parity_order || ((l & 1) == (ql & 1));
            if (first != (pass == 0)) continue;
            hs.sb_perm.push_back(k);
            hs.sb_lidx.push_back((uint16_t)l);
          }
        hs.sb_rows.push_back(row);
        hs.sb_rp.push_back((int)hs.sb_perm.size());
        ++ql;
      }
      for (int c : cols) hs.sb_cols.push_back(c);
      hs.sb_cptr.push_back((int)hs.sb_cols.size());
      hs.sb_rowptr.push_back((int)hs.sb_rows.size());
      hs.sb_max_cols = std::max(hs.sb_max_cols, (int)cols.size());
      hs.sb_max_nnz = std::max(hs.sb_max_nnz, (int)(hs.sb_perm.size() - nnz0));
      ++bid;
    }
    hs.sb_nblk = bid;
  }

  // ---- aggregation coarse space -------------------------------------------
  hs.kc = hs.kcv = hs.kcp = 0;
  hs.agg_ptr.assign(1, 0);
  hs.agg_rows.clear();
  hs.aggof.assign(n, 0);
  hs.E0.clear();
  hs.EM.clear();
  hs.EJ.clear();
  if (!o.use_coarse) return;
  // graph for velocity aggregates: pattern of cal E if it is a genuine
  // (mass-like) matrix -- keeps the components apart -- else the union pattern
  const bool e_graph = E.nnz() > (size_t)(2 * nv);
  // Without a mass-like cal E (lau.solve_sadpnt_smw hands over ONE matrix) the aggregates follow the union
  // pattern and mix the velocity components; a child level built on those stagnates (measured at n = 1e5:
  // relres 0.9 after 3000 iterations, two levels: 169) -- stay with two levels then.
  if (!e_graph) max_levels = 2;
  const int* g_rp = e_graph ? E.rp.data() : vv_rp.data();
  const int* g_ci = e_graph ? E.ci.data() : vv_ci.data();
  int av = std::max(1, o.agg_v), ap = std::max(1, o.agg_p);
  std::vector<int> va(nv), pa(std::max(np, 1));
  int kv = 0, kp = 0;
  // Is this an operator the smoothed prolongation is made for (stiffness-like, symmetric part dominant)?
  double sa_rs = -1.0, sa_gamma = -1.0;
  const bool stiff = sa_omega > 0.0 && np > 0 && sa_criterion(A, sa_rs, sa_gamma);
  for (int attempt = 0; attempt < 16; ++attempt) {
    // The coarse pressure aggregates must not coincide with the Schur
    // block-Jacobi blocks (same graph, same greedy rule, same size): with
    // identical partitions the multiplicative two-level cycle stagnates
    // (measured: N=40, ap == bs == 32 stalls at 1e-2, ap in {16,24,48,64}
    // converges in 137-182 iterations).
    if (ap == bs) ap = ap + ap / 2;
    kv = aggregate(nv, g_rp, g_ci, av, va.data());
    kp = np > 0 ? aggregate(np, pp_rp.data(), pp_ci.data(), ap, pa.data()) : 0;
    // Dense inverse of this level's coarse matrix whenever it fits coarse_max.  (Rounds 2-3 handed the coarse problem
    // to a child level from HALF of coarse_max on -- the per-shift inversion grows with k^3 and the child's matrix is
    // half as large -- tuned on the one-solve-per-shift cycle: cfg3, k 3 046 -> 1 658, 72 -> 79 iterations, 188 -> 160 ms
    // per 32-shift cycle.  The workload the library exists for solves every shift many times per setup: the cfg3
    // Newton step -- 200 ADI steps over 32 shifts -- takes 955 ms with the dense inverse against 1 003 ms with the
    // child level (round 4, same-call A/B), so the dense inverse wins wherever it fits.)
    // For a stiffness-dominated operator the child level is a poor stand-in at the small shifts (measured at n = 2e5,
    // nu = 0.05, shift 1, same aggregates (36, 54), k = 5 415: child 229 iterations, dense inverse 175, dense inverse
    // with the smoothed prolongation 118 -- and the smoothed coarse operator handed to a child: 150 at these aggregates
    // but 572 against 214 at (81, 121), so smoothing stays a two-level affair).  Such operators keep two levels up to
    // 1.5 x coarse_max and grow their aggregates up to (121, 182) for it: n = 5e5, (81, 121), k = 5 969: 181 -> 132
    // iterations per shift-solve, 9.99 -> 8.86 s per pass over 128 shifts with the per-shift inversions inside.  A
    // convection-dominated or mass-like operator (criterion false) is served as well by the child as by the inverse
    // (n = 1e5, nu = 0.0025: 182 vs 153 iterations at shift 1, equal from shift 50 on, 219 vs 381 ms per 16 shifts).
    const int direct_max = std::max(16, stiff ? o.coarse_max + o.coarse_max / 2 : o.coarse_max);
    if (kv + kp <= direct_max) break;
    const bool grow_first = stiff && av + av / 2 <= 128;
    // A third level, only where it can be GENTLE: the coarse problem of these aggregates goes to a
    // child level whose own aggregates are pairs of velocity aggregates and single pressure aggregates
    // (so that the child's two-level cycle is a near-exact solve).  Coarsening the child harder makes
    // its cycle -- one multiplicative coarse correction + one SIMPLE sweep, not a contraction -- too
    // poor a stand-in for the coarse solve: GMRES stagnates (measured at n = 5e5 with every tried
    // pair of level-2 aggregate sizes).  Larger problems therefore still grow the
    // aggregates of THIS level, but only until the gentle child fits (in steps of 1.5, not 2).
    if (max_levels > 2 && np > 0 && !grow_first && 0.55 * kv + kp <= std::max(16, o.coarse_max)) {
      hs.multilevel = true;
      break;
    }
    if (max_levels > 2 && np > 0) {
      av += av / 2;
      ap += ap / 2;
      continue;
    }
    av *= 2;
    ap *= 2;
  }
  hs.kcv = kv;
  hs.kcp = kp;
  hs.kc = kv + kp;
  const int kc = hs.kc;
  for (int i = 0; i < nv; ++i) hs.aggof[i] = va[i];
  for (int k = 0; k < np; ++k) hs.aggof[nv + k] = kv + pa[k];
  lists_from_blocks(n, hs.aggof.data(), kc, hs.agg_ptr, hs.agg_rows);
  // ---- prolongation P (rows): plain aggregation, or smoothed on the velocity rows ----------------
  hs.sa = sa_omega > 0.0 && !hs.multilevel && np > 0 && kv > 0;
  if (hs.sa) {
    hs.sa = stiff;
    if (o.verbose)
      fprintf(stderr, "[ricadi] smoothed aggregation %s: row sums / diagonal of sym(cal A) = %.3f, skew / symmetric "
              "off-diagonal mass %.3f\n", hs.sa ? "on" : "off", sa_rs, sa_gamma);
  }
  hs.p_rp.clear(); hs.p_ci.clear(); hs.p_v.clear();
  hs.pt_rp.clear(); hs.pt_ci.clear(); hs.pt_v.clear();
  hs.pd_rp.clear(); hs.pd_ci.clear(); hs.pd_v.clear();
  if (hs.sa) {
    const HostCsr At = transpose(A);
    // damping relative to the spectral radius of D^-1 K0 (power iteration): the prolongation smoother
    // I - omega D^-1 K0 must not amplify -- a mass-like cal A (lau.app_prj_via_sadpnt hands the mass matrix over
    // as the operator) has rho ~ 4 for P2 elements, and omega = 0.67 made GMRES fail there
    {
      std::vector<double> dg(nv, 0.0), x(nv), y(nv);
      for (int i = 0; i < nv; ++i)
        for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
          if (A.ci[k] == i) dg[i] += A.v[k];
      unsigned sd = 12345u;
      for (int i = 0; i < nv; ++i) {
        sd = sd * 1664525u + 1013904223u;
        x[i] = (double)(sd >> 8) / 16777216.0 - 0.5;
      }
      double rho = 0.0;
      for (int it = 0; it < 20; ++it) {
        double nx = 0.0, ny = 0.0;
        for (int i = 0; i < nv; ++i) {
          double t = 0.0;
          for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) t += A.v[k] * x[A.ci[k]];
          for (int k = At.rp[i]; k < At.rp[i + 1]; ++k) t += At.v[k] * x[At.ci[k]];
          y[i] = dg[i] != 0.0 ? 0.5 * t / dg[i] : 0.0;
          nx += x[i] * x[i];
          ny += y[i] * y[i];
        }
        rho = nx > 0.0 ? std::sqrt(ny / nx) : 0.0;
        const double sc = ny > 0.0 ? 1.0 / std::sqrt(ny) : 0.0;
        for (int i = 0; i < nv; ++i) x[i] = y[i] * sc;
      }
      if (rho > 2.0) sa_omega *= 2.0 / rho;
      if (o.verbose) fprintf(stderr, "[ricadi] smoothed aggregation: rho(D^-1 K0) ~ %.2f, omega %.3f\n", rho, sa_omega);
    }
    hs.p_rp.assign(1, 0);
    hs.pd_rp.assign(1, 0);
    std::vector<int> where(kc, -1);
    for (int i = 0; i < nv; ++i) {
      const int r0 = (int)hs.p_ci.size();
      auto add = [&](int a, double w) {
        int at = where[a];
        if (at < r0) {
          at = (int)hs.p_ci.size();
          where[a] = at;
          hs.p_ci.push_back(a);
          hs.p_v.push_back(0.0);
        }
        hs.p_v[at] += w;
      };
      add(va[i], 1.0);
      double d = 0.0;
      for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
        if (A.ci[k] == i) d += A.v[k];
      if (d != 0.0) {
        const double sc = -0.5 * sa_omega / d;         // K0 = (A + A^T) / 2
        for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) add(va[A.ci[k]], sc * A.v[k]);
        for (int k = At.rp[i]; k < At.rp[i + 1]; ++k) add(va[At.ci[k]], sc * At.v[k]);
      }
      hs.p_rp.push_back((int)hs.p_ci.size());
      for (int k = r0; k < (int)hs.p_ci.size(); ++k) {
        hs.pd_ci.push_back(hs.p_ci[k]);
        hs.pd_v.push_back(hs.p_v[k] - (hs.p_ci[k] == va[i] ? 1.0 : 0.0));
      }
      hs.pd_rp.push_back((int)hs.pd_ci.size());
    }
    for (int k = 0; k < np; ++k) {
      hs.p_ci.push_back(kv + pa[k]);
      hs.p_v.push_back(1.0);
      hs.p_rp.push_back((int)hs.p_ci.size());
    }
    // P^T by rows
    hs.pt_rp.assign(kc + 1, 0);
    for (int c : hs.p_ci) hs.pt_rp[c + 1]++;
    for (int a = 0; a < kc; ++a) hs.pt_rp[a + 1] += hs.pt_rp[a];
    hs.pt_ci.resize(hs.p_ci.size());
    hs.pt_v.resize(hs.p_ci.size());
    std::vector<int> pos(hs.pt_rp.begin(), hs.pt_rp.end() - 1);
    for (int i = 0; i < n; ++i)
      for (int k = hs.p_rp[i]; k < hs.p_rp[i + 1]; ++k) {
        const int at = pos[hs.p_ci[k]]++;
        hs.pt_ci[at] = i;
        hs.pt_v[at] = hs.p_v[k];
      }
  }
  // the entries of row j of P (plain aggregation: the single (aggof[j], 1))
  auto prow = [&](int j, const int*& ci, const double*& v) -> int {
    static const double one = 1.0;
    if (hs.sa) {
      ci = hs.p_ci.data() + hs.p_rp[j];
      v = hs.p_v.data() + hs.p_rp[j];
      return hs.p_rp[j + 1] - hs.p_rp[j];
    }
    ci = hs.aggof.data() + j;
    v = &one;
    return 1;
  };
  if (hs.multilevel) {
    hs.l1A = galerkin(A, kv, hs.agg_ptr.data(), hs.agg_rows.data(), 0, va.data(), kv);
    hs.l1E = galerkin(E, kv, hs.agg_ptr.data(), hs.agg_rows.data(), 0, va.data(), kv);
    hs.l1J = galerkin(J, kp, hs.agg_ptr.data() + kv, hs.agg_rows.data(), nv, va.data(), kv);
  } else {
  hs.E0.assign((size_t)kc * kc, 0.0);
  hs.EM.assign((size_t)kc * kc, 0.0);
  hs.EJ.assign((size_t)kc * kc, 0.0);
  for (int i = 0; i < nv; ++i) {
    const int *ri, *cj;
    const double *rw, *cw;
    const int nri = prow(i, ri, rw);
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) {
      const int ncj = prow(A.ci[k], cj, cw);
      for (int a = 0; a < nri; ++a)
        for (int b = 0; b < ncj; ++b) hs.E0[(size_t)ri[a] * kc + cj[b]] += rw[a] * A.v[k] * cw[b];
    }
    for (int k = E.rp[i]; k < E.rp[i + 1]; ++k) {
      const int ncj = prow(E.ci[k], cj, cw);
      for (int a = 0; a < nri; ++a)
        for (int b = 0; b < ncj; ++b) hs.EM[(size_t)ri[a] * kc + cj[b]] += rw[a] * E.v[k] * cw[b];
    }
  }
  for (int k = 0; k < np; ++k)
    for (int q = J.rp[k]; q < J.rp[k + 1]; ++q) {
      const int cp = kv + pa[k];
      const int* cj;
      const double* cw;
      const int ncj = prow(J.ci[q], cj, cw);
      for (int b = 0; b < ncj; ++b) {
        hs.EJ[(size_t)cp * kc + cj[b]] += J.v[q] * cw[b];
        hs.EJ[(size_t)cj[b] * kc + cp] += J.v[q] * cw[b];
      }
    }
  }
  // ---- prolongated operator S*Y (n x kc, sparse) -----------------------------
  // Row i of the unified saddle pattern with its columns mapped to their aggregates and
  // duplicates merged (a row touches ~6 aggregates instead of ~28 columns).  The
  // residual after the coarse correction, r - S (Y e), is then one short-row CSR SpMM
  // over the L2-resident coarse vector instead of a full saddle SpMM.
  hs.sy_rp.assign(1, 0);
  hs.sy_ci.clear();
  hs.sy_A.clear();
  hs.sy_E.clear();
  hs.sy_J.clear();
  {
    std::vector<int> where(kc, -1);
    for (int i = 0; i < n; ++i) {
      const int r0 = (int)hs.sy_ci.size();
      for (int k = hs.s_rp[i]; k < hs.s_rp[i + 1]; ++k) {
        const int* cj;
        const double* cw;
        const int ncj = prow(hs.s_ci[k], cj, cw);
        for (int b = 0; b < ncj; ++b) {
          const int a = cj[b];
          int at = where[a];
          if (at < r0) {              // not seen in this row yet
            at = (int)hs.sy_ci.size();
            where[a] = at;
            hs.sy_ci.push_back(a);
            hs.sy_A.push_back(0.0);
            hs.sy_E.push_back(0.0);
            hs.sy_J.push_back(0.0);
          }
          hs.sy_A[at] += hs.s_srcA[k] * cw[b];
          hs.sy_E[at] += hs.s_srcE[k] * cw[b];
          hs.sy_J[at] += hs.s_srcJ[k] * cw[b];
        }
      }
      hs.sy_rp.push_back((int)hs.sy_ci.size());
    }
  }
  // S*Y in the tile format of the LDS-tiled SpMM, on the SAME row blocks as S: per block
  // the distinct aggregates its rows touch (the LDS tile of coarse-vector rows) and per
  // entry the 16-bit tile row; values are gathered through sy_perm.
  hs.syb_rp.assign(1, 0);
  hs.syb_cptr.assign(1, 0);
  hs.syb_cols.clear();
  hs.syb_perm.clear();
  hs.syb_lidx.clear();
  hs.syb_max_cols = 0;
  {
    std::vector<int> pos(kc, -1), cols;
    for (int b = 0; b < hs.sb_nblk; ++b) {
      cols.clear();
      for (int q = hs.sb_rowptr[b]; q < hs.sb_rowptr[b + 1]; ++q) {
        const int row = hs.sb_rows[q];
        for (int k = hs.sy_rp[row]; k < hs.sy_rp[row + 1]; ++k)
          if (pos[hs.sy_ci[k]] < 0) {
            pos[hs.sy_ci[k]] = 0;
            cols.push_back(hs.sy_ci[k]);
          }
      }
      std::sort(cols.begin(), cols.end());
      for (size_t j = 0; j < cols.size(); ++j) pos[cols[j]] = (int)j;
      for (int q = hs.sb_rowptr[b]; q < hs.sb_rowptr[b + 1]; ++q) {
        const int row = hs.sb_rows[q];
        for (int k = hs.sy_rp[row]; k < hs.sy_rp[row + 1]; ++k) {
          hs.syb_perm.push_back(k);
          hs.syb_lidx.push_back((uint16_t)pos[hs.sy_ci[k]]);
        }
        hs.syb_rp.push_back((int)hs.syb_perm.size());
      }
      for (int c : cols) {
        hs.syb_cols.push_back(c);
        pos[c] = -1;
      }
      hs.syb_cptr.push_back((int)hs.syb_cols.size());
      hs.syb_max_cols = std::max(hs.syb_max_cols, (int)cols.size());
    }
  }
}

// Cauchy data of one shift-parallel ADI sweep:
//   C_ij = -1/(p_i + p_j)  (s.p.d. for distinct negative real shifts),
//   C = R^T R;  rinv = R^-1 (upper, row-major);  cinv1 = C^-1 * ones.
// Owner rank of every shift of the list (see ricadi_host_deal in include/ricadi.h).
int deal_shifts(const double* shifts, int ns, int world, int32_t* owner) {
  double lmin = 1e300, lmax = -1e300;
  for (int i = 0; i < ns; ++i) {
    if (!(shifts[i] < 0.0)) return RICADI_EINVAL;
    const double l = std::log(-shifts[i]);
    lmin = std::min(lmin, l);
    lmax = std::max(lmax, l);
  }
  // predicted GMRES iterations (relative): slowest at the smallest |p| (cfg2: ~110 at p = -1,
  // ~30-45 over the upper half of a 1 ... 3e3 list)
  std::vector<double> it(ns);
  for (int i = 0; i < ns; ++i) {
    const double t = lmax > lmin ? (std::log(-shifts[i]) - lmin) / (lmax - lmin) : 1.0;
    it[i] = 1.0 + 2.5 * (1.0 - t) * (1.0 - t);
  }
  std::vector<int> order(ns);
  for (int i = 0; i < ns; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return it[a] > it[b]; });
  const double a = 3.2, b = 1.0;   // latency floor : per-group slope of a lockstep iteration
  std::vector<double> mx(world, 0.0), sm(world, 0.0);
  std::vector<int> cnt(world, 0);
  const int cap = (ns + world - 1) / world + 1;   // keeps the per-rank batch (and the exchange slots) small
  for (int k = 0; k < ns; ++k) {
    const int i = order[k];
    int best = -1;
    double bt = 0.0;
    for (int r = 0; r < world; ++r) {
      if (cnt[r] >= cap) continue;
      const double t = a * std::max(mx[r], it[i]) + b * (sm[r] + it[i]);
      if (best < 0 || t < bt - 1e-12) {
        best = r;
        bt = t;
      }
    }
    owner[i] = best;
    mx[best] = std::max(mx[best], it[i]);
    sm[best] += it[i];
    ++cnt[best];
  }
  return RICADI_OK;
}

// Least squares through the normal equations with a rank-revealing (diagonally pivoted)
// Cholesky factorisation:  Y = argmin || b - B Y ||_F  given  Ghh = B^T B (h x h) and
// Ghb = B^T b (h x m), both row-major.  Columns of B whose pivot falls below rtol * the largest
// diagonal entry are left out (their rows of Y are zero).  Returns the rank used.
int gram_lstsq(int h, int m, const double* Ghh, const double* Ghb, double rtol, double* Y) {
  std::vector<double> L((size_t)h * h, 0.0), d(h);
  std::vector<int> piv(h);
  double dmax = 0.0;
  for (int i = 0; i < h; ++i) {
    d[i] = Ghh[(size_t)i * h + i];
    piv[i] = i;
    dmax = std::max(dmax, d[i]);
  }
  for (size_t i = 0; i < (size_t)h * m; ++i) Y[i] = 0.0;
  if (!(dmax > 0.0)) return 0;
  int r = 0;
  for (; r < h; ++r) {
    int best = r;
    for (int i = r + 1; i < h; ++i)
      if (d[piv[i]] > d[piv[best]]) best = i;
    if (!(d[piv[best]] > rtol * dmax)) break;
    std::swap(piv[r], piv[best]);
    const int pr = piv[r];
    const double lrr = std::sqrt(d[pr]);
    L[(size_t)pr * h + r] = lrr;
    for (int i = r + 1; i < h; ++i) {
      const int pi = piv[i];
      double sum = Ghh[(size_t)pi * h + pr];
      for (int k = 0; k < r; ++k) sum -= L[(size_t)pi * h + k] * L[(size_t)pr * h + k];
      const double l = sum / lrr;
      L[(size_t)pi * h + r] = l;
      d[pi] -= l * l;
    }
  }
  // L (rows piv[0..r), r columns) L^T Y_P = Ghb_P
  std::vector<double> t((size_t)r);
  for (int c = 0; c < m; ++c) {
    for (int i = 0; i < r; ++i) {
      double sum = Ghb[(size_t)piv[i] * m + c];
      for (int k = 0; k < i; ++k) sum -= L[(size_t)piv[i] * h + k] * t[k];
      t[i] = sum / L[(size_t)piv[i] * h + i];
    }
    for (int i = r - 1; i >= 0; --i) {
      double sum = t[i];
      for (int k = i + 1; k < r; ++k) sum -= L[(size_t)piv[k] * h + i] * t[k];
      t[i] = sum / L[(size_t)piv[i] * h + i];
    }
    for (int i = 0; i < r; ++i) Y[(size_t)piv[i] * m + c] = t[i];
  }
  return r;
}

int gram_lstsq_scaled(int h, int m, std::vector<double>& Ghh, std::vector<double>& Ghb, double rtol,
                      std::vector<double>& Y) {
  std::vector<double> sc(h);
  for (int i = 0; i < h; ++i) {
    const double d = Ghh[(size_t)i * h + i];
    sc[i] = d > 0.0 ? 1.0 / std::sqrt(d) : 0.0;
  }
  for (int i = 0; i < h; ++i) {
    for (int j = 0; j < h; ++j) Ghh[(size_t)i * h + j] *= sc[i] * sc[j];
    for (int j = 0; j < m; ++j) Ghb[(size_t)i * m + j] *= sc[i];
  }
  Y.assign((size_t)h * m, 0.0);
  const int r = gram_lstsq(h, m, Ghh.data(), Ghb.data(), rtol, Y.data());
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < m; ++j) Y[(size_t)i * m + j] *= sc[i];
  return r;
}

// Cauchy data of an ADI sweep, C_ij = -1 / (p_i + p_j) = R^T R:  rinv = R^-1 (g x g, row major, upper
// triangular) and cinv1 = C^-1 1 -- in CLOSED FORM, not by a numerical Cholesky factorisation.  Column j of R^-1
// holds the partial-fraction coefficients of the rational function of ADI step j,
//   f_j(s) = sqrt(-2 p_j) / (s + p_j) * prod_{k<j} (s - p_k) / (s + p_k) = sum_{i<=j} c_ij / (s + p_i),
//   c_ij = sqrt(-2 p_j) prod_{k<j} (-p_i - p_k) / ( [i<j] (p_j - p_i) prod_{k<j, k!=i} (p_k - p_i) ),
// and C^-1 1 those of the residual's  prod_k (s - p_k) / (s + p_k) = 1 + sum_i d_i / (s + p_i),
//   d_i = -2 p_i prod_{k!=i} (p_i + p_k) / (p_i - p_k):
// products of sums and differences of the shifts, each entry accurate to a few ulp however ill conditioned C is.
// (Round 1-3 factorised C numerically: the 16 x 16 matrix of 16 NEIGHBOURS of a 32-shift list has condition 4e13,
// its computed R^-1 was wrong by 1e-6 relative, and the gain of the cfg3 Newton iteration came out 1.3e-5 off the
// oracle's -- whatever the GMRES tolerance -- while 16 shifts spread over the same range, cfg2, agreed to 1e-10.)
// Admissible sweeps: the recombination still amplifies the ERRORS OF THE SOLVES by ~ cond(R), so a sweep whose
// smallest pivot R_jj^2 / C_jj = prod_{k<j} ((p_j - p_k) / (p_j + p_k))^2 falls below 1e-6 (cond(R) >~ 1e5; cfg2:
// 2e-3) is refused: the drivers then halve the sweep width.
int cauchy_data(const double* shifts, int g, double* rinv, double* cinv1) {
  if (g < 1) return RICADI_EINVAL;
  typedef long double ld;
  std::vector<ld> p(g);
  for (int i = 0; i < g; ++i) {
    p[i] = (ld)shifts[i];
    if (!(shifts[i] < 0.0)) return RICADI_EINVAL;
    for (int k = 0; k < i; ++k)
      if (shifts[k] == shifts[i]) return RICADI_EBREAKDOWN;
  }
  for (int j = 0; j < g; ++j) {
    ld piv = 1.0L;
    for (int k = 0; k < j; ++k) {
      const ld q = (p[j] - p[k]) / (p[j] + p[k]);
      piv *= q * q;
    }
    if (!(piv > 1e-6L)) return RICADI_EBREAKDOWN;
  }
  for (int i = 0; i < g; ++i)
    for (int j = 0; j < g; ++j) rinv[(size_t)i * g + j] = 0.0;
  for (int j = 0; j < g; ++j)
    for (int i = 0; i <= j; ++i) {
      ld num = sqrtl(-2.0L * p[j]), den = 1.0L;
      for (int k = 0; k < j; ++k) num *= (-p[i] - p[k]);
      if (i < j) den *= (p[j] - p[i]);
      for (int k = 0; k < j; ++k)
        if (k != i) den *= (p[k] - p[i]);
      rinv[(size_t)i * g + j] = (double)(num / den);
    }
  for (int i = 0; i < g; ++i) {
    ld v = -2.0L * p[i];
    for (int k = 0; k < g; ++k)
      if (k != i) v *= (p[i] + p[k]) / (p[i] - p[k]);
    cinv1[i] = (double)v;
  }
  return RICADI_OK;
}

}  // namespace ricadi

extern "C" {

int ricadi_host_aggregate(int n, const int32_t* rowptr, const int32_t* col, int bsize,
                          int32_t* blk_out) {
  if (n < 0 || !rowptr || !col || !blk_out) {
    ricadi::set_error("ricadi_host_aggregate: bad argument");
    return RICADI_EINVAL;
  }
  return ricadi::aggregate(n, rowptr, col, bsize, blk_out);
}

int ricadi_host_sa_criterion(int nv, const int32_t* a_rp, const int32_t* a_ci, const double* a_v, double* rowsum_ratio_out,
                             double* skew_ratio_out, int* on_out) {
  if (nv < 1 || !a_rp || !a_ci || !a_v || !on_out) {
    ricadi::set_error("ricadi_host_sa_criterion: bad argument");
    return RICADI_EINVAL;
  }
  try {
    const ricadi::HostCsr A = ricadi::make_csr(nv, nv, a_rp, a_ci, a_v);
    double rs = -1.0, g = -1.0;
    *on_out = ricadi::sa_criterion(A, rs, g) ? 1 : 0;
    if (rowsum_ratio_out) *rowsum_ratio_out = rs;
    if (skew_ratio_out) *skew_ratio_out = g;
  } catch (...) {
    ricadi::set_error("ricadi_host_sa_criterion: exception");
    return RICADI_EINVAL;
  }
  return RICADI_OK;
}

int ricadi_host_plan_levels(int nv, int np, const int32_t* a_rp, const int32_t* a_ci, const double* a_v,
                            const int32_t* e_rp, const int32_t* e_ci, const double* e_v, const int32_t* j_rp,
                            const int32_t* j_ci, const double* j_v, const ricadi_opts* opts, int32_t* out) {
  if (nv < 1 || np < 0 || !a_rp || !a_ci || !a_v || !e_rp || !e_ci || !e_v || (np > 0 && (!j_rp || !j_ci || !j_v)) ||
      !opts || !out) {
    ricadi::set_error("ricadi_host_plan_levels: bad argument");
    return RICADI_EINVAL;
  }
  try {
    const ricadi::HostCsr A = ricadi::make_csr(nv, nv, a_rp, a_ci, a_v), E = ricadi::make_csr(nv, nv, e_rp, e_ci, e_v);
    const int32_t zero = 0;
    const ricadi::HostCsr J = np > 0 ? ricadi::make_csr(np, nv, j_rp, j_ci, j_v) : ricadi::make_csr(0, nv, &zero, &zero, a_v);
    ricadi::HostSetup hs;
    const double sa_omega = (np == 0 || opts->bj_block != 32) ? 0.0 : 0.5;
    ricadi::build_setup(A, E, J, *opts, hs, std::max(2, opts->max_levels), sa_omega);
    out[0] = hs.kc == 0 ? 1 : hs.multilevel ? 3 : 2;
    out[1] = hs.kc;
    out[2] = hs.kcv;
    out[3] = hs.kcp;
    out[4] = hs.sa ? 1 : 0;
  } catch (...) {
    ricadi::set_error("ricadi_host_plan_levels: exception");
    return RICADI_EINVAL;
  }
  return RICADI_OK;
}

int ricadi_host_deal(const double* shifts, int ns, int world, int32_t* owner_out) {
  if (!shifts || !owner_out || ns < 1 || world < 1) {
    ricadi::set_error("ricadi_host_deal: bad argument");
    return RICADI_EINVAL;
  }
  return ricadi::deal_shifts(shifts, ns, world, owner_out);
}

int ricadi_host_cauchy(const double* shifts, int g, double* rinv_out, double* cinv1_out) {
  if (!shifts || !rinv_out || !cinv1_out) {
    ricadi::set_error("ricadi_host_cauchy: bad argument");
    return RICADI_EINVAL;
  }
  int rc = ricadi::cauchy_data(shifts, g, rinv_out, cinv1_out);
  if (rc == RICADI_EBREAKDOWN)
    ricadi::set_error("ricadi_host_cauchy: Cauchy matrix not positive definite "
                      "(shifts must be distinct, negative and few)");
  return rc;
}

}  // extern "C"
