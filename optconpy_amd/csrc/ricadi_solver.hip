// ricadi_solver.hip -- context, two-level preconditioner, panel GMRES, low-rank
// ADI, Newton-Kleinman, compression, gain; and the C-ABI of include/ricadi.h.
//
// New code (the reference has no native source).  The
// algorithms restate what the reference *calls* -- see include/ricadi.h for the
// reference call site behind each entry point.
#include <rccl/rccl.h>
#include <rocsolver/rocsolver.h>

#include <chrono>
#include <cmath>
#include <future>
#include <numeric>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <tuple>

#include "ricadi_internal.h"

namespace ricadi {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

struct HipError {
  std::string msg;
};
#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      throw HipError{std::string(#expr) + " : " + hipGetErrorString(e_)};                 \
  } while (0)
#define RBCHK(expr)                                                                       \
  do {                                                                                    \
    rocblas_status s_ = (expr);                                                           \
    if (s_ != rocblas_status_success)                                                     \
      throw HipError{std::string(#expr) + " : rocblas status " + std::to_string((int)s_)}; \
  } while (0)

template <class T>
struct DArr {
  T* p = nullptr;
  size_t n = 0;
  DArr() = default;
  DArr(const DArr&) = delete;
  DArr& operator=(const DArr&) = delete;
  ~DArr() { release(); }
  void release() {
    if (p) {
      (void)hipFree(p);
    }
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    if (count) HIPCHK(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
  void ensure(size_t count) {
    if (count > n) alloc(count);
  }
  void upload(const std::vector<T>& h, hipStream_t st) {
    alloc(h.size());
    if (!h.empty()) {
      HIPCHK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
  }
};

// Device scratch of a context: buffers handed out by take() come back with give() and are
// kept for the next request instead of going through hipMalloc / hipFree (a hipFree drains
// the device; the recompression, the block QR and the TSQR tree allocate dozens of
// temporaries per Newton step).  Everything a pool serves runs on ONE stream, so a buffer
// may be reused as soon as the host has released it: the kernels are ordered.
struct DevPool {
  struct Buf {
    void* p;
    size_t bytes;
  };
  std::vector<Buf> free_;
  size_t held = 0;
  ~DevPool() { trim(); }
  void trim() {
    for (Buf& b : free_) (void)hipFree(b.p);
    free_.clear();
    held = 0;
  }
  Buf take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (int i = 0; i < (int)free_.size(); ++i)
      if (free_[i].bytes >= bytes && free_[i].bytes <= 2 * bytes + 4096 &&
          (best < 0 || free_[i].bytes < free_[best].bytes))
        best = i;
    if (best >= 0) {
      Buf b = free_[best];
      free_.erase(free_.begin() + best);
      held -= b.bytes;
      return b;
    }
    Buf b{nullptr, bytes};
    if (hipMalloc(&b.p, bytes) != hipSuccess) {
      trim();                                   // give cached buffers back and retry once
      HIPCHK(hipMalloc(&b.p, bytes));
    }
    return b;
  }
  void give(Buf b) {
    if (!b.p) return;
    free_.push_back(b);
    held += b.bytes;
  }
};

// Temporary device array from a pool (scope bound, like DArr).
template <class T>
struct TArr {
  DevPool* pool = nullptr;
  DevPool::Buf b{nullptr, 0};
  T* p = nullptr;
  size_t n = 0;
  TArr() = default;
  explicit TArr(DevPool& pl) : pool(&pl) {}
  TArr(DevPool& pl, size_t count) : pool(&pl) { alloc(count); }
  TArr(const TArr&) = delete;
  TArr& operator=(const TArr&) = delete;
  TArr(TArr&& o) noexcept : pool(o.pool), b(o.b), p(o.p), n(o.n) {
    o.b = DevPool::Buf{nullptr, 0};
    o.p = nullptr;
    o.n = 0;
  }
  ~TArr() { release(); }
  void release() {
    if (pool && b.p) pool->give(b);
    b = DevPool::Buf{nullptr, 0};
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    if (count) {
      b = pool->take(count * sizeof(T));
      p = static_cast<T*>(b.p);
    }
    n = count;
  }
  void swap(TArr& o) {
    std::swap(pool, o.pool);
    std::swap(b, o.b);
    std::swap(p, o.p);
    std::swap(n, o.n);
  }
};

// Restores a value when the scope is left, also by an exception (a throw between a
// temporary change of the context's state and its restoration must not leak the change).
template <class T>
struct Restore {
  T& ref;
  T saved;
  explicit Restore(T& r) : ref(r), saved(r) {}
  ~Restore() { ref = saved; }
  Restore(const Restore&) = delete;
  Restore& operator=(const Restore&) = delete;
};

struct Tick {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double lap() {
    const auto t1 = std::chrono::steady_clock::now();
    const double s = std::chrono::duration<double>(t1 - t0).count();
    t0 = t1;
    return s;
  }
};

struct ShiftData {
  double alpha = 0, beta = 0;
  bool valid = false;   // contents computed for the current operator (buffers are kept when invalid)
  DArr<double> sval, svalb, syval, syvalb, bvinv, bpinv, einv;
  // FP32 copies of the inverses, the ones the preconditioner applies (a fixed linear
  // operator either way; halves its HBM traffic).  einvf is stored in 16 x 16 tiles
  // (dense_apply_tiled_kernel).  RICADI_PRECOND64=1 applies the FP64 originals instead.
  DArr<float> bvinvf, bpinvf, einvf;
  // G_b = Ahat_b^-1 J^T[rows_b, pcols_b] of the last velocity sweep (block_apply_rect_kernel)
  DArr<double> gtm;
  DArr<float> gtmf;
  // Ahat_b^-1 D_b (D_b: dense slice of S*Y) of the first velocity sweep with the coarse residual folded in
  DArr<double> adym;
  DArr<float> adymf;
  // Sherman-Morrison-Woodbury data for the current low-rank term (ctx->lr_epoch):
  // smw_w = S^-1 [U;0] (I - V^T S^-1 U)^-1, n x q
  DArr<double> smw_w;
  long smw_epoch = -1;
  ShiftData* sub = nullptr;   // the same shift on the child level (multilevel preconditioner)
  // recycled solves (ricadi_set_recycle): y with S(alpha,beta) y = b for the right-hand side panels of
  // the context's ring that carry the same serial number; n x w each
  struct RecY {
    long serial = -1;
    int w = 0;
    DArr<double> y;
  };
  std::vector<std::unique_ptr<RecY>> rec;
};

struct DevCsr {
  int nrows = 0;
  DArr<int> rp, ci;
  DArr<double> v;
  void upload(const HostCsr& h, hipStream_t st) {
    nrows = h.nrows;
    rp.upload(h.rp, st);
    ci.upload(h.ci, st);
    v.upload(h.v, st);
  }
};

// Where a dense stage runs: stream, rocBLAS / rocSOLVER handle bound to it, scratch pool and
// info word.  The context has two: its main one and an auxiliary one on a second stream, on
// which the in-ADI recompressions run concurrently with the next sweeps (a helper thread
// issues them: rocSOLVER's eigensolver is thousands of tiny launches, bound by the host).
struct Exec {
  hipStream_t st = nullptr;
  rocblas_handle rb = nullptr;
  DevPool* pool = nullptr;
  int* info = nullptr;
};

}  // namespace ricadi

using namespace ricadi;

struct ricadi_ctx {
  int dev = 0;
  hipStream_t st = nullptr;
  rocblas_handle rb = nullptr;
  // Multilevel preconditioner: when the coarse saddle problem of this level is too large for a
  // dense inverse (kc > coarse_max at the base aggregate sizes) it becomes the operator of a child
  // context (same stream and rocBLAS handle, borrowed), whose own preconditioner cycle -- sweep +
  // coarse correction, again dense or through a grandchild -- replaces the dense coarse apply.
  std::unique_ptr<ricadi_ctx> child;
  bool borrowed = false;      // st / rb belong to the parent level
  int levels = 2;             // levels this context may use (RICADI_LEVELS; 2 = two-level only)
  ricadi_opts opts;
  bool has_op = false;
  int nv = 0, np = 0, n = 0;
  int bs = 32, nbv = 0, nbp = 0, kc = 0;
  size_t snnz = 0;
  // operator
  DArr<int> s_rp, s_ci;
  DArr<double> srcA, srcE, srcJ;
  DevCsr A, E, J, JT;
  DArr<int> bv_ptr, bv_rows, bp_ptr, bp_rows, jd_ptr, jd_vblk;
  DArr<double> bvA, bvE, jd_val;
  DArr<int> agg_ptr, agg_rows, aggof;
  // last velocity sweep in rectangular form: per velocity block the pressure dofs its rows touch
  // and the dense slice of J^T over (block rows x those dofs); gt_ks = padded slice width
  // first velocity sweep with the residual of the coarse correction folded in: per velocity block
  // the coarse columns its S*Y rows touch and the dense slices of the three value sources
  bool ady_ok = false;
  int ady_ks = 0;
  DArr<int> cy_ptr, cy_cols;
  DArr<double> cy_dA, cy_dE, cy_dJ, cy_dT;
  // smoothed aggregation (HostSetup::sa): P^T by rows for the restriction; cy_dT = dense slices of P - Y
  bool sa = false;
  DArr<int> pt_rp, pt_ci;
  DArr<double> pt_v;
  bool gt_ok = false;
  int gt_ks = 0;
  DArr<int> gt_ptr, gt_cols;
  DArr<double> gt_jtd;
  DArr<double> E0, EM, EJ, ones;
  // prolongated operator S*Y (CSR, n x kc) for the residual after the coarse correction
  size_t synnz = 0;
  int sy_chunk = 16;          // 8 when its rows are short (mean <= 10 entries)
  DArr<int> sy_rp, sy_ci;
  DArr<double> sy_A, sy_E, sy_J;
  // tile format of S*Y on the saddle operator's row blocks (rows2 shared)
  int syb_max_cols = 0;
  bool syb_ok = false;
  DArr<int> syb_rp2, syb_cols2, syb_perm;
  DArr<uint16_t> syb_lidx;
  // LDS-tiled SpMM structure
  int sb_nblk = 0, sb_max_cols = 0, sb_max_nnz = 0;
  bool sb_ok = false;
  DArr<int> sb_perm;
  // block metadata padded to fixed strides (see spmm_blocked_kernel): rows2 [nblk][32],
  // rp2 [nblk][33], cols2 [nblk][sb_max_cols], colsm2 = cols2 through the aggregate map
  DArr<int> sb_rows2, sb_rp2, sb_cols2, sb_colsm2;
  DArr<uint16_t> sb_lidx;
  // the three value sources in tile order, for the multi-shift kernel (values of all shifts
  // from ONE read): saddle operator and prolongated operator
  DArr<double> sbAJ, sbE, sybAJ, sybE;
  DArr<uint16_t> sb_lidx_ms, syb_lidx_ms;
  bool ms_spmm = true;        // RICADI_MS_SPMM=0: one assembled value array per shift instead
  int ms_force = 0;           // RICADI_MS_SPMM=2: multi-shift kernel for every launch it can serve
  // low rank
  int q = 0;
  DArr<double> U, V, lrc, scratch;
  long lr_epoch = 0;          // bumped whenever U / V change
  bool smw = true;            // RICADI_SMW=0: keep the low-rank term inside the Krylov operator
  DArr<double> smw_rhs, smw_x, smw_cap;
  DArr<double> split_b, split_x;   // wide panels as sixteen-column groups (gmres_core_any)
  DArr<double> sweep_u, sweep_t, sweep_coef, sweep_part;   // ADI sweeps: the G solutions, a panel, coefficients, norm partials
  // per-shift data
  std::map<std::pair<double, double>, std::unique_ptr<ShiftData>> cache;
  // workspaces
  int wcols = 0, wrestart = 0;   // total columns (width x groups) and restart length the workspace holds
  DArr<double> basis, vcur, wv, zv, r2, tp, rc, ec, xs, bvec, pw1, pw2;
  DArr<float> basisf, zbasisf;   // zbasisf: Z_j = P^-1 v_j of the flexible GMRES, FP32
  bool flex = true;              // RICADI_FGMRES=0: plain right preconditioning (x += P^-1 (V y) per cycle)
  bool basis32 = true;
  bool basis16 = true;        // FP16-stored Krylov basis (default for n <= 2^21)
  bool precond32 = true;
  DArr<double> partial, h1, h2, H, cs, sn, g, scale, resid, yv, bnorm2, nrm2;
  DArr<int> flag, ipiv, info;
  DArr<double*> eptrs;
  DArr<double> gj_cb, gj_rp, gj_rb, gj_d;   // block Gauss-Jordan inverse of the coarse matrices
  DArr<double*> gj_ptrs;
  double* h_resid = nullptr;  // pinned, 4 slots of MAX_GROUPS*MAX_M: norms, rhs norms, two residual slots
  hipEvent_t ev_res[2] = {nullptr, nullptr};
  // factor
  DArr<double> Z;
  int zc = 0, zld = 0;
  // scratch of the dense stages (recompression, block QR, gain): see DevPool
  DevPool pool;
  // auxiliary execution resources for the asynchronous recompression (created on first use)
  hipStream_t st2 = nullptr;
  rocblas_handle rb2 = nullptr;
  DevPool pool2;
  DArr<int> info2;
  hipEvent_t ev_z = nullptr;
  // recycling of solved right-hand sides: ring of the last shared rhs panels (nv x w; pressure rows are zero)
  struct RecB {
    long serial = -1;
    int w = 0;
    DArr<double> b;
  };
  std::vector<std::unique_ptr<RecB>> rec_ring;
  // the same panels side by side (nv x 8 w_pan, slot i of the ring in columns [i w_pan, (i+1) w_pan)): the normal
  // equations of a recycled guess are then two GEMM launches instead of one per pair of stored panels
  DArr<double> rec_pan;
  int rec_pan_w = 0;
  long rec_serial = 0;
  int rec_depth = 0;          // depth in force for the next solves (the ADI drivers set it for their sweeps)
  int rec_user_depth = 0;     // ricadi_set_recycle: depth for direct solve calls
  // Sherman-Morrison-Woodbury: the low-rank factor U equals columns [lr_ucol, lr_ucol + q) of the right-hand
  // side of the next shared-rhs solve (first sweep of a Newton step: rhs = [W, K_k], U = K_k), so the
  // augmented columns [b, U] would be duplicates; -1 otherwise
  int lr_ucol = -1;
  // shift-parallel sweeps across processes (ricadi_set_exchange)
  int xrank = 0, xworld = 1;
  ricadi_allgather_fn xfn = nullptr;
  void* xuser = nullptr;
  double* xsend = nullptr;
  double* xrecv = nullptr;
  size_t xcap = 0;            // capacity of xsend in bytes
  // RCCL transport (ricadi_set_exchange_rccl): the all-gather is enqueued on the context's stream -- no host
  // round trip, no callback; the buffers are the library's own
  ncclComm_t xcomm = nullptr;
  bool xcomm_owned = false;
  bool xforce = false;        // a communicator of ONE rank still runs the exchange path (transport test)
  DArr<double> xsend_own, xrecv_own;
  long xcount = 0;            // collectives issued so far (ricadi_exchange_count)
  int coarse_route = -1;      // route the last batch of coarse inverses took (invert_dense_batch); -1: none yet
  int k1_variant = -1;        // saddle SpMM kernel of the last batched launch (saddle_spmm): 0 CSR, 1 tiled, 2 tiled multi-shift; +4: FP32 x
  // stats
  long total_iters = 0, total_solves = 0;
  long escalations = 0;       // solves repeated with wider storage of basis / preconditioner (safety net)
  int pc_stage = -1;          // >= 0: precond_apply issues only that stage (ricadi_time_kernel_dev)
  // wall-clock split of the drivers (RICADI_TIMING=1 prints it per Newton step; the stream is
  // drained at the section ends only in that mode)
  bool timing = false;
  double t_setup = 0, t_solve = 0, t_recomb = 0, t_compress = 0, t_updnorm = 0, t_proj = 0, t_gain = 0;
  double t_cyc = 0, t_iter = 0, t_guess = 0, t_smw = 0;   // inside t_solve: restart-cycle bookkeeping, Arnoldi iterations, recycling, SMW + checks

  ~ricadi_ctx() {
    if (h_resid) (void)hipHostFree(h_resid);
    if (xcomm && xcomm_owned) (void)ncclCommDestroy(xcomm);
    for (int i = 0; i < 2; ++i)
      if (ev_res[i]) (void)hipEventDestroy(ev_res[i]);
    child.reset();
    if (rb && !borrowed) rocblas_destroy_handle(rb);
    if (rb2) rocblas_destroy_handle(rb2);
    if (ev_z) (void)hipEventDestroy(ev_z);
    if (st2) (void)hipStreamDestroy(st2);
    if (st && !borrowed) (void)hipStreamDestroy(st);
  }
};

namespace ricadi {

// Width of the column groups a wide panel is solved in (0: the panel stays whole); see gmres_core_any.
static int wide_split_width(const ricadi_ctx* c, int m) {
  static const int off = getenv("RICADI_WIDE_SPLIT") && atoi(getenv("RICADI_WIDE_SPLIT")) == 0 ? 1 : 0;
  (void)c;
  return (!off && m > 32) ? 16 : 0;
}

// Workspace for batches of up to `groups` panels of width m (group-major: every
// buffer holds one slab per group; basis is vector-major, i.e. Krylov vector j of
// all groups is contiguous).
// `extra` columns per group are reserved on top of m (default: the low-rank width, for
// the augmented Sherman-Morrison-Woodbury solves) so that a nested, wider solve never
// reallocates buffers the caller has already filled.
static void ensure_work(ricadi_ctx* c, int m, int groups = 1, int extra = -1) {
  const int restart = c->opts.gmres_restart;
  if (extra < 0) extra = std::max(c->q, 0);
  if (c->child) {
    c->child->opts.gmres_restart = std::min(c->opts.gmres_restart, 4);   // its Krylov buffers are not used
    ensure_work(c->child.get(), m, groups, extra);
  }
  // every buffer scales with the total number of columns (m + extra) * groups
  int want = (m + extra) * groups;
  // a wide panel is solved as chunks of up to RICADI_MAX_GROUPS sixteen-column groups: the buffers must hold a
  // full chunk already NOW -- the caller's right-hand side lives in them (c->bvec) when the solve starts
  if (wide_split_width(c, m + extra)) want = std::max(want, 16 * RICADI_MAX_GROUPS);
  if (want <= c->wcols && restart == c->wrestart) return;
  const size_t gm = (size_t)std::max(want, c->wcols);
  const size_t nm = (size_t)c->n * gm;
  // Krylov basis: stored in FP16 by default (FP32 with RICADI_BASIS32=1, FP64 with
  // RICADI_BASIS64=1); ALL arithmetic stays FP64 -- the three passes over the basis per
  // iteration are the largest share of the HBM traffic.  The current vector is also
  // kept in FP64 (vcur, holding the same rounded values) for the operator /
  // preconditioner application, so the Arnoldi relation holds exactly for the stored
  // vectors; what the storage precision limits is the residual reduction one restart
  // cycle can deliver (~1e-3 for FP16, cycles gain ~1e-2), and every cycle starts from
  // the true FP64 residual.  Unit vectors of dimension n have entries ~ n^-1/2: FP16
  // (normal range from 6e-5) is used up to n = 2^21, FP32 beyond.
  c->basis32 = getenv("RICADI_BASIS64") == nullptr;
  c->basis16 = c->basis32 && getenv("RICADI_BASIS32") == nullptr && c->n <= (1 << 21);
  if (c->basis32) {
    c->basisf.alloc((size_t)(restart + 1) * nm);
    c->vcur.alloc(nm);
    c->basis.release();
  } else {
    c->basis.alloc((size_t)(restart + 1) * nm);
    c->basisf.release();
  }
  c->flex = true;      // flexible GMRES: Z_j = P^-1 v_j kept (FP32), x += Z y at the cycle end
  if (c->flex) c->zbasisf.alloc((size_t)restart * nm);
  else c->zbasisf.release();
  c->wv.alloc(nm);
  c->zv.alloc(nm);
  c->r2.alloc(nm);
  c->xs.alloc(nm);
  c->bvec.alloc(nm);
  c->pw1.alloc(nm);
  c->pw2.alloc(nm);
  c->tp.alloc((size_t)std::max(c->np, 1) * gm);
  c->rc.alloc((size_t)std::max(c->kc, 1) * gm);
  c->ec.alloc((size_t)std::max(c->kc, 1) * gm);
  c->partial.alloc((size_t)dots_num_blocks(c->n) * (restart + 2) * gm);
  c->h1.alloc((size_t)(restart + 2) * gm);
  c->h2.alloc((size_t)2 * (restart + 2) * gm);      // two buffers (atomic dot passes alternate between them)
  c->H.alloc(gm * (restart + 1) * restart);
  c->cs.alloc(gm * restart);
  c->sn.alloc(gm * restart);
  c->g.alloc(gm * (restart + 1));
  c->scale.alloc(gm);
  c->resid.alloc(2 * gm);          // two buffers (the fused update + Hessenberg launch alternates between them)
  c->yv.alloc((size_t)restart * gm);
  c->bnorm2.alloc(gm);
  c->nrm2.alloc(gm);
  c->lrc.alloc((size_t)64 * gm + 64);
  if (!c->h_resid) {
    HIPCHK(hipHostMalloc((void**)&c->h_resid,
                         sizeof(double) * 4 * RICADI_MAX_M * RICADI_MAX_GROUPS));
    for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev_res[i], hipEventDisableTiming));
  }
  c->wcols = (int)gm;
  c->wrestart = restart;
}

// ---- per-shift setup ---------------------------------------------------------
template <class T>
static void stable_alloc(DArr<T>& a, size_t n) {
  if (a.n != n) a.alloc(n);
}

// In-place inverses of nb (<= RICADI_MAX_GROUPS) dense k x k matrices (row-major, device pointers in hmats) by
// block Gauss-Jordan elimination without pivoting: per 128-row block three small kernels and two batched
// rocBLAS GEMMs (ricadi_kernels.hip).  Returns false if a diagonal block had a vanishing pivot (the matrices
// are garbage then; the caller assembles them again and takes the pivoted rocSOLVER route).
static bool gj_invert_batched(ricadi_ctx* c, double* const* hmats, int nb, int k) {
  hipStream_t st = c->st;
  const int NB = gj_block();
  const size_t pan = (size_t)k * NB;
  c->gj_cb.ensure(pan * nb);
  c->gj_rp.ensure(pan * nb);
  c->gj_rb.ensure(pan * nb);
  c->gj_d.ensure((size_t)NB * NB * nb);
  std::vector<double*> hp((size_t)5 * nb);
  for (int i = 0; i < nb; ++i) {
    hp[i] = hmats[i];
    hp[nb + i] = c->gj_cb.p + pan * i;
    hp[2 * nb + i] = c->gj_rp.p + pan * i;
    hp[3 * nb + i] = c->gj_rb.p + pan * i;
    hp[4 * nb + i] = c->gj_d.p + (size_t)NB * NB * i;
  }
  c->gj_ptrs.ensure((size_t)5 * nb);
  HIPCHK(hipMemcpyAsync(c->gj_ptrs.p, hp.data(), sizeof(double*) * hp.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(c->flag.p + 2, 0, sizeof(int), st));
  HIPCHK(hipStreamSynchronize(st));   // hp is a stack object
  double* const* dA = c->gj_ptrs.p;
  double* const* dCb = dA + nb;
  double* const* dRp = dA + 2 * nb;
  double* const* dRb = dA + 3 * nb;
  double* const* dD = dA + 4 * nb;
  const double one = 1.0, zero = 0.0, mone = -1.0;
  for (int k0 = 0; k0 < k; k0 += NB) {
    const int nbe = std::min(NB, k - k0);
    launch_gj_prep(st, nb, hmats, k, k0, nbe, c->gj_cb.p, c->gj_rp.p, c->gj_d.p);
    launch_gj_diag(st, nb, c->gj_d.p, nbe, c->flag.p + 2);
    // row-major Rb = D^-1 Rp  ==  column-major Rb^T = Rp^T (D^-1)^T
    RBCHK(rocblas_dgemm_batched(c->rb, rocblas_operation_none, rocblas_operation_none, k, nbe, nbe, &one,
                                (const double* const*)dRp, k, (const double* const*)dD, NB, &zero, dRb, k, nb));
    // row-major A -= Cb Rb  ==  column-major A^T -= Rb^T Cb^T
    RBCHK(rocblas_dgemm_batched(c->rb, rocblas_operation_none, rocblas_operation_none, k, k, nbe, &mone,
                                (const double* const*)dRb, k, (const double* const*)dCb, NB, &one, dA, k, nb));
    launch_gj_rows(st, nb, hmats, k, k0, nbe, c->gj_rb.p);
  }
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->flag.p + 2, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flag == 0;
}

// In-place inverses of nb dense k x k matrices (row major, device pointers in hp): the coarse matrices of a setup.
// Route 0: block Gauss-Jordan WITHOUT pivoting on batched GEMMs (gj_invert_batched) -- with the velocity
// aggregates ordered before the pressure aggregates that is block elimination of the coarse saddle matrix: the
// velocity block has a definite symmetric part for ADI shifts (and is s.p.d. for the projection), the Schur
// complement -B Av^-1 B^T inherits it.  A pivot that vanishes relative to its block's scale sends ALL matrices of
// the call through route 1: rocSOLVER's getrf / getri with partial pivoting (its unpivoted routines, the step in
// between until round 3, only notice an EXACTLY zero pivot -- a pivot of 1e-14 of the block's scale passed and left
// a garbage inverse).  `reassemble` restores the matrices the first route has overwritten.  info (nb entries):
// rocSOLVER's status.  Returns the route.
template <class F>
static int invert_dense_batch(ricadi_ctx* c, const std::vector<double*>& hp, int k, std::vector<int>& info,
                              F&& reassemble) {
  hipStream_t st = c->st;
  const int nb = (int)hp.size();
  bool done = true;
  for (int i0 = 0; i0 < nb && done; i0 += gj_max_batch())
    done = gj_invert_batched(c, hp.data() + i0, std::min(gj_max_batch(), nb - i0), k);
  if (done) {
    std::fill(info.begin(), info.end(), 0);
    return 0;
  }
  reassemble();
  // row-major E == column-major E^T; inv(E^T) column-major == inv(E) row-major
  c->ipiv.ensure((size_t)k * nb);
  c->info.ensure(nb);
  c->eptrs.ensure(nb);
  HIPCHK(hipMemcpyAsync(c->eptrs.p, hp.data(), sizeof(double*) * nb, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));   // hp may be a stack object of the caller
  RBCHK(rocsolver_dgetrf_batched(c->rb, k, k, c->eptrs.p, k, c->ipiv.p, k, c->info.p, nb));
  RBCHK(rocsolver_dgetri_batched(c->rb, k, c->eptrs.p, k, c->ipiv.p, k, c->info.p, nb));
  HIPCHK(hipMemcpyAsync(info.data(), c->info.p, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 1;
}

// Per-shift data for the given (alpha, beta) pairs; whatever is missing is built for
// all of them together: the element-wise / block kernels per shift, the dense coarse
// inverses in ONE batched rocSOLVER factorisation + inversion (its many small
// panel kernels then serve all shifts of a sweep per launch instead of one).
static void get_shifts(ricadi_ctx* c, const double* alphas, const double* betas, int ng,
                       ShiftData** out) {
  hipStream_t st = c->st;
  std::vector<ShiftData*> todo;
  for (int g = 0; g < ng; ++g) {
    auto key = std::make_pair(alphas[g], betas[g]);
    auto it = c->cache.find(key);
    if (it == c->cache.end()) it = c->cache.emplace(key, std::unique_ptr<ShiftData>(new ShiftData)).first;
    ShiftData* sd = it->second.get();
    out[g] = sd;
    if (sd->valid || std::find(todo.begin(), todo.end(), sd) != todo.end()) continue;
    sd->alpha = alphas[g];
    sd->beta = betas[g];
    sd->smw_epoch = -1;
    for (auto& r : sd->rec) r->serial = -1;     // stale, but the buffers stay (a hipFree / hipMalloc pair per panel
                                                // cost 13 ms per setup of 17 shifts)
    todo.push_back(sd);
  }
  if (todo.empty()) return;
  Tick tks;
  double tph[6] = {0, 0, 0, 0, 0, 0};
  auto lapS = [&](int i) {
    if (c->timing) {
      (void)hipStreamSynchronize(st);
      tph[i] += tks.lap();
    }
  };
  HIPCHK(hipMemsetAsync(c->flag.p, 0, sizeof(int), st));
  const size_t bsz = (size_t)c->bs * c->bs;
  const int k = c->kc;
  const int kd = c->child ? 0 : c->kc;   // size of the dense coarse inverse (none with a child level)
  if (c->child) {
    std::vector<double> al(todo.size()), be(todo.size());
    std::vector<ShiftData*> subs(todo.size(), nullptr);
    for (size_t i = 0; i < todo.size(); ++i) {
      al[i] = todo[i]->alpha;
      be[i] = todo[i]->beta;
    }
    get_shifts(c->child.get(), al.data(), be.data(), (int)todo.size(), subs.data());
    for (size_t i = 0; i < todo.size(); ++i) todo[i]->sub = subs[i];
  }
  lapS(0);
  for (ShiftData* sd : todo) {
    const double alpha = sd->alpha, beta = sd->beta;
    stable_alloc(sd->sval, c->snnz);
    launch_assemble_shift(st, (int)c->snnz, c->srcA.p, c->srcE.p, c->srcJ.p, alpha, beta,
                          sd->sval.p);
    if (c->sb_ok) {
      stable_alloc(sd->svalb, c->snnz);
      launch_gather_vals(st, (int)c->snnz, c->sb_perm.p, sd->sval.p, sd->svalb.p);
    }
    stable_alloc(sd->bvinv, (size_t)c->nbv * bsz);
    launch_block_combine(st, (size_t)c->nbv * bsz, c->bvA.p, c->bvE.p, alpha, beta, sd->bvinv.p);
    if (c->nbp > 0) stable_alloc(sd->bpinv, (size_t)c->nbp * bsz);
    if (c->gt_ok) stable_alloc(sd->gtm, (size_t)c->nbv * c->bs * c->gt_ks);
    if (c->ady_ok && k > 0) stable_alloc(sd->adym, (size_t)c->nbv * c->bs * c->ady_ks);
    if (kd > 0) {
      stable_alloc(sd->einv, (size_t)k * k);
      launch_combine3(st, (size_t)k * k, c->E0.p, c->EM.p, c->EJ.p, alpha, beta, sd->einv.p);
    }
    if (k > 0) {
      stable_alloc(sd->syval, c->synnz);
      launch_assemble_shift(st, (int)c->synnz, c->sy_A.p, c->sy_E.p, c->sy_J.p, alpha, beta,
                            sd->syval.p);
      if (c->syb_ok) {
        stable_alloc(sd->syvalb, c->synnz);
        launch_gather_vals(st, (int)c->synnz, c->syb_perm.p, sd->syval.p, sd->syvalb.p);
      }
    }
  }
  lapS(1);
  // block inversions and Schur blocks: one launch each for all shifts (<= 16 per call)
  for (size_t t0 = 0; t0 < todo.size(); t0 += RICADI_MAX_GROUPS) {
    const int cnt = (int)std::min<size_t>(RICADI_MAX_GROUPS, todo.size() - t0);
    GroupPtrs pv = same_ptr((const double*)nullptr), pp = pv;
    for (int i = 0; i < cnt; ++i) {
      pv.p[i] = todo[t0 + i]->bvinv.p;
      pp.p[i] = todo[t0 + i]->bpinv.p;
    }
    launch_block_invert(st, cnt, c->nbv, c->bs, c->bv_ptr.p, pv, c->flag.p);
    if (c->gt_ok) {
      GroupPtrs pg = same_ptr((const double*)nullptr);
      for (int i = 0; i < cnt; ++i) pg.p[i] = todo[t0 + i]->gtm.p;
      launch_gt_blocks(st, cnt, c->nbv, c->bs, c->gt_ks, c->gt_jtd.p, pv, pg);
    }
    if (c->ady_ok && k > 0) {
      GroupPtrs pa_ = same_ptr((const double*)nullptr);
      double al[RICADI_MAX_GROUPS], be[RICADI_MAX_GROUPS];
      for (int i = 0; i < cnt; ++i) {
        pa_.p[i] = todo[t0 + i]->adym.p;
        al[i] = todo[t0 + i]->alpha;
        be[i] = todo[t0 + i]->beta;
      }
      launch_ady_blocks(st, cnt, c->nbv, c->bs, c->ady_ks, c->cy_dA.p, c->cy_dE.p, c->cy_dJ.p,
                        c->sa ? c->cy_dT.p : nullptr, al, be, pv, pa_);
    }
    if (c->nbp > 0) {
      launch_schur_blocks_bj(st, cnt, c->nbp, c->bs, c->bp_ptr.p, c->jd_ptr.p, c->jd_vblk.p,
                             c->jd_val.p, pv, pp);
      launch_block_invert(st, cnt, c->nbp, c->bs, c->bp_ptr.p, pp, c->flag.p);
    }
  }
  lapS(2);
  const int nb = (int)todo.size();
  std::vector<int> info(nb, 0);
  if (kd > 0) {
    std::vector<double*> hp(nb);
    for (int i = 0; i < nb; ++i) hp[i] = todo[i]->einv.p;
    c->coarse_route = invert_dense_batch(c, hp, k, info, [&] {
      for (ShiftData* sd : todo)
        launch_combine3(st, (size_t)k * k, c->E0.p, c->EM.p, c->EJ.p, sd->alpha, sd->beta, sd->einv.p);
    });
  }
  lapS(3);
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int i = 0; i < nb; ++i)
    if (info[i] != 0)
      throw HipError{"coarse matrix singular (getrf/getri info " + std::to_string(info[i]) + ")"};
  if (flag) throw HipError{"singular block-Jacobi block"};
  if (c->precond32) {
    const int bs2 = c->bs * c->bs;
    for (ShiftData* sd : todo) {
      if (sd->bvinvf.n != sd->bvinv.n) sd->bvinvf.alloc(sd->bvinv.n);
      launch_to_f32(st, c->nbv, bs2, sd->bvinv.p, bs2, sd->bvinvf.p, bs2);
      if (c->nbp > 0) {
        if (sd->bpinvf.n != sd->bpinv.n) sd->bpinvf.alloc(sd->bpinv.n);
        launch_to_f32(st, c->nbp, bs2, sd->bpinv.p, bs2, sd->bpinvf.p, bs2);
      }
      if (c->gt_ok) {
        const int gsz = c->bs * c->gt_ks;
        if (sd->gtmf.n != sd->gtm.n) sd->gtmf.alloc(sd->gtm.n);
        launch_to_f32(st, c->nbv, gsz, sd->gtm.p, gsz, sd->gtmf.p, gsz);
      }
      if (c->ady_ok && k > 0) {
        const int gsz = c->bs * c->ady_ks;
        if (sd->adymf.n != sd->adym.n) sd->adymf.alloc(sd->adym.n);
        launch_to_f32(st, c->nbv, gsz, sd->adym.p, gsz, sd->adymf.p, gsz);
      }
      if (kd > 0) {
        const size_t kp = (size_t)(k + 15) / 16;
        if (sd->einvf.n != kp * kp * 256) sd->einvf.alloc(kp * kp * 256);
        launch_to_f32_tiled(st, k, sd->einv.p, sd->einvf.p);
      }
    }
    HIPCHK(hipStreamSynchronize(st));
  }
  lapS(4);
  if (c->timing && !c->borrowed)
    fprintf(stderr, "[ricadi timing] setup of %d shifts: child %.1f ms, per-shift assembly %.1f, block inverses + Schur blocks %.1f, coarse inverses %.1f, FP32 copies %.1f\n",
            (int)todo.size(), 1e3 * tph[0], 1e3 * tph[1], 1e3 * tph[2], 1e3 * tph[3], 1e3 * tph[4]);
  for (ShiftData* sd : todo) sd->valid = true;
}

static ShiftData* get_shift(ricadi_ctx* c, double alpha, double beta) {
  ShiftData* sd = nullptr;
  get_shifts(c, &alpha, &beta, 1, &sd);
  return sd;
}

// ---- batches -----------------------------------------------------------------------
// The shifts of one batched solve: per group id the shift-dependent operands, and
// the table of groups a launch works on (ricadi_internal.h).  All workspace
// buffers are group-major with the strides below.
struct Batch {
  int G = 0;                 // groups in the solve (ids 0 .. G-1)
  int m = 0;                 // panel width of every group
  GroupTab tab;              // groups the next launches act on
  double alpha[RICADI_MAX_GROUPS], beta[RICADI_MAX_GROUPS];   // shift of every group id
  GroupPtrs sval, svalb, syval, syvalb, bvinv, bpinv, einv;
  GroupPtrsF bvinvf, bpinvf, einvf;
  GroupPtrs gtm, adym;
  GroupPtrsF gtmf, adymf;
  size_t gs = 0, gsp = 0, gsc = 0, gsq = 0;   // strides: n*m, np*m, kc*m, q*m
  std::shared_ptr<Batch> sub;                 // the same groups on the child level

  void all() {
    tab.ng = G;
    for (int g = 0; g < G; ++g) tab.gid[g] = g;
  }
  void only(int g) {
    tab.ng = 1;
    tab.gid[0] = g;
  }
  void set(const std::vector<int>& ids) {
    tab.ng = (int)ids.size();
    for (int i = 0; i < tab.ng; ++i) tab.gid[i] = ids[i];
  }
};

static Batch make_batch(ricadi_ctx* c, ShiftData* const* sds, int G, int m) {
  Batch bt;
  bt.G = G;
  bt.m = m;
  bt.tab = GroupTab{};
  bt.sval = bt.svalb = bt.syval = bt.syvalb = bt.bvinv = bt.bpinv = bt.einv = same_ptr((const double*)nullptr);
  bt.bvinvf = bt.bpinvf = bt.einvf = same_ptr((const float*)nullptr);
  bt.gtm = bt.adym = same_ptr((const double*)nullptr);
  bt.gtmf = bt.adymf = same_ptr((const float*)nullptr);
  for (int g = 0; g < RICADI_MAX_GROUPS; ++g) bt.alpha[g] = bt.beta[g] = 0.0;
  for (int g = 0; g < G; ++g) {
    bt.alpha[g] = sds[g]->alpha;
    bt.beta[g] = sds[g]->beta;
    bt.bvinvf.p[g] = sds[g]->bvinvf.p;
    bt.bpinvf.p[g] = sds[g]->bpinvf.p;
    bt.einvf.p[g] = sds[g]->einvf.p;
    bt.gtm.p[g] = sds[g]->gtm.p;
    bt.gtmf.p[g] = sds[g]->gtmf.p;
    bt.adym.p[g] = sds[g]->adym.p;
    bt.adymf.p[g] = sds[g]->adymf.p;
    bt.sval.p[g] = sds[g]->sval.p;
    bt.syval.p[g] = sds[g]->syval.p;
    bt.syvalb.p[g] = sds[g]->syvalb.p;
    bt.svalb.p[g] = sds[g]->svalb.p;
    bt.bvinv.p[g] = sds[g]->bvinv.p;
    bt.bpinv.p[g] = sds[g]->bpinv.p;
    bt.einv.p[g] = sds[g]->einv.p;
  }
  bt.gs = (size_t)c->n * m;
  bt.gsp = (size_t)c->np * m;
  bt.gsc = (size_t)c->kc * m;
  bt.gsq = (size_t)std::max(c->q, 1) * m;
  bt.all();
  if (c->child) {
    ShiftData* subs[RICADI_MAX_GROUPS];
    for (int g = 0; g < G; ++g) subs[g] = sds[g]->sub;
    bt.sub = std::make_shared<Batch>(make_batch(c->child.get(), subs, G, m));
  }
  return bt;
}
static Batch make_batch(ricadi_ctx* c, ShiftData* sd, int m) { return make_batch(c, &sd, 1, m); }

// Multi-shift tile kernel or one workgroup per (row block, group)?  The multi-shift kernel
// reads the matrix once for all groups (26 -> 18 B per non-zero in total instead of 10 B per
// group) but walks the groups of a row block one after the other at 4 waves per SIMD; it
// pays where the per-shift value arrays of the active groups no longer fit the caches
// (measured: n = 5e5, 16 groups: 1.53 -> 1.25 ms per launch; n = 3e4: 83 -> 87 us).
static bool ms_pays(const ricadi_ctx* c, int ng, size_t nnz) {
  if (!c->ms_spmm) return false;
  if (c->ms_force) return true;
  // per-shift value arrays of the active groups near or beyond the 256 MB infinity cache (measured with the FP32
  // operator input that follows this switch: cfg3, 227 MB: 197 -> 205 shift-solves/s; cfg2, 136 MB: 1.4 % slower)
  return ng >= 4 && (double)nnz * 10.0 * ng > 200e6;
}

// ---- operator and preconditioner on device panels ---------------------------------
// y = beta_r * r + alpha * S x on the saddle operator (optionally through the
// prolongation map): LDS-tiled kernel when the block tiles fit, else the CSR one.
// gsx / gsy / gsr: group strides of x, y, r.
// The LDS-tiled kernels serve panels of width m (else the CSR kernel runs)
static bool saddle_tiled(const ricadi_ctx* c, int m) {
  return c->sb_ok &&
         spmm_blocked_lds_bytes(m, c->sb_max_cols, c->sb_max_nnz) <= (size_t)40 * 1024;
}
// x32 (optional): FP32 copy of x with the same leading dimension and group stride; the tiled kernels read it
// instead of x (plain products only: no residual term, no low-rank epilogue, no prolongation map)
static void saddle_spmm(ricadi_ctx* c, const Batch& bt, const double* x, size_t gsx,
                        const int* xmap, double* y, size_t gsy, const double* r, size_t gsr,
                        double alpha, double beta_r, const LowRankArgs& lr = LowRankArgs(),
                        const float* x32 = nullptr) {
  const int m = bt.m;
  const bool fits = saddle_tiled(c, m);
  const bool has_lr = lr.q > 0 && lr.nrows > 0;
  if (x32 && fits && !r && !xmap && !has_lr) {
    const bool ms = ms_pays(c, bt.tab.ng, c->snnz) && spmm_blocked_ms_ok(m, c->sb_max_cols, (size_t)c->n);
    c->k1_variant = (ms ? 2 : 1) + 4;
    if (ms)
      launch_spmm_blocked_ms_x32(c->st, bt.tab, bt.alpha, bt.beta, c->sb_nblk, c->sb_rows2.p, c->sb_rp2.p,
                                 c->sb_cols2.p, c->sb_lidx_ms.p, c->sbAJ.p, c->sbE.p, x32, m, gsx, y, m, gsy, alpha,
                                 m, c->sb_max_cols);
    else
      launch_spmm_blocked_x32(c->st, bt.tab, c->sb_nblk, c->sb_rows2.p, c->sb_rp2.p, c->sb_cols2.p, c->sb_lidx.p,
                              bt.svalb, x32, m, gsx, y, m, gsy, alpha, m, c->sb_max_cols);
    return;
  }
  const bool ms = fits && ms_pays(c, bt.tab.ng, c->snnz) && !xmap && !has_lr &&
                  spmm_blocked_ms_ok(m, c->sb_max_cols, (size_t)c->n);
  if (!xmap) c->k1_variant = ms ? 2 : fits ? 1 : 0;
  if (ms)
    launch_spmm_blocked_ms(c->st, bt.tab, bt.alpha, bt.beta, c->sb_nblk, c->sb_rows2.p, c->sb_rp2.p,
                           c->sb_cols2.p, c->sb_lidx_ms.p, c->sbAJ.p, c->sbE.p, x, m, gsx, y, m, gsy,
                           r, m, gsr, alpha, beta_r, m, c->sb_max_cols);
  else if (fits)
    launch_spmm_blocked_b(c->st, bt.tab, c->sb_nblk, c->sb_rows2.p, c->sb_rp2.p,
                          xmap ? c->sb_colsm2.p : c->sb_cols2.p, c->sb_lidx.p, bt.svalb, x, m, gsx, y,
                          m, gsy, r, m, gsr, alpha, beta_r, m, c->sb_max_cols, lr);
  else
    launch_spmm_b(c->st, bt.tab, c->n, c->s_rp.p, c->s_ci.p, bt.sval, x, m, gsx, xmap, y, m, gsy, r,
                  m, gsr, alpha, beta_r, m, lr);
}

// Does the GMRES iteration apply the operator to the FP32-stored Z_j (RICADI_X32=0: to the FP64 z)?
static bool operator_reads_x32(const ricadi_ctx* c, int m) {
  return c->flex && saddle_tiled(c, m);
}

// y = S(alpha,beta) x for every active group (n x m panels, ld = m, group stride gsx /
// bt.gs); optional low-rank  - U V^T x_v  (U, V shared by the groups)
static void op_apply(ricadi_ctx* c, const Batch& bt, const double* x, size_t gsx, double* y,
                     bool lowrank, const float* x32 = nullptr) {
  hipStream_t st = c->st;
  const int m = bt.m;
  LowRankArgs lr;
  if (lowrank && c->q > 0) {
    // coefficients V^T x first; the product with U rides in the SpMM's epilogue
    HIPCHK(hipMemsetAsync(c->lrc.p, 0, sizeof(double) * bt.gsq * bt.G, st));
    launch_gemm_tn_b(st, bt.tab, c->nv, c->q, m, c->V.p, c->q, x, m, gsx, c->lrc.p, m, bt.gsq);
    lr.U = c->U.p;
    lr.c = c->lrc.p;
    lr.gsc = bt.gsq;
    lr.q = c->q;
    lr.nrows = c->nv;
  }
  saddle_spmm(c, bt, x, gsx, nullptr, y, bt.gs, nullptr, 0, 1.0, 0.0, lr, x32);
}

// z = P^-1 r for every active group: multiplicative two-level, coarse correction
// first, then one consistent SIMPLE block-Jacobi sweep on the updated residual.
// r has group stride gsr; z lives in a workspace buffer (stride bt.gs).
// z32 (optional, group stride gs32): FP32 copy of z, written by the sweeps that write z last.
// only32: z itself need not be stored where the sweeps write the copy (the operator will read z32).
// r16: the same residual panel as stored in FP16 (the current Krylov vector; group stride gsr); where the folded
// path runs, its three readers of r take the 2-byte copy (exactly the same values) and r itself is not touched.
static bool precond_folds(const ricadi_ctx* c) {
  return c->kc > 0 && c->ady_ok && c->np > 0;
}
static bool precond_reads_h16_static(const ricadi_ctx* c) {
  return precond_folds(c);
}
// Does the GMRES iteration hand the preconditioner the FP16-stored vector (RICADI_H16=0: the FP64 copy)?
static bool precond_reads_h16(const ricadi_ctx* c, int m) {
  return c->basis16 && m <= 16 && precond_folds(c);
}
static void precond_apply(ricadi_ctx* c, const Batch& bt, const double* r, size_t gsr, double* z,
                          float* z32 = nullptr, size_t gs32 = 0, bool only32 = false,
                          const _Float16* r16 = nullptr) {
  hipStream_t st = c->st;
  bool mirrored = false;
  const int nv = c->nv, np = c->np, m = bt.m;
  const GroupTab& gt = bt.tab;
  const GroupPtrs ones = same_ptr(c->ones.p), jv = same_ptr(c->J.v.p), jtv = same_ptr(c->JT.v.p);
  const double* rr = r;
  size_t gsrr = gsr;
  bool folded = false;
  // ricadi_time_kernel_dev times one stage at a time through exactly these launchers (c->pc_stage >= 0)
  auto on = [&](int stage) { return c->pc_stage < 0 || c->pc_stage == stage; };
  // the pressure step -- pressure rows of r - (S Y) e, J product, Schur sweep -- as ONE launch (K2p) for
  // 16-column panels (RICADI_PFUSE=0: the three launches of round 2)
  const bool fusedp = np > 0 && m == 16 && c->bs == 32;
  if (c->kc > 0) {
    // restriction Y^T r = CSR product with unit values (aggregate lists as rows)
    folded = precond_folds(c);
    if (!folded || m > 16) r16 = nullptr;
    // (smoothed aggregation: P^T r with the rows of P^T)
    const int* rrp = c->sa ? c->pt_rp.p : c->agg_ptr.p;
    const int* rci = c->sa ? c->pt_ci.p : c->agg_rows.p;
    const GroupPtrs rvals = c->sa ? same_ptr((const double*)c->pt_v.p) : ones;
    if (c->sa && !folded) throw HipError{"smoothed aggregation needs the folded preconditioner cycle"};
    if (!on(0)) {
    } else if (r16)
      launch_spmm_h(st, gt, c->kc, rrp, rci, rvals, nullptr, r16, m, gsr, c->rc.p, m, bt.gsc,
                    nullptr, 0, 0, 1.0, 0.0, m, 16);
    else
      launch_spmm_b(st, gt, c->kc, rrp, rci, rvals, r, m, gsr, nullptr, c->rc.p, m,
                    bt.gsc, nullptr, 0, 0, 1.0, 0.0, m);
    if (!on(1)) {
    } else if (c->child) {
      // coarse problem by one cycle of the child level's preconditioner (a fixed linear operator)
      Batch cb = *bt.sub;
      cb.tab = gt;
      precond_apply(c->child.get(), cb, c->rc.p, bt.gsc, c->ec.p);
    } else if (c->precond32)
      launch_dense_apply_b(st, gt, c->kc, m, bt.einvf, (c->kc + 3) & ~3, c->rc.p, c->ec.p);
    else
      launch_dense_apply_b(st, gt, c->kc, m, bt.einv, c->rc.p, c->ec.p);
    if (!on(2) || (fusedp && folded)) {
    } else if (folded) {
      // only the PRESSURE rows of r - (S Y) ec are formed (short CSR product over np rows); the
      // velocity rows ride inside the first velocity sweep (block_apply2_kernel, below)
      if (r16)
        launch_spmm_h(st, gt, np, c->sy_rp.p + nv, c->sy_ci.p, bt.syval, c->ec.p, nullptr, m, bt.gsc,
                      c->r2.p + (size_t)nv * m, m, bt.gs, r16 + (size_t)nv * m, m, gsr, -1.0, 1.0, m, c->sy_chunk);
      else
        launch_spmm_b(st, gt, np, c->sy_rp.p + nv, c->sy_ci.p, bt.syval, c->ec.p, m, bt.gsc, nullptr,
                      c->r2.p + (size_t)nv * m, m, bt.gs, r + (size_t)nv * m, m, gsr, -1.0, 1.0, m, LowRankArgs(),
                      c->sy_chunk);
    } else {
      // Residual after the coarse correction, r2 = r - (S Y) ec, with the prolongated
      // operator (short rows over the L2-resident coarse vector) -- not a full saddle SpMM
      // through the prolongation map.  (Forming the velocity rows of r2 inside the first
      // velocity sweep instead, like the J^T product below, was measured slower: 249 vs
      // 257 shift-solves/s -- 8 rows x 7.6 dependent gathers per lane.)
      // Tile form: the aggregates a row block touches (a few dozen coarse rows) go to LDS once.
      const bool sy_csr = false;
      if (c->syb_ok && !sy_csr && ms_pays(c, gt.ng, c->snnz) &&
          spmm_blocked_ms_ok(m, c->syb_max_cols, (size_t)c->kc))
        launch_spmm_blocked_ms(st, gt, bt.alpha, bt.beta, c->sb_nblk, c->sb_rows2.p, c->syb_rp2.p,
                               c->syb_cols2.p, c->syb_lidx_ms.p, c->sybAJ.p, c->sybE.p, c->ec.p, m,
                               bt.gsc, c->r2.p, m, bt.gs, r, m, gsr, -1.0, 1.0, m, c->syb_max_cols);
      else if (c->syb_ok && !sy_csr &&
          spmm_blocked_lds_bytes(m, c->syb_max_cols, 0) <= (size_t)40 * 1024)
        launch_spmm_blocked_b(st, gt, c->sb_nblk, c->sb_rows2.p, c->syb_rp2.p, c->syb_cols2.p,
                              c->syb_lidx.p, bt.syvalb, c->ec.p, m, bt.gsc, c->r2.p, m, bt.gs, r, m, gsr,
                              -1.0, 1.0, m, c->syb_max_cols);
      else
        launch_spmm_b(st, gt, c->n, c->sy_rp.p, c->sy_ci.p, bt.syval, c->ec.p, m, bt.gsc, nullptr,
                      c->r2.p, m, bt.gs, r, m, gsr, -1.0, 1.0, m, LowRankArgs(), c->sy_chunk);
    }
    rr = c->r2.p;
    gsrr = bt.gs;
  }
  // the LAST velocity sweep also adds the coarse correction Y ec to all of z
  // (its surplus waves take the pressure rows)
  ProlongArgs pro;
  if (c->kc > 0) {
    pro.aggof = c->aggof.p;
    pro.ec = c->ec.p;
    pro.gse = bt.gsc;
    pro.row0 = nv;
    pro.nextra = np;
  }
  auto vel_apply = [&](const double* in, size_t gsi, int subtract, bool last,
                       const CsrInArgs& cin = CsrInArgs()) {
    const ProlongArgs pa = last ? pro : ProlongArgs();
    if (c->precond32)
      launch_block_apply_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinvf, in, m, gsi, z,
                           m, bt.gs, m, subtract, pa, cin);
    else
      launch_block_apply_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinv, in, m, gsi, z,
                           m, bt.gs, m, subtract, pa, cin);
  };
  if (!on(3)) {
  } else if (folded) {
    // z_v = Ahat^-1 r_v - (Ahat^-1 D) ec : first velocity sweep on the corrected residual without
    // ever writing it
    Seg2 s1, s2;
    s1.kstride = c->bs;
    s1.in = r16 ? nullptr : r;
    s1.in16 = r16;
    s1.gs = gsr;
    s2.iptr = c->cy_ptr.p;
    s2.irows = c->cy_cols.p;
    s2.kstride = c->ady_ks;
    s2.in = c->ec.p;
    s2.gs = bt.gsc;
    if (c->precond32)
      launch_block_apply2_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinvf, s1, bt.adymf, s2, z, m,
                            bt.gs, m, ProlongArgs());
    else
      launch_block_apply2_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinv, s1, bt.adym, s2, z, m,
                            bt.gs, m, ProlongArgs());
  } else {
    vel_apply(rr, gsrr, 0, np == 0);
  }
  if (np > 0) {
    // t = J z_v - r_p
    if (on(4) && !fusedp)
      launch_spmm_b(st, gt, np, c->J.rp.p, c->J.ci.p, jv, z, m, bt.gs, nullptr, c->tp.p, m, bt.gsp,
                    rr + (size_t)nv * m, m, gsrr, 1.0, -1.0, m);
    double* zp = z + (size_t)nv * m;
    const bool fuse_jt = true, rect = true;
    // Fused variant: the pressure sweep writes z_p already WITH its coarse part and keeps
    // the plain z_p (the operand of the J^T product below) in tp -- in place: a wave
    // reads its block's rows of tp before it writes them, blocks are disjoint.
    ProlongArgs ppro;
    if (fuse_jt) {
      ppro.out2 = c->tp.p;
      ppro.gs2 = bt.gsp;
      if (z32) {
        ppro.out32 = z32 + (size_t)nv * m;
        ppro.gs32 = gs32;
        ppro.only32 = only32 && rect && c->gt_ok;   // the rectangle sweep below completes the FP32 copy
      }
      if (c->kc > 0) {
        ppro.aggof = c->aggof.p + nv;
        ppro.ec = c->ec.p;
        ppro.gse = bt.gsc;
      }
    }
    if (!on(5)) {
    } else if (fusedp) {
      // r_p: of the folded cycle the input vector itself (FP64 or FP16-stored) with the coarse term formed in
      // the kernel; else the pressure rows of the corrected residual r2
      const bool sy = folded;
      const double* rp64 = sy ? (r16 ? nullptr : r + (size_t)nv * m) : rr + (size_t)nv * m;
      const _Float16* rp16 = sy && r16 ? r16 + (size_t)nv * m : nullptr;
      const size_t gsrp = sy ? gsr : gsrr;
      if (c->precond32)
        launch_pressure_step_b(st, gt, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinvf, c->J.rp.p, c->J.ci.p, c->J.v.p, z,
                               bt.gs, sy ? c->sy_rp.p + nv : nullptr, c->sy_ci.p, bt.syval, c->ec.p, bt.gsc, rp64, rp16,
                               gsrp, zp, bt.gs, ppro);
      else
        launch_pressure_step_b(st, gt, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinv, c->J.rp.p, c->J.ci.p, c->J.v.p, z,
                               bt.gs, sy ? c->sy_rp.p + nv : nullptr, c->sy_ci.p, bt.syval, c->ec.p, bt.gsc, rp64, rp16,
                               gsrp, zp, bt.gs, ppro);
    } else if (c->precond32)
      launch_block_apply_b(st, gt, c->bs, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinvf, c->tp.p, m,
                           bt.gsp, zp, m, bt.gs, m, 0, ppro);
    else
      launch_block_apply_b(st, gt, c->bs, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinv, c->tp.p, m,
                           bt.gsp, zp, m, bt.gs, m, 0, ppro);
    // z_v -= Ahat^-1 (J^T z_p): the same block-Jacobi inverse as in the Schur blocks; the
    // J^T product is formed inside the sweep, row by row as the blocks gather them
    // (z_p is small and L2 resident), instead of through an intermediate panel
    if (!on(6)) {
    } else if (fuse_jt && rect && c->gt_ok) {
      // z_v -= G z_p with the per-shift blocks G_b = Ahat_b^-1 J^T[rows_b, pcols_b] formed at setup
      pro.nextra = 0;            // the pressure rows already carry their coarse part
      pro.out32 = z32;
      pro.gs32 = gs32;
      pro.only32 = only32;
      mirrored = true;
      if (c->precond32)
        launch_block_apply_rect_b(st, gt, c->bs, c->gt_ks, c->nbv, c->bv_ptr.p, c->bv_rows.p, c->gt_ptr.p,
                                  c->gt_cols.p, bt.gtmf, c->tp.p, m, bt.gsp, z, m, bt.gs, m, 1, pro);
      else
        launch_block_apply_rect_b(st, gt, c->bs, c->gt_ks, c->nbv, c->bv_ptr.p, c->bv_rows.p, c->gt_ptr.p,
                                  c->gt_cols.p, bt.gtm, c->tp.p, m, bt.gsp, z, m, bt.gs, m, 1, pro);
    } else if (fuse_jt) {
      CsrInArgs cin;
      cin.rp = c->JT.rp.p;
      cin.ci = c->JT.ci.p;
      cin.v = jtv;
      cin.src = c->tp.p;
      cin.gss = bt.gsp;
      pro.nextra = 0;          // the pressure rows already carry their coarse part
      vel_apply(nullptr, 0, 1, true, cin);
    } else {
      double* tmp = c->r2.p;   // the corrected residual is no longer needed at this point
      launch_spmm_b(st, gt, nv, c->JT.rp.p, c->JT.ci.p, jtv, zp, m, bt.gs, nullptr, tmp, m, bt.gs,
                    nullptr, 0, 0, 1.0, 0.0, m, LowRankArgs(), 8);    // J^T has ~5 entries per row
      vel_apply(tmp, bt.gs, 1, true);
    }
  }
  if (z32 && !mirrored && c->pc_stage < 0)
    for (int i = 0; i < gt.ng; ++i)
      launch_to_f32(st, c->n, m, z + (size_t)gt.gid[i] * bt.gs, m, z32 + (size_t)gt.gid[i] * gs32, m);
}

static void op_apply(ricadi_ctx* c, ShiftData* sd, const double* x, double* y, int m, bool lowrank) {
  const Batch bt = make_batch(c, sd, m);
  op_apply(c, bt, x, bt.gs, y, lowrank);
}
static void precond_apply(ricadi_ctx* c, ShiftData* sd, const double* r, double* z, int m) {
  const Batch bt = make_batch(c, sd, m);
  precond_apply(c, bt, r, bt.gs, z);
}

static void col_norms2(ricadi_ctx* c, const double* w, int nrows, int m, double* out) {
  launch_cols_dots(c->st, nrows, m, 0, nullptr, 0, w, 1, c->partial.p, out);
}

// ---- batched panel GMRES --------------------------------------------------------------
// Solves S(shift_g) x_g = b_g for the m columns of every group's n x m panel:
// one Arnoldi process per column, all groups in lockstep inside ONE sequence of
// launches (grid.z = active groups).  At n ~ 3e4 a single panel leaves most of
// the chip idle and the launch path dominates; batching the shifts of a sweep
// fills it.  Right preconditioning, CGS2, per-column Givens QR.  A group whose
// columns have all converged leaves the active table; its correction is formed
// at the end of the restart cycle from the basis vectors it had by then.
//   b: group stride gsb (0 = one right-hand side shared by all groups);
//   x: group stride n*m, overwritten.
struct GmresResult {
  int iters = 0;
  bool converged = false;
  bool stalled = false;       // gave up before gmres_maxit: three full-length cycles in a row gained < 30 %
  double max_relres = 0.0;
};

// have_x0: x holds an initial guess (else it is zeroed);  only: the groups to iterate on (NULL = all; the
// panels of the other groups are not touched);  allow_stall: a group whose full-length restart cycles no
// longer gain is given up early (the caller repeats it with wider storage).
static void gmres_core(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, size_t gsb, double* x,
                       int m, bool lowrank, GmresResult* res, bool have_x0, const std::vector<int>* only,
                       bool allow_stall) {
  ensure_work(c, m, G, 0);
  hipStream_t st = c->st;
  const int n = c->n, restart = c->opts.gmres_restart, maxit = c->opts.gmres_maxit;
  const double tol = c->opts.gmres_tol;
  Batch bt = make_batch(c, sds, G, m);
  const size_t nm = bt.gs;             // one panel
  const size_t vs = nm * G;            // one Krylov vector of all groups
  const size_t gsh = (size_t)(restart + 2) * m;
  const size_t gspart = (size_t)dots_num_blocks(n) * (restart + 2) * m;
  const int GM = G * m;
  double* V = c->basis.p;          // FP64 basis (RICADI_BASIS64) ...
  float* Vf = c->basisf.p;         // ... or the FP32-stored one
  const bool b16 = c->basis16;
  const bool b32 = c->basis32 && !b16;
  const bool flex = c->flex;
  // (only where the launches are bandwidth bound -- the multi-shift SpMM regime: cfg5 K1 1252 -> 1150 us per
  // launch, cycle +2 %; at cfg2 the FP32 gathers are no faster and the step was 1.4 % slower)
  // the preconditioner reads the current vector from the FP16 basis itself; its FP64 copy is then not written
  const bool h16 = precond_reads_h16(c, m);
  // dot passes with atomic accumulation (no partial rows, no reduce launches): FP16 basis, 16 columns
  // w is not rewritten between the two Gram-Schmidt passes: the final update subtracts V (h1 + h2) from the original w
  const bool keepw = update_dots_keeps_w(m, b16, restart);
  // last Arnoldi pass and Hessenberg update in ONE launch (K3h)
  const bool fuseh = update_hess_fused_ok(m, b16);
  const size_t resbuf = (size_t)c->wcols;                    // doubles between the two residual-estimate buffers
  struct NoStoreScope {
    explicit NoStoreScope(bool v) { set_update_dots_nostore(v); }
    ~NoStoreScope() { set_update_dots_nostore(false); }
  } nostore_scope(keepw);
  const size_t h2buf = (size_t)(restart + 2) * c->wcols;        // doubles between the two second-pass buffers
  const bool x32 = operator_reads_x32(c, m) && ms_pays(c, G, c->snnz) && !(lowrank && c->q > 0);
  _Float16* Vh = reinterpret_cast<_Float16*>(c->basisf.p);   // FP16 storage shares the FP32 buffer
  double* hb = c->h_resid;
  const size_t slot = (size_t)RICADI_MAX_M * RICADI_MAX_GROUPS;
  for (int g = 0; g < G; ++g) res[g] = GmresResult();

  auto norms2 = [&](const double* w, size_t gsw, double* out) {
    launch_cols_dots_b(st, bt.tab, n, m, 0, (const double*)nullptr, 0, 0, w, gsw, 1, c->partial.p, gspart, out,
                       (size_t)m);
  };
  bt.all();
  norms2(b, gsb, c->bnorm2.p);
  HIPCHK(hipMemcpyAsync(hb + slot, c->bnorm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::vector<double> bn(GM);
  for (int j = 0; j < GM; ++j) bn[j] = std::sqrt(std::max(hb[slot + j], 0.0));
  // device copy of the norms (not squared) for the hess kernel
  HIPCHK(hipMemcpyAsync(c->bnorm2.p, bn.data(), sizeof(double) * GM, hipMemcpyHostToDevice, st));
  if (!have_x0) HIPCHK(hipMemsetAsync(x, 0, sizeof(double) * vs, st));

  auto group_converged = [&](const double* r, int g) {
    double worst = 0.0;
    bool ok = true;
    for (int j = g * m; j < (g + 1) * m; ++j) {
      const double rel = bn[j] > 0.0 ? r[j] / bn[j] : 0.0;
      worst = std::max(worst, rel);
      if (!(r[j] <= tol * bn[j])) ok = false;
    }
    res[g].max_relres = worst;
    return ok;
  };

  std::vector<char> done(G, 0);
  std::vector<int> act, live, kk(G, 0), nstall(G, 0);
  if (only) act = *only;
  else
    for (int g = 0; g < G; ++g) act.push_back(g);
  bool first = !have_x0;
  // Cycle length: short cycles keep the Krylov basis (the dominant HBM traffic of an
  // iteration: three passes over it) small; a cycle that gains less than a factor 10
  // on some column lengthens the following ones, up to gmres_restart.
  const int cyc0 = 10;
  int cyc = std::min(restart, cyc0);
  std::vector<double> rstart(GM, 0.0);
  Tick tkc;
  auto lapc = [&](double& acc) {
    if (c->timing) {
      (void)hipStreamSynchronize(st);
      acc += tkc.lap();
    }
  };
  while (!act.empty()) {
    lapc(c->t_iter);
    bt.set(act);
    // residual of the current iterates
    if (first) {
      if (gsb == nm) {
        HIPCHK(hipMemcpyAsync(c->wv.p, b, sizeof(double) * vs, hipMemcpyDeviceToDevice, st));
      } else {
        for (int g = 0; g < G; ++g)
          HIPCHK(hipMemcpyAsync(c->wv.p + (size_t)g * nm, b + (size_t)g * gsb, sizeof(double) * nm,
                                hipMemcpyDeviceToDevice, st));
      }
    } else if (lowrank && c->q > 0) {
      op_apply(c, bt, x, nm, c->wv.p, lowrank);
      launch_axpby_b(st, bt.tab, nm, 1.0, b, gsb, -1.0, c->wv.p, nm);
    } else {
      // r = b - S x in one launch (the residual form of the SpMM)
      saddle_spmm(c, bt, x, nm, nullptr, c->wv.p, nm, b, gsb, -1.0, 1.0);
    }
    first = false;
    norms2(c->wv.p, nm, c->nrm2.p);
    launch_gmres_start_b(st, bt.tab, m, restart, c->nrm2.p, c->g.p, c->scale.p, c->resid.p);
    HIPCHK(hipMemcpyAsync(hb, c->resid.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int> next;
    bool slow = false;
    for (int g : act) {
      if (group_converged(hb, g)) {
        res[g].converged = true;
        done[g] = 1;
      } else if (res[g].iters >= maxit) {
        done[g] = 1;
      } else {
        bool flat = false;
        for (int j = g * m; j < (g + 1) * m; ++j) {
          if (rstart[j] > 0.0 && hb[j] > tol * bn[j] && hb[j] > 0.1 * rstart[j]) slow = true;
          if (rstart[j] > 0.0 && hb[j] > tol * bn[j] && hb[j] > 0.7 * rstart[j]) flat = true;
          rstart[j] = hb[j];
        }
        nstall[g] = (flat && cyc >= restart) ? nstall[g] + 1 : 0;
        if (allow_stall && nstall[g] >= 3) {
          res[g].stalled = true;
          done[g] = 1;
        } else {
          next.push_back(g);
        }
      }
    }
    if (slow) cyc = std::min(restart, cyc + (cyc + 1) / 2);
    // Few groups left (the stragglers of the sweep): the launches are latency bound then and
    // the traffic of a longer Krylov basis costs nothing -- let the cycles run to the full
    // restart length instead of throwing the subspace away every `cyc` vectors.
    act.swap(next);
    if (act.empty()) break;
    bt.set(act);
    if (b16)
      launch_colscale_b(st, bt.tab, n, m, c->scale.p, c->wv.p, nm, 0.0, c->vcur.p, nm, Vh, nm);
    else if (b32)
      launch_colscale_b(st, bt.tab, n, m, c->scale.p, c->wv.p, nm, 0.0, c->vcur.p, nm, Vf, nm);
    else
      launch_colscale_b(st, bt.tab, n, m, c->scale.p, c->wv.p, nm, 0.0, V, nm);
    live = act;
    for (int g : act) kk[g] = 0;
    lapc(c->t_cyc);
    for (int j = 0; j < cyc && !live.empty(); ++j) {
      bt.set(live);
      const double* vj = (b32 || b16) ? c->vcur.p : V + (size_t)j * vs;
      // flexible form: Z_j = P^-1 v_j is kept (FP32), the cycle's correction is x += Z y -- no
      // preconditioner application at the cycle end, and P may differ from step to step
      // ... and the operator reads that stored FP32 copy (half the bytes of the x gathers; S Z_j = V H then
      // holds for exactly the vectors the correction uses), so the sweeps need not store the FP64 z at all
      float* zj = flex ? c->zbasisf.p + (size_t)j * vs : nullptr;
      precond_apply(c, bt, vj, nm, c->zv.p, zj, nm, x32, h16 ? Vh + (size_t)j * vs : nullptr);
      op_apply(c, bt, c->zv.p, nm, c->wv.p, lowrank, x32 ? zj : nullptr);
      double* h2cur = c->h2.p;
      if (b16) {
        launch_cols_dots_b(st, bt.tab, n, m, j + 1, Vh, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart,
                           c->h1.p, gsh);
        launch_cols_update_dots_b(st, bt.tab, n, m, j + 1, Vh, vs, nm, c->h1.p, gsh, c->wv.p, nm,
                                  c->partial.p, gspart, c->h2.p, gsh);
      } else if (b32) {
        launch_cols_dots_b(st, bt.tab, n, m, j + 1, Vf, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart,
                           c->h1.p, gsh);
        launch_cols_update_dots_b(st, bt.tab, n, m, j + 1, Vf, vs, nm, c->h1.p, gsh, c->wv.p, nm,
                                  c->partial.p, gspart, c->h2.p, gsh);
      } else {
        launch_cols_dots_b(st, bt.tab, n, m, j + 1, V, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart,
                           c->h1.p, gsh);
        // first update fused with the dot products of the second pass
        launch_cols_update_dots_b(st, bt.tab, n, m, j + 1, V, vs, nm, c->h1.p, gsh, c->wv.p, nm,
                                  c->partial.p, gspart, c->h2.p, gsh);
      }
      // the residual estimates also go straight to a pinned host slot (read one
      // iteration later, behind the event below)
      double* cur = hb + 2 * slot + (size_t)(j & 1) * slot;
      if (fuseh)
        launch_cols_update16_hess_b(st, bt.tab, n, j + 1, Vh, vs, nm, c->h1.p, h2cur, gsh, keepw ? 1 : 0, c->wv.p, nm,
                                    h16 ? nullptr : c->vcur.p, nm, Vh + (size_t)(j + 1) * vs, nm, j, restart, c->H.p,
                                    c->cs.p, c->sn.p, c->g.p, c->resid.p + (size_t)(j & 1) * resbuf,
                                    c->resid.p + (size_t)((j + 1) & 1) * resbuf, c->bnorm2.p, tol, cur);
      else
        launch_gmres_hess_b(st, bt.tab, m, j, restart, c->h1.p, h2cur, c->H.p, c->cs.p, c->sn.p,
                            c->g.p, c->scale.p, c->resid.p, c->bnorm2.p, tol, cur, nullptr, nullptr,
                            keepw ? c->h2.p + h2buf : nullptr);
      if (fuseh) {
      } else if (b16)
        launch_cols_update_b(st, bt.tab, n, m, j + 1, Vh, vs, nm, keepw ? c->h2.p + h2buf : h2cur, gsh, -1.0, c->wv.p, nm,
                             c->scale.p, h16 ? nullptr : c->vcur.p, nm, Vh + (size_t)(j + 1) * vs, nm);
      else if (b32)
        launch_cols_update_b(st, bt.tab, n, m, j + 1, Vf, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm,
                             c->scale.p, c->vcur.p, nm, Vf + (size_t)(j + 1) * vs, nm);
      else
        launch_cols_update_b(st, bt.tab, n, m, j + 1, V, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm,
                             c->scale.p, V + (size_t)(j + 1) * vs, nm);
      // Residual estimates travel to a pinned slot behind an event; the host
      // looks at the PREVIOUS iteration's slot, so it never drains the stream
      // (one iteration of lag: at most one surplus Arnoldi step per group).
      HIPCHK(hipEventRecord(c->ev_res[j & 1], st));
      for (int g : live) {
        ++res[g].iters;
        kk[g] = j + 1;
      }
      std::vector<int> still;
      if (j >= 1) {
        HIPCHK(hipEventSynchronize(c->ev_res[(j - 1) & 1]));
        const double* prev = hb + 2 * slot + (size_t)((j - 1) & 1) * slot;
        for (int g : live)
          if (!group_converged(prev, g) && res[g].iters < maxit) still.push_back(g);
      } else {
        for (int g : live)
          if (res[g].iters < maxit) still.push_back(g);
      }
      live.swap(still);
    }
    lapc(c->t_iter);
    // corrections: x_g += P^-1 (V_g y_g) with the k_g basis vectors group g built
    // (one launch each for all groups of the cycle, k_g per group by value)
    bt.set(act);
    {
      GroupInts ks = same_int(0);
      for (int g : act) ks.v[g] = kk[g];
      launch_gmres_backsolve_b(st, bt.tab, m, ks, restart, c->H.p, c->g.p, c->yv.p);
      if (flex)      // x += Z y in one launch
        launch_cols_update_bk(st, bt.tab, n, m, ks, c->zbasisf.p, vs, nm, c->yv.p, (size_t)restart * m, x, nm, x, nm);
      else if (b16)
        launch_cols_update_bk(st, bt.tab, n, m, ks, Vh, vs, nm, c->yv.p, (size_t)restart * m, c->wv.p, nm);
      else if (b32)
        launch_cols_update_bk(st, bt.tab, n, m, ks, Vf, vs, nm, c->yv.p, (size_t)restart * m, c->wv.p, nm);
      else
        launch_cols_update_bk(st, bt.tab, n, m, ks, V, vs, nm, c->yv.p, (size_t)restart * m, c->wv.p, nm);
    }
    bt.set(act);
    if (flex) {
    } else {
      precond_apply(c, bt, c->wv.p, nm, c->zv.p);
      launch_axpby_b(st, bt.tab, nm, 1.0, c->zv.p, nm, 1.0, x, nm);
    }
    lapc(c->t_cyc);
  }
  lapc(c->t_cyc);
}

// ---- wide panels as sixteen-column groups -------------------------------------------------------
// The columns of a panel are independent Arnoldi processes (per-column Givens), so an n x m panel with
// m > 32 -- the time-varying Riccati loop's [M^T Z_c, sqrt(tau) C~^T, K_k] of up to comprz_maxc + NY' + NU
// columns, /root/reference/solve_dae_ric.py:149 -- is solved as groups of 16 columns of the SAME shift in
// the lockstep batch: every kernel tuned for the 16-column case (LDS-tiled SpMM, 16-byte Arnoldi kernels,
// fused pressure step, FP16 vector input) then carries the iteration instead of the generic-width ones.
// The shifts of the call are walked in chunks of floor(RICADI_MAX_GROUPS / groups per shift); the column
// groups are scattered into / gathered from group-major panels (pad columns are zero: a zero column is
// inert in every kernel of the iteration).  RICADI_WIDE_SPLIT=0 keeps the wide panels whole.
static void gmres_core_any(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, size_t gsb, double* x,
                           int m, bool lowrank, GmresResult* res, bool have_x0, const std::vector<int>* only,
                           bool allow_stall) {
  const int W0 = wide_split_width(c, m);
  if (!W0) {
    gmres_core(c, sds, G, b, gsb, x, m, lowrank, res, have_x0, only, allow_stall);
    return;
  }
  hipStream_t st = c->st;
  const int n = c->n;
  const size_t nm = (size_t)n * m;
  std::vector<int> todo;
  if (only) todo = *only;
  else
    for (int g = 0; g < G; ++g) todo.push_back(g);
  for (int g = 0; g < G; ++g) res[g] = GmresResult();
  for (int s : todo) res[s].converged = true;
  // (ensure_work of the caller reserved a full chunk; growing the workspace here would free the buffer b lives in)
  if (c->wcols < W0 * RICADI_MAX_GROUPS) throw HipError{"workspace not sized for the column groups of a wide panel"};
  c->split_b.ensure((size_t)n * W0 * RICADI_MAX_GROUPS);
  c->split_x.ensure((size_t)n * W0 * RICADI_MAX_GROUPS);
  // columns [col0, col0 + ncols) of every panel as groups of W columns
  auto run_pass = [&](int col0, int ncols, int W) {
    const int ncg = (ncols + W - 1) / W;
    const int per = std::max(1, RICADI_MAX_GROUPS / ncg);
    const size_t nmw = (size_t)n * W;
    // chunks of equal size (16 shifts, 3 per chunk: 3 3 3 3 2 2 rather than 3 3 3 3 3 1); the caller's order is
    // kept: neighbouring shifts of a sorted list need similar iteration counts, which is what a lockstep batch wants
    const int nchunk = ((int)todo.size() + per - 1) / per;
    size_t at = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
      const int cnt = ((int)todo.size() - (int)at + (nchunk - ch) - 1) / (nchunk - ch);
      const int Gv = cnt * ncg;
      std::vector<ShiftData*> vsds(Gv);
      if (ncg * W != ncols) {
        HIPCHK(hipMemsetAsync(c->split_b.p, 0, sizeof(double) * nmw * Gv, st));
        if (have_x0) HIPCHK(hipMemsetAsync(c->split_x.p, 0, sizeof(double) * nmw * Gv, st));
      }
      for (int k = 0; k < cnt; ++k) {
        const int s = todo[at + k];
        for (int cg = 0; cg < ncg; ++cg) {
          const int v = k * ncg + cg, w = std::min(W, ncols - cg * W), sc = col0 + cg * W;
          vsds[v] = sds[s];
          launch_copy_cols(st, n, w, b + (size_t)s * gsb, m, sc, c->split_b.p + (size_t)v * nmw, W, 0, 1.0);
          if (have_x0)
            launch_copy_cols(st, n, w, x + (size_t)s * nm, m, sc, c->split_x.p + (size_t)v * nmw, W, 0, 1.0);
        }
      }
      std::vector<GmresResult> vres(Gv);
      gmres_core(c, vsds.data(), Gv, c->split_b.p, nmw, c->split_x.p, W, lowrank, vres.data(), have_x0, nullptr,
                 allow_stall);
      for (int k = 0; k < cnt; ++k) {
        const int s = todo[at + k];
        GmresResult& r = res[s];
        for (int cg = 0; cg < ncg; ++cg) {
          const int v = k * ncg + cg, w = std::min(W, ncols - cg * W);
          launch_copy_cols(st, n, w, c->split_x.p + (size_t)v * nmw, W, 0, x + (size_t)s * nm, m, col0 + cg * W, 1.0);
          r.iters = std::max(r.iters, vres[v].iters);
          r.converged = r.converged && vres[v].converged;
          r.stalled = r.stalled || vres[v].stalled;
          r.max_relres = std::max(r.max_relres, vres[v].max_relres);
        }
      }
      at += cnt;
    }
  };
  // (a remainder of up to 8 columns -- m = 66 = 4 x 16 + 2 -- as one more batch of 8-column groups over all
  // shifts instead of a fifth sixteen-column group per shift was measured at n = 1e5: 2172 vs 2176 ms per pass
  // over 64 shifts; the sweeps of an 8-column batch cost what those of a 16-column one do -- the block inverses
  // they read are as many bytes as the panels)
  run_pass(0, m, W0);
}

// ---- recycled right-hand sides (ricadi_set_recycle) ---------------------------------------------
// Initial guesses  x_g = sum_e Y_{g,e} C_e  from the stored pairs (B_e, Y_{g,e}),  S_g Y_{g,e} = B_e, with
// C = argmin || b - [B_e] C ||_F  (normal equations on the matrix cores, rank-revealing Cholesky on the
// host).  b: the right-hand side shared by the groups (n x m, pressure rows zero).  Returns false when no
// stored panel is common to all groups (x is not touched then).
static bool recycle_guess(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, int m, double* x) {
  std::vector<const ricadi_ctx::RecB*> ent;
  for (auto& e : c->rec_ring) {
    if (!e || e->serial < 0) continue;
    bool all = true;
    for (int g = 0; g < G && all; ++g) {
      bool has = false;
      for (auto& y : sds[g]->rec)
        if (y && y->serial == e->serial && y->w == e->w) has = true;
      all = has;
    }
    if (all) ent.push_back(e.get());
  }
  if (ent.empty()) return false;
  int h = 0;
  for (auto* e : ent) h += e->w;
  hipStream_t st = c->st;
  const int nv = c->nv, n = c->n, hw = h + m;
  TArr<double> Gd(c->pool), Yd(c->pool, (size_t)h * m);
  std::vector<double> Ghh((size_t)h * h), Ghb((size_t)h * m), Y;
  int r0 = 0;
  // slot of every entry in the side-by-side panel (all of the panel's width, ring of at most 8 slots)
  std::vector<int> slot_of(ent.size(), -1);
  bool pan = c->rec_pan_w == m && c->rec_pan.p && c->rec_ring.size() <= 8;
  for (size_t i = 0; i < ent.size() && pan; ++i) {
    for (size_t si = 0; si < c->rec_ring.size(); ++si)
      if (c->rec_ring[si].get() == ent[i]) slot_of[i] = (int)si;
    pan = slot_of[i] >= 0 && ent[i]->w == m;
  }
  if (pan) {
    // Gram matrix of ALL slots and their products with b in two launches; the live entries are picked on the host
    const int H = 8 * m, Hw = H + m;
    Gd.alloc((size_t)H * Hw);
    HIPCHK(hipMemsetAsync(Gd.p, 0, sizeof(double) * H * Hw, st));
    launch_gemm_tn(st, nv, H, H, c->rec_pan.p, H, c->rec_pan.p, H, Gd.p, Hw);
    launch_gemm_tn(st, nv, H, m, c->rec_pan.p, H, b, m, Gd.p + H, Hw);
    std::vector<double> Gh((size_t)H * Hw);
    HIPCHK(hipMemcpyAsync(Gh.data(), Gd.p, sizeof(double) * Gh.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (size_t ei = 0; ei < ent.size(); ++ei)
      for (int a = 0; a < m; ++a) {
        const int i = (int)ei * m + a, gi = slot_of[ei] * m + a;
        for (size_t ej = 0; ej < ent.size(); ++ej)
          for (int bcol = 0; bcol < m; ++bcol)
            Ghh[(size_t)i * h + ej * m + bcol] = Gh[(size_t)gi * Hw + slot_of[ej] * m + bcol];
        for (int j = 0; j < m; ++j) Ghb[(size_t)i * m + j] = Gh[(size_t)gi * Hw + H + j];
      }
  } else {
    Gd.alloc((size_t)h * hw);
    HIPCHK(hipMemsetAsync(Gd.p, 0, sizeof(double) * h * hw, st));
    for (size_t i = 0; i < ent.size(); ++i) {
      int c0 = r0;
      for (size_t j = i; j < ent.size(); ++j) {
        launch_gemm_tn(st, nv, ent[i]->w, ent[j]->w, ent[i]->b.p, ent[i]->w, ent[j]->b.p, ent[j]->w,
                       Gd.p + (size_t)r0 * hw + c0, hw);
        c0 += ent[j]->w;
      }
      launch_gemm_tn(st, nv, ent[i]->w, m, ent[i]->b.p, ent[i]->w, b, m, Gd.p + (size_t)r0 * hw + h, hw);
      r0 += ent[i]->w;
    }
    std::vector<double> Gh((size_t)h * hw);
    HIPCHK(hipMemcpyAsync(Gh.data(), Gd.p, sizeof(double) * Gh.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < h; ++i) {
      for (int j = 0; j < h; ++j) Ghh[(size_t)i * h + j] = j >= i ? Gh[(size_t)i * hw + j] : Gh[(size_t)j * hw + i];
      for (int j = 0; j < m; ++j) Ghb[(size_t)i * m + j] = Gh[(size_t)i * hw + h + j];
    }
  }
  // the diagonal blocks come from a symmetric kernel, the off-diagonal ones were computed above the
  // diagonal only: the mirror image is exact
  const int rank = gram_lstsq_scaled(h, m, Ghh, Ghb, 1e-11, Y);
  if (rank == 0) return false;
  HIPCHK(hipMemcpyAsync(Yd.p, Y.data(), sizeof(double) * h * m, hipMemcpyHostToDevice, st));
  GroupTab all{};
  all.ng = G;
  for (int g = 0; g < G; ++g) all.gid[g] = g;
  r0 = 0;
  for (size_t i = 0; i < ent.size(); ++i) {
    GroupPtrs A = same_ptr((const double*)nullptr);
    for (int g = 0; g < G; ++g)
      for (auto& y : sds[g]->rec)
        if (y && y->serial == ent[i]->serial && y->w == ent[i]->w) A.p[g] = y->y.p;
    launch_gemm_nn_bp(st, all, n, ent[i]->w, m, A, ent[i]->w, Yd.p + (size_t)r0 * m, m, 0, x, m, (size_t)n * m,
                      1.0, i == 0 ? 0.0 : 1.0);
    r0 += ent[i]->w;
  }
  HIPCHK(hipStreamSynchronize(st));   // Y is a stack object
  if (c->opts.verbose > 1) fprintf(stderr, "[ricadi] recycled guess from %d stored columns (rank %d)\n", h, rank);
  return true;
}

static void recycle_store(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, int m, const double* x) {
  hipStream_t st = c->st;
  const int depth = c->rec_depth;
  ricadi_ctx::RecB* slot = nullptr;
  if ((int)c->rec_ring.size() < depth) {
    c->rec_ring.emplace_back(new ricadi_ctx::RecB);
    slot = c->rec_ring.back().get();
  } else {
    for (auto& e : c->rec_ring)
      if (!slot || e->serial < slot->serial) slot = e.get();
  }
  slot->serial = ++c->rec_serial;
  slot->w = m;
  slot->b.ensure((size_t)c->nv * m);
  HIPCHK(hipMemcpyAsync(slot->b.p, b, sizeof(double) * c->nv * m, hipMemcpyDeviceToDevice, st));
  {
    // side-by-side copy (slots of another width invalidate the panel: recycle_guess then takes the pairwise path)
    int si = 0;
    for (; si < (int)c->rec_ring.size(); ++si)
      if (c->rec_ring[si].get() == slot) break;
    if (c->rec_pan_w != m || c->rec_pan.n < (size_t)c->nv * 8 * m) {
      c->rec_pan.ensure((size_t)c->nv * 8 * m);
      HIPCHK(hipMemsetAsync(c->rec_pan.p, 0, sizeof(double) * (size_t)c->nv * 8 * m, st));
      c->rec_pan_w = m;
      for (auto& e : c->rec_ring)
        if (e.get() != slot && e->serial >= 0 && e->w == m)
          launch_copy_cols(st, c->nv, m, e->b.p, m, 0, c->rec_pan.p, 8 * m, (int)(&e - &c->rec_ring[0]) * m, 1.0);
    }
    if (si < 8) launch_copy_cols(st, c->nv, m, b, m, 0, c->rec_pan.p, 8 * m, si * m, 1.0);
  }
  auto live = [&](long serial) {
    for (auto& e : c->rec_ring)
      if (e->serial == serial) return true;
    return false;
  };
  const size_t nm = (size_t)c->n * m;
  for (int g = 0; g < G; ++g) {
    ShiftData::RecY* y = nullptr;
    for (auto& r : sds[g]->rec)
      if (!live(r->serial)) y = r.get();          // a solution whose right-hand side has left the ring
    if (!y && (int)sds[g]->rec.size() < depth) {
      sds[g]->rec.emplace_back(new ShiftData::RecY);
      y = sds[g]->rec.back().get();
    }
    if (!y)
      for (auto& r : sds[g]->rec)
        if (!y || r->serial < y->serial) y = r.get();
    y->serial = slot->serial;
    y->w = m;
    y->y.ensure(nm);
    HIPCHK(hipMemcpyAsync(y->y.p, x + (size_t)g * nm, sizeof(double) * nm, hipMemcpyDeviceToDevice, st));
  }
}

// Storage of the Krylov basis / the preconditioner inverses for the solves inside the scope:
//   level 1: FP32-stored basis, FP64 inverses;  level 2: FP64-stored basis, FP64 inverses
// (level 0 = the context's defaults: FP16 / FP32 basis by size, FP32 inverses).  All levels of a
// multilevel preconditioner follow.  The arithmetic is FP64 at every level.
struct StorageScope {
  ricadi_ctx* c;
  bool b16, b32;
  std::vector<bool> p32;
  StorageScope(ricadi_ctx* ctx, int level) : c(ctx), b16(ctx->basis16), b32(ctx->basis32) {
    for (ricadi_ctx* l = c; l; l = l->child.get()) {
      p32.push_back(l->precond32);
      l->precond32 = false;
    }
    c->basis16 = false;
    if (level >= 2) {
      c->basis32 = false;
      c->basis.ensure((size_t)(c->wrestart + 1) * c->n * c->wcols);
    }
  }
  ~StorageScope() {
    size_t i = 0;
    for (ricadi_ctx* l = c; l; l = l->child.get()) l->precond32 = p32[i++];
    c->basis16 = b16;
    c->basis32 = b32;
  }
};
static int storage_level(const ricadi_ctx* c) {
  if (!c->precond32 && !c->basis32) return 2;
  if (!c->precond32 && !c->basis16) return 1;
  return 0;
}

// The batched solve as the drivers call it: recycled initial guess (shared right-hand side, plain
// operator), the lockstep GMRES, the storage safety net -- a group that stops at gmres_maxit or
// stagnates is continued from its iterate with the FP32- and then the FP64-stored basis and FP64
// preconditioner inverses (counted in c->escalations) -- true residuals on request.
static void gmres_solve_batch(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b,
                              size_t gsb, double* x, int m, bool lowrank, double* relres_host,
                              GmresResult* res) {
  const bool no_net = false;
  hipStream_t st = c->st;
  const bool plain = !(lowrank && c->q > 0);
  const bool shared = (gsb == 0 || G == 1) && plain && c->rec_depth > 0;
  Tick tkg;
  const bool guess = shared && recycle_guess(c, sds, G, b, m, x);
  if (c->timing) {
    (void)hipStreamSynchronize(st);
    c->t_guess += tkg.lap();
  }
  const int lvl0 = storage_level(c);
  gmres_core_any(c, sds, G, b, gsb, x, m, lowrank, res, guess, nullptr, !no_net && lvl0 < 2);
  std::vector<int> bad;
  for (int g = 0; g < G; ++g)
    if (!res[g].converged) bad.push_back(g);
  for (int level = lvl0 + 1; level <= 2 && !bad.empty() && !no_net; ++level) {
    StorageScope wide(c, level);
    std::vector<GmresResult> r2(G);
    gmres_core_any(c, sds, G, b, gsb, x, m, lowrank, r2.data(), true, &bad, level < 2);
    c->escalations += (long)bad.size();
    std::vector<int> still;
    for (int g : bad) {
      if (c->opts.verbose)
        fprintf(stderr, "[ricadi] shift (%g, %g): %s after %d iterations at relres %.2e -> storage level %d: %d more, %.2e\n",
                sds[g]->alpha, sds[g]->beta, res[g].stalled ? "stagnation" : "gmres_maxit", res[g].iters,
                res[g].max_relres, level, r2[g].iters, r2[g].max_relres);
      res[g].iters += r2[g].iters;
      res[g].converged = r2[g].converged;
      res[g].stalled = r2[g].stalled;
      res[g].max_relres = r2[g].max_relres;
      if (!r2[g].converged) still.push_back(g);
    }
    bad.swap(still);
  }
  if (relres_host) {
    // true residuals
    Batch bt = make_batch(c, sds, G, m);
    const size_t nm = bt.gs;
    const size_t gspart = (size_t)dots_num_blocks(c->n) * (c->opts.gmres_restart + 2) * m;
    const int GM = G * m;
    double* hb = c->h_resid;
    op_apply(c, bt, x, nm, c->wv.p, lowrank);
    launch_axpby_b(st, bt.tab, nm, 1.0, b, gsb, -1.0, c->wv.p, nm);
    launch_cols_dots_b(st, bt.tab, c->n, m, 0, (const double*)nullptr, 0, 0, c->wv.p, nm, 1, c->partial.p,
                       gspart, c->nrm2.p, (size_t)m);
    launch_cols_dots_b(st, bt.tab, c->n, m, 0, (const double*)nullptr, 0, 0, b, gsb, 1, c->partial.p,
                       gspart, c->bnorm2.p, (size_t)m);
    HIPCHK(hipMemcpyAsync(hb, c->nrm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hb + GM, c->bnorm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < GM; ++j)
      relres_host[j] = hb[GM + j] > 0.0 ? std::sqrt(std::max(hb[j], 0.0) / hb[GM + j]) : 0.0;
  }
  if (shared) recycle_store(c, sds, G, b, m, x);
  for (int g = 0; g < G; ++g) c->total_iters += res[g].iters;
  c->total_solves += G;
}

// In-place inverse of a small dense matrix on the host (Gauss-Jordan, partial pivoting).
static bool host_invert(std::vector<double>& a, int q) {
  std::vector<double> inv((size_t)q * q, 0.0);
  for (int i = 0; i < q; ++i) inv[(size_t)i * q + i] = 1.0;
  double amax = 0.0;
  for (double v : a) amax = std::max(amax, std::fabs(v));
  for (int k = 0; k < q; ++k) {
    int p = k;
    for (int i = k + 1; i < q; ++i)
      if (std::fabs(a[(size_t)i * q + k]) > std::fabs(a[(size_t)p * q + k])) p = i;
    const double piv = a[(size_t)p * q + k];
    if (!(std::fabs(piv) > 1e-12 * amax)) return false;
    if (p != k)
      for (int j = 0; j < q; ++j) {
        std::swap(a[(size_t)k * q + j], a[(size_t)p * q + j]);
        std::swap(inv[(size_t)k * q + j], inv[(size_t)p * q + j]);
      }
    for (int j = 0; j < q; ++j) {
      a[(size_t)k * q + j] /= piv;
      inv[(size_t)k * q + j] /= piv;
    }
    for (int i = 0; i < q; ++i) {
      if (i == k) continue;
      const double f = a[(size_t)i * q + k];
      if (f == 0.0) continue;
      for (int j = 0; j < q; ++j) {
        a[(size_t)i * q + j] -= f * a[(size_t)k * q + j];
        inv[(size_t)i * q + j] -= f * inv[(size_t)k * q + j];
      }
    }
  }
  a.swap(inv);
  return true;
}

// Relative true residuals ||b - (S - U V^T) x|| / ||b|| per column (G*m values, host);
// the residual panels are left in c->wv.
static void true_relres(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, size_t gsb,
                        const double* x, int m, bool lowrank, double* out) {
  hipStream_t st = c->st;
  Batch bt = make_batch(c, sds, G, m);
  const size_t nm = bt.gs;
  const size_t gspart = (size_t)dots_num_blocks(c->n) * (c->opts.gmres_restart + 2) * m;
  const int GM = G * m;
  double* hb = c->h_resid;
  op_apply(c, bt, x, nm, c->wv.p, lowrank);
  launch_axpby_b(st, bt.tab, nm, 1.0, b, gsb, -1.0, c->wv.p, nm);
  launch_cols_dots_b(st, bt.tab, c->n, m, 0, (const double*)nullptr, 0, 0, c->wv.p, nm, 1, c->partial.p,
                     gspart, c->nrm2.p, (size_t)m);
  launch_cols_dots_b(st, bt.tab, c->n, m, 0, (const double*)nullptr, 0, 0, b, gsb, 1, c->partial.p,
                     gspart, c->bnorm2.p, (size_t)m);
  HIPCHK(hipMemcpyAsync(hb, c->nrm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(hb + GM, c->bnorm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int j = 0; j < GM; ++j)
    out[j] = hb[GM + j] > 0.0 ? std::sqrt(std::max(hb[j], 0.0) / hb[GM + j]) : 0.0;
}

// Batched solve with the low-rank term  (S_g - U V^T) x_g = b_g.
//
// Default: Sherman-Morrison-Woodbury, as the reference's lau.solve_sadpnt_smw does --
// GMRES runs on the plain saddle operator (no thin GEMMs inside the iteration), and
//   x = y + W (V^T y),   y = S^-1 b,   W = S^-1 [U;0] (I - V^T S^-1 U)^-1 .
// W_g is cached per shift and low-rank term; a batch that meets a shift without it
// solves the augmented panels [b_g, U] (m + q columns) once.  The closed-loop residual
// is then verified in FP64; columns above the tolerance (ill-conditioned capacitance
// matrix) are refined by one GMRES on the closed-loop operator itself.
static void solve_batch(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, size_t gsb,
                        double* x, int m, bool lowrank, double* relres_host, GmresResult* res) {
  const int q = c->q;
  if (!lowrank || q <= 0 || !c->smw || m + q > RICADI_MAX_M) {
    gmres_solve_batch(c, sds, G, b, gsb, x, m, lowrank && q > 0, relres_host, res);
    return;
  }
  hipStream_t st = c->st;
  const int n = c->n, nv = c->nv, np = c->np;
  const size_t nm = (size_t)n * m;
  const double tol = c->opts.gmres_tol;
  bool need = false;
  for (int g = 0; g < G; ++g) need = need || sds[g]->smw_epoch != c->lr_epoch;
  GroupTab all{};
  all.ng = G;
  for (int g = 0; g < G; ++g) all.gid[g] = g;
  bool bad = false;
  // U = columns [ucol, ucol + q) of the (shared) right-hand side: S^-1 U is part of the plain solution,
  // no augmented columns needed (first sweep of a Newton step without mtxoldb: rhs = [W, K_k], U = K_k)
  const int ucol = c->lr_ucol;
  c->lr_ucol = -1;             // the hint holds for one solve
  const bool dup = need && (gsb == 0 || G == 1) && ucol >= 0 && ucol + q <= m;
  if (need) {
    const int ma = dup ? m : m + q;
    const size_t nma = (size_t)n * ma;
    double* xa;
    int xoff;     // column of S^-1 U inside the solution panels xa (leading dimension ma)
    if (dup) {
      gmres_solve_batch(c, sds, G, b, gsb, x, m, false, nullptr, res);
      xa = x;
      xoff = ucol;
    } else {
      // augmented panels [b_g, U]; one panel for all groups when they share b
      const int nra = gsb == 0 ? 1 : G;
      c->smw_rhs.ensure(nma * nra);
      c->smw_x.ensure(nma * G);
      double* ra = c->smw_rhs.p;
      xa = c->smw_x.p;
      xoff = m;
      for (int g = 0; g < nra; ++g) {
        launch_copy_cols(st, n, m, b + (size_t)g * gsb, m, 0, ra + g * nma, ma, 0, 1.0);
        launch_copy_cols(st, nv, q, c->U.p, q, 0, ra + g * nma, ma, m, 1.0);
        if (np > 0)
          HIPCHK(hipMemset2DAsync(ra + g * nma + (size_t)nv * ma + m, sizeof(double) * ma, 0,
                                  sizeof(double) * q, np, st));
      }
      gmres_solve_batch(c, sds, G, ra, gsb == 0 ? 0 : nma, xa, ma, false, nullptr, res);
    }
    // capacitance matrices I - V^T (S^-1 U)
    c->smw_cap.ensure((size_t)G * q * q);
    HIPCHK(hipMemsetAsync(c->smw_cap.p, 0, sizeof(double) * G * q * q, st));
    launch_gemm_tn_b(st, all, nv, q, q, c->V.p, q, xa + xoff, ma, nma, c->smw_cap.p, q, (size_t)q * q);
    std::vector<double> caps((size_t)G * q * q);
    HIPCHK(hipMemcpyAsync(caps.data(), c->smw_cap.p, sizeof(double) * caps.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int g = 0; g < G && !bad; ++g) {
      std::vector<double> cap((size_t)q * q);
      for (int i = 0; i < q; ++i)
        for (int j = 0; j < q; ++j)
          cap[(size_t)i * q + j] = (i == j ? 1.0 : 0.0) - caps[(size_t)g * q * q + (size_t)i * q + j];
      if (!host_invert(cap, q) || !res[g].converged) bad = true;
      std::copy(cap.begin(), cap.end(), caps.begin() + (size_t)g * q * q);
    }
    if (!dup)
      for (int g = 0; g < G; ++g)
        launch_copy_cols(st, n, m, xa + g * nma, ma, 0, x + g * nm, m, 0, 1.0);
    if (!bad) {
      HIPCHK(hipMemcpyAsync(c->smw_cap.p, caps.data(), sizeof(double) * caps.size(), hipMemcpyHostToDevice, st));
      for (int g = 0; g < G; ++g) {
        ShiftData* sd = sds[g];
        if (sd->smw_w.n != (size_t)n * q) sd->smw_w.alloc((size_t)n * q);
        launch_gemm_nn(st, n, q, q, xa + g * nma + xoff, ma, c->smw_cap.p + (size_t)g * q * q, q,
                       sd->smw_w.p, q, 1.0, 0.0);
        sd->smw_epoch = c->lr_epoch;
      }
      HIPCHK(hipStreamSynchronize(st));   // caps is a stack object
    }
  } else {
    gmres_solve_batch(c, sds, G, b, gsb, x, m, false, nullptr, res);
  }
  const size_t gsq = (size_t)q * m;
  if (!bad) {
    // x_g += W_g (V^T x_g)
    GroupPtrs W = same_ptr((const double*)nullptr);
    for (int g = 0; g < G; ++g) W.p[g] = sds[g]->smw_w.p;
    HIPCHK(hipMemsetAsync(c->lrc.p, 0, sizeof(double) * gsq * G, st));
    launch_gemm_tn_b(st, all, nv, q, m, c->V.p, q, x, m, nm, c->lrc.p, m, gsq);
    launch_gemm_nn_bp(st, all, n, q, m, W, q, c->lrc.p, m, gsq, x, m, nm, 1.0, 1.0);
  }
  // verification on the closed-loop operator, refinement where needed
  std::vector<double> rr((size_t)G * m);
  true_relres(c, sds, G, b, gsb, x, m, true, rr.data());
  bool ok = true;
  for (double v : rr) ok = ok && v <= tol;
  if (!ok) {
    c->smw_rhs.ensure(nm * G);
    c->smw_x.ensure(nm * G);
    HIPCHK(hipMemcpyAsync(c->smw_rhs.p, c->wv.p, sizeof(double) * nm * G, hipMemcpyDeviceToDevice, st));
    std::vector<GmresResult> r2(G);
    // residual equation on the closed-loop operator; its tolerance is relative to ||r||
    double worst = 0.0;
    for (double v : rr) worst = std::max(worst, v);
    {
      Restore<double> keep_tol(c->opts.gmres_tol);
      c->opts.gmres_tol = std::min(0.5, std::max(1e-14, 0.5 * tol / worst));
      gmres_solve_batch(c, sds, G, c->smw_rhs.p, nm, c->smw_x.p, m, true, nullptr, r2.data());
    }
    launch_axpby_b(st, all, nm, 1.0, c->smw_x.p, nm, 1.0, x, nm);
    for (int g = 0; g < G; ++g) res[g].iters += r2[g].iters;
    true_relres(c, sds, G, b, gsb, x, m, true, rr.data());
  }
  for (int g = 0; g < G; ++g) {
    double w = 0.0;
    for (int j = 0; j < m; ++j) w = std::max(w, rr[(size_t)g * m + j]);
    res[g].max_relres = w;
    res[g].converged = w <= tol * 1.0000001;
  }
  if (relres_host) std::copy(rr.begin(), rr.end(), relres_host);
}

static GmresResult gmres_solve(ricadi_ctx* c, ShiftData* sd, const double* b, double* x, int m,
                               bool lowrank, double* relres_host) {
  GmresResult r;
  solve_batch(c, &sd, 1, b, (size_t)c->n * m, x, m, lowrank, relres_host, &r);
  return r;
}

// rhs panel (n x m) from an NV x m device block (pressure rows zero)
static void load_rhs(ricadi_ctx* c, const double* dR, int m, double* b) {
  HIPCHK(hipMemcpyAsync(b, dR, sizeof(double) * (size_t)c->nv * m, hipMemcpyDeviceToDevice, c->st));
  if (c->np > 0)
    HIPCHK(hipMemsetAsync(b + (size_t)c->nv * m, 0, sizeof(double) * (size_t)c->np * m, c->st));
}

// Per-shift data of the ADI shifts an iteration is about to use -- and of the projection
// operator (alpha, beta) = (1, 0) when `with_projection` -- built in ONE setup pass: the
// coarse matrices of all of them go through the same batched factorisation (a matrix set
// up alone costs ~8x its share of a batch of 16).
static void prefetch_setup(ricadi_ctx* c, const double* shifts, int nuse, bool with_projection) {
  std::vector<double> al, be;
  if (with_projection && c->np > 0) {
    al.push_back(1.0);
    be.push_back(0.0);
  }
  for (int i = 0; i < nuse; ++i) {
    al.push_back(shifts[i]);
    be.push_back(1.0);
  }
  if (al.empty()) return;
  std::vector<ShiftData*> sds(al.size());
  get_shifts(c, al.data(), be.data(), (int)al.size(), sds.data());
}

// W (NV x m, device, in place) <- P^T W  through one saddle solve with cal E
static void project_panel(ricadi_ctx* c, double* dW, int m) {
  if (c->np == 0) return;
  ShiftData* sd = get_shift(c, 1.0, 0.0);
  ensure_work(c, m);
  load_rhs(c, dW, m, c->bvec.p);
  GmresResult r = gmres_solve(c, sd, c->bvec.p, c->xs.p, m, false, nullptr);
  if (!r.converged) throw HipError{"projection solve did not converge"};
  launch_spmm(c->st, c->nv, c->E.rp.p, c->E.ci.p, c->E.v.p, c->xs.p, m, nullptr, dW, m, nullptr, 0,
              1.0, 0.0, nullptr, m);
}

struct DScalar {
  // tiny helper: Frobenius norm of W^T W and ||W||_F^2 of a device panel
  static void gram_norms(ricadi_ctx* c, const double* dW, int nrows, int m, double* gram_fro,
                         double* nrm2) {
    DArr<double>& G = c->scratch;
    G.ensure((size_t)m * m + 64);
    HIPCHK(hipMemsetAsync(G.p, 0, sizeof(double) * m * m, c->st));
    launch_gemm_tn(c->st, nrows, m, m, dW, m, dW, m, G.p, m);
    std::vector<double> h((size_t)m * m);
    HIPCHK(hipMemcpyAsync(h.data(), G.p, sizeof(double) * m * m, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    double f = 0.0, t = 0.0;
    for (int i = 0; i < m; ++i) {
      t += h[(size_t)i * m + i];
      for (int j = 0; j < m; ++j) f += h[(size_t)i * m + j] * h[(size_t)i * m + j];
    }
    if (gram_fro) *gram_fro = std::sqrt(f);
    if (nrm2) *nrm2 = t;
  }
};

static int compress_dev(ricadi_ctx* c, const double* dZ, int cz, int ldz, double thresh, int kmax,
                        bool thresh_relative, double* dOut, std::vector<double>* sv_host,
                        bool use_qr = false);
static void block_qr_dev(ricadi_ctx* c, const double* D, int ldd, int n, int kk, double* Q,
                         double* R, int split = 0);

// Truncation level of the internal recompressions: the Gram-matrix route
// resolves singular values down to sqrt(eps)*sigma_1; dropping what lies below
// changes Z Z^T by at most eps*||Z Z^T|| -- rounding level.
static const double kInternalRelThresh = 3e-8;

static Exec main_exec(ricadi_ctx* c);
static int recompress_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz, double rel,
                           double* dOut);

// Recompress the device factor in place (columns [0, zc) of c->Z).
static void factor_recompress(ricadi_ctx* c) {
  if (c->zc == 0) return;
  TArr<double> tmp(c->pool, (size_t)c->nv * c->zc);
  const int k = recompress_exec(c, main_exec(c), c->Z.p, c->zc, c->zld, kInternalRelThresh, tmp.p);
  if (k > 0) launch_copy_cols(c->st, c->nv, k, tmp.p, k, 0, c->Z.p, c->zld, 0, 1.0);
  HIPCHK(hipStreamSynchronize(c->st));
  c->zc = k;
}

static int compress_gram_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz,
                              double thresh, int kmax, bool thresh_relative, double* dOut,
                              std::vector<double>* sv_host);
static int recompress_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz, double rel,
                           double* dOut);

// Auxiliary stream + handle for work that runs beside the main stream (created on first use).
static Exec aux_exec(ricadi_ctx* c) {
  if (!c->st2) {
    HIPCHK(hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
    RBCHK(rocblas_create_handle(&c->rb2));
    RBCHK(rocblas_set_stream(c->rb2, c->st2));
    c->info2.alloc(4);
    HIPCHK(hipEventCreateWithFlags(&c->ev_z, hipEventDisableTiming));
  }
  Exec ex;
  ex.st = c->st2;
  ex.rb = c->rb2;
  ex.pool = &c->pool2;
  ex.info = c->info2.p;
  return ex;
}

// In-ADI recompression that does not stall the sweeps: the columns [0, snap) of the factor are
// compressed on the auxiliary stream by a helper thread (same arithmetic as
// factor_recompress) while the main stream goes on appending columns behind them; finish()
// splices the result in:  Z <- [compressed prefix | columns appended meanwhile].
// Member order matters: `fut` is destroyed first and waits for the helper, then `out`.
struct AsyncRecompress {
  ricadi_ctx* c;
  TArr<double> out;
  int snap = 0;
  bool active = false;
  std::future<int> fut;
  explicit AsyncRecompress(ricadi_ctx* ctx) : c(ctx), out(ctx->pool) {}
  void start() {
    if (active || c->zc == 0) return;
    const Exec ex = aux_exec(c);
    snap = c->zc;
    out.alloc((size_t)c->nv * snap);
    HIPCHK(hipEventRecord(c->ev_z, c->st));            // the prefix is complete on the main stream
    HIPCHK(hipStreamWaitEvent(c->st2, c->ev_z, 0));
    ricadi_ctx* cc = c;
    const double* Zp = c->Z.p;
    const int ld = c->zld, sn = snap, dev = c->dev;
    double* op = out.p;
    fut = std::async(std::launch::async, [cc, ex, Zp, ld, sn, dev, op]() {
      (void)hipSetDevice(dev);
      return recompress_exec(cc, ex, Zp, sn, ld, kInternalRelThresh, op);
    });
    active = true;
  }
  void finish() {
    if (!active) return;
    active = false;
    const int k = fut.get();                            // the auxiliary stream is drained in there
    hipStream_t st = c->st;
    const int nv = c->nv, tail = c->zc - snap;
    if (tail > 0) {
      TArr<double> tmp(c->pool, (size_t)nv * tail);
      launch_copy_cols(st, nv, tail, c->Z.p, c->zld, snap, tmp.p, tail, 0, 1.0);
      launch_copy_cols(st, nv, tail, tmp.p, tail, 0, c->Z.p, c->zld, k, 1.0);
    }
    if (k > 0) launch_copy_cols(st, nv, k, out.p, k, 0, c->Z.p, c->zld, 0, 1.0);
    c->zc = k + tail;
    out.release();
  }
};

// ---- low-rank ADI (device resident) -------------------------------------------------
struct AdiStats {
  int steps = 0;
  double rel = 0.0;
  long gmres_iters = 0;
  long shift_solves = 0;
  double res_fro = 0.0;
  long nonconverged = 0;      // shift-solves that hit gmres_maxit above the tolerance
  double worst_relres = 0.0;
  int sweeps = 0;             // sweep form: batched sweeps run (= all-gathers when sharded)
};

// dW: NV x m device panel (overwritten by the final residual factor).
// Appends sqrt(-2p) V_i to c->Z (ld = c->zld) starting at column c->zc.
// Sweep form of the same ADI: G consecutive steps
// with distinct shifts are G independent solves against the SAME residual factor,
//   S(p_g) [U_g; *] = [W; 0],
// recombined with the G x G Cauchy matrix C_ij = -1/(p_i+p_j) = R^T R:
//   Z-block = U (R^-1 (x) I),   W <- W + E U ((C^-1 1) (x) I)
// -- identical to the G sequential steps up to a rotation of the block's columns (Z Z^T
// and the gain are the same).  The G solves go through ONE batched lockstep GMRES, which
// is what fills the GPU at n ~ 3e4.  The stopping rule is applied per sweep (mean block
// norm).  Returns false (nothing done) if the shift list does not allow sweeps.
// All-gather of `count` doubles per rank through the host's collective (ricadi_set_exchange): the ranks'
// first `count` doubles of c->xsend arrive rank-major in c->xrecv.  The context stream is drained first.
// The last RICADI_XCTL bytes of the send buffer (and the last world * RICADI_XCTL of the receive buffer) are
// kept for the small control messages (decisions, statistics), so that they never touch panels in flight.
#define RICADI_XCTL 4096
static size_t exchange_panel_capacity(const ricadi_ctx* c) { return c->xcap > RICADI_XCTL ? c->xcap - RICADI_XCTL : 0; }
static void exchange_at(ricadi_ctx* c, double* send, double* recv, size_t count) {
  ++c->xcount;
  if (c->xcomm) {
    // RCCL: stream ordered behind the solves that filled `send`, ahead of the recombination that reads `recv`
    const ncclResult_t r = ncclAllGather(send, recv, count, ncclDouble, c->xcomm, c->st);
    if (r != ncclSuccess) throw HipError{std::string("ncclAllGather: ") + ncclGetErrorString(r)};
    return;
  }
  HIPCHK(hipStreamSynchronize(c->st));
  const int rc = c->xfn(c->xuser, send, recv, (int64_t)(count * sizeof(double)));
  if (rc != 0) throw HipError{"the all-gather callback of ricadi_set_exchange failed (" + std::to_string(rc) + ")"};
}
static void exchange(ricadi_ctx* c, size_t count) {
  if (count * sizeof(double) > exchange_panel_capacity(c))
    throw HipError{"exchange buffer too small: " + std::to_string(count * sizeof(double) + RICADI_XCTL) +
                   " bytes per rank needed, " + std::to_string(c->xcap) + " given to ricadi_set_exchange"};
  exchange_at(c, c->xsend, c->xrecv, count);
}
static double* ctl_send(ricadi_ctx* c) { return c->xsend + exchange_panel_capacity(c) / sizeof(double); }
static double* ctl_recv(ricadi_ctx* c) {
  return c->xrecv + (size_t)c->xworld * exchange_panel_capacity(c) / sizeof(double);
}
static bool sharded(const ricadi_ctx* c) { return (c->xworld > 1 || c->xforce) && (c->xfn != nullptr || c->xcomm != nullptr); }
// v[0..n) <- rank 0's values (decisions must not differ between the ranks: the norms they rest on come
// from kernels with atomic accumulation).  One tiny all-gather.
static void values_of_rank0(ricadi_ctx* c, double* v, int n) {
  if (!sharded(c)) return;
  if ((size_t)n * sizeof(double) > RICADI_XCTL) throw HipError{"control message too long"};
  HIPCHK(hipMemcpyAsync(ctl_send(c), v, sizeof(double) * n, hipMemcpyHostToDevice, c->st));
  exchange_at(c, ctl_send(c), ctl_recv(c), (size_t)n);
  HIPCHK(hipMemcpyAsync(v, ctl_recv(c), sizeof(double) * n, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
}
// v[0..nsum) <- sum over the ranks, v[nsum..nsum+nmax) <- maximum over the ranks (statistics)
static void reduce_over_ranks(ricadi_ctx* c, double* v, int nsum, int nmax) {
  if (!sharded(c)) return;
  const int n = nsum + nmax;
  if ((size_t)n * sizeof(double) > RICADI_XCTL) throw HipError{"control message too long"};
  HIPCHK(hipMemcpyAsync(ctl_send(c), v, sizeof(double) * n, hipMemcpyHostToDevice, c->st));
  exchange_at(c, ctl_send(c), ctl_recv(c), (size_t)n);
  std::vector<double> all((size_t)n * c->xworld);
  HIPCHK(hipMemcpyAsync(all.data(), ctl_recv(c), sizeof(double) * all.size(), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  for (int i = 0; i < n; ++i) {
    double t = all[i];
    for (int r = 1; r < c->xworld; ++r) {
      const double o = all[(size_t)r * n + i];
      t = i < nsum ? t + o : std::max(t, o);
    }
    v[i] = t;
  }
}

static bool lyap_adi_sweeps_dev(ricadi_ctx* c, const double* shifts, int ns, double* dW, int m,
                                const ricadi_adi_params& prm, AdiStats& stt) {
  int G = std::min(std::min(prm.sweep_width, ns), RICADI_MAX_GROUPS);
  while (G >= 2 && (prm.adi_max_steps < G || G * m > 2048)) G /= 2;
  if (G < 2) return false;
  for (int i = 0; i < ns; ++i)
    for (int j = i + 1; j < ns; ++j)
      if (shifts[i] == shifts[j]) return false;     // sweeps need distinct shifts
  // Cauchy data of every sweep of the cycle up front.  A numerically singular Cauchy matrix (the shifts of a
  // sweep too many / too close: 16 consecutive entries of a 128-shift list over 3.5 decades) halves the sweep
  // width until every sweep of the cycle is admissible -- as shift_parallel.py does --; only when no width
  // >= 2 is left does the caller go back to the sequential form
  int ncyc = 0;
  std::vector<std::vector<double>> rinvs, cinvs, pss;
  for (; G >= 2; G /= 2) {
    ncyc = ns / std::gcd(ns, G);                  // sweeps until the shift pattern repeats
    rinvs.assign(ncyc, {});
    cinvs.assign(ncyc, {});
    pss.assign(ncyc, {});
    bool ok = true;
    for (int sw = 0; sw < ncyc && ok; ++sw) {
      pss[sw].resize(G);
      for (int g = 0; g < G; ++g) pss[sw][g] = shifts[(sw * G + g) % ns];
      rinvs[sw].resize((size_t)G * G);
      cinvs[sw].resize(G);
      ok = cauchy_data(pss[sw].data(), G, rinvs[sw].data(), cinvs[sw].data()) == RICADI_OK;
    }
    if (ok) break;
  }
  if (G < 2) return false;
  hipStream_t st = c->st;
  const int n = c->n, nv = c->nv;
  const size_t nm = (size_t)n * m;
  // Shift-parallel form (ricadi_set_exchange): every rank owns a fixed subset of the shift list --
  // fixed, because the per-shift setup, the Sherman-Morrison-Woodbury panels and the recycled
  // solutions live with the owner -- and solves only its shifts of a sweep; one all-gather per sweep.
  const bool shard = sharded(c);
  const int world = shard ? c->xworld : 1, rank = shard ? c->xrank : 0;
  std::vector<int32_t> owner(ns, 0);
  if (shard && deal_shifts(shifts, ns, world, owner.data()) != RICADI_OK) throw HipError{"bad shift list"};
  ensure_work(c, m, G);
  Tick tk;
  auto lap = [&](double& acc) {
    if (c->timing) {
      (void)hipStreamSynchronize(st);
      acc += tk.lap();
    }
  };
  // A failure in the OWNER-LOCAL work of a rank (per-shift setup: a singular block; its solves) must not leave the
  // other ranks waiting in the sweep's all-gather: it is recorded here, the rank still takes part in the exchange
  // -- with zero panels and its status word set --, and all ranks throw together once the words have gone round.
  // The words ride in the pressure rows of each rank's first solution panel (the recombination reads velocity
  // rows only), so a sweep costs ONE collective.
  std::string fail;
  auto guarded = [&](auto&& body) {
    if (!shard) {
      body();
      return;
    }
    try {
      body();
    } catch (const HipError& e) {
      fail = e.msg;
    } catch (const std::exception& e) {
      fail = e.what();
    }
  };
  const bool words_fit = (size_t)c->np * m >= 2;
  guarded([&] {
    std::vector<double> mine;
    const int nuse = std::min(ns, prm.adi_max_steps);
    for (int i = 0; i < nuse; ++i)
      if (owner[i] == rank) mine.push_back(shifts[i]);
    prefetch_setup(c, mine.data(), (int)mine.size(), prm.project_w != 0);
  });
  lap(c->t_setup);
  if (prm.project_w) {
    // (replicated: the projection operator is set up by every rank; a rank whose own setup failed skips it)
    if (fail.empty()) guarded([&] { project_panel(c, dW, m); });
  }
  lap(c->t_proj);
  const long it0 = c->total_iters;
  if (!shard) c->sweep_u.ensure(nm * G);
  c->sweep_t.ensure((size_t)nv * m);
  double znorm2 = 0.0;
  int zc_last = c->zc;
  std::vector<double> be(G, 1.0), coef;
  std::vector<ShiftData*> sds(G);
  std::vector<GmresResult> res(G);
  const bool sync_recompress = false, narrow_tail = true;
  AsyncRecompress job(c);
  int steps = 0;
  // relative block norm of the last two visits of every position of the shift cycle
  std::vector<double> rel_h1(ns, 0.0), rel_h2(ns, 0.0);
  std::vector<double> ps_var, rinv_var, cinv_var, cinv_kept, rdummy, hn;
  for (int sw = 0;; ++sw) {
    // Width of this sweep.  With C = R^T R (R upper triangular) column block j of U R^-1 lies in
    // span{U_1..U_j}: it IS the block the step-by-step iteration appends at step j (up to its
    // sign), so the reference's stopping rule -- relative norm of the new block below
    // adi_newZ_reltol (optcont_main.py:123-124) -- is applied block by block below, and the
    // iteration ends after the same step as the sequential one.  So that the solves behind
    // the stopping step are not spent in vain, the block norms of the last two passes over
    // the shift cycle predict that step (per cycle position: same shift, geometric decay)
    // and the sweep is cut there (any run of consecutive, distinct shifts is a valid sweep;
    // its Cauchy data are computed on the spot).
    int g_now = G;
    if (narrow_tail && prm.adi_newZ_reltol > 0.0) {
      for (int g = 0; g < G; ++g) {
        const int pos = (steps + g) % ns;
        if (rel_h1[pos] > 0.0 && rel_h2[pos] > rel_h1[pos]) {
          const double pred = rel_h1[pos] * (rel_h1[pos] / rel_h2[pos]);
          if (pred < prm.adi_newZ_reltol) {
            g_now = g + 1;
            break;
          }
        }
      }
    }
    g_now = std::min(g_now, prm.adi_max_steps - steps);
    if (g_now < 1) break;
    const std::vector<double>* psp;
    const std::vector<double>* rinvp;
    const std::vector<double>* cinvp;
    if (g_now == G && steps % G == 0) {
      psp = &pss[(steps / G) % ncyc];
      rinvp = &rinvs[(steps / G) % ncyc];
      cinvp = &cinvs[(steps / G) % ncyc];
    } else {
      ps_var.resize(g_now);
      for (int g = 0; g < g_now; ++g) ps_var[g] = shifts[(steps + g) % ns];
      rinv_var.assign((size_t)g_now * g_now, 0.0);
      cinv_var.assign(g_now, 0.0);
      if (cauchy_data(ps_var.data(), g_now, rinv_var.data(), cinv_var.data()) != RICADI_OK)
        throw HipError{"Cauchy matrix of a partial ADI sweep is numerically singular"};
      psp = &ps_var;
      rinvp = &rinv_var;
      cinvp = &cinv_var;
    }
    const std::vector<double>& ps = *psp;
    const std::vector<double>& rinv = *rinvp;
    const std::vector<double>& cinv1 = *cinvp;
    const int Gs = g_now;
    // who solves what, and where solution g sits in the buffer the recombination reads
    std::vector<int> slot_of(Gs), mine;
    int per_rank = Gs;
    if (shard) {
      std::vector<int> cnt(world, 0);
      for (int g = 0; g < Gs; ++g) {
        const int r = owner[(steps + g) % ns];
        slot_of[g] = cnt[r]++;                       // index among its owner's items, completed below
        if (r == rank) mine.push_back(g);
      }
      per_rank = *std::max_element(cnt.begin(), cnt.end());
      for (int g = 0; g < Gs; ++g) slot_of[g] += owner[(steps + g) % ns] * per_rank;
    } else {
      for (int g = 0; g < Gs; ++g) {
        slot_of[g] = g;
        mine.push_back(g);
      }
    }
    const int nslot = world * per_rank, nmine = (int)mine.size();
    std::vector<double> psm(nmine);
    for (int k = 0; k < nmine; ++k) psm[k] = ps[mine[k]];
    if (nmine && fail.empty()) guarded([&] { get_shifts(c, psm.data(), be.data(), nmine, sds.data()); });
    lap(c->t_setup);
    double* usolve = shard ? c->xsend : c->sweep_u.p;
    if (shard) {
      if ((size_t)per_rank * nm * sizeof(double) > exchange_panel_capacity(c))
        throw HipError{"exchange buffer too small: " + std::to_string((size_t)per_rank * nm * sizeof(double) + RICADI_XCTL) +
                       " bytes per rank needed, " + std::to_string(c->xcap) + " given to ricadi_set_exchange"};
      // padding slots travel as zeros (their coefficients are zero, but 0 * NaN is not)
      if (nmine < per_rank)
        HIPCHK(hipMemsetAsync(c->xsend + (size_t)nmine * nm, 0, sizeof(double) * nm * (per_rank - nmine), st));
    }
    if (nmine && fail.empty())
      guarded([&] {
        // test hook (tests/test_gpu_round4.py): this rank's share of sweep k fails
        if (const char* inj = shard ? getenv("RICADI_INJECT_SWEEP_FAILURE") : nullptr)
          if (atoi(inj) == sw) throw HipError{"injected failure in sweep " + std::to_string(sw)};
        load_rhs(c, dW, m, c->bvec.p);
        solve_batch(c, sds.data(), nmine, c->bvec.p, 0, usolve, m, true, nullptr, res.data());
      });
    c->lr_ucol = -1;            // only the first solve of a Newton step has U among its rhs columns
    lap(c->t_solve);
    if (fail.empty()) {
      for (int k = 0; k < nmine; ++k)
        if (!res[k].converged) {
          stt.nonconverged++;
          stt.worst_relres = std::max(stt.worst_relres, res[k].max_relres);
        }
      stt.shift_solves += nmine;
    }
    const double* ubase = usolve;
    double words[2] = {fail.empty() ? 0.0 : 1.0, 0.0};
    if (shard) {
      if (!fail.empty()) HIPCHK(hipMemsetAsync(c->xsend, 0, sizeof(double) * nm * per_rank, st));
      if (words_fit) {
        HIPCHK(hipMemcpyAsync(c->xsend + (size_t)nv * m, words, sizeof(words), hipMemcpyHostToDevice, st));
      } else {
        // no pressure rows to carry the words: a control message of their own
        double any = words[0];
        reduce_over_ranks(c, &any, 0, 1);
        if (any != 0.0)
          throw HipError{fail.empty() ? "another rank failed in its share of an ADI sweep" : fail};
      }
      exchange(c, (size_t)per_rank * nm);
      ubase = c->xrecv;
    }
    // coefficient rows (replicated over the m columns), in buffer order: Gs columns of R^-1, then C^-1 1
    coef.assign((size_t)(Gs + 1) * nslot * m, 0.0);
    auto fill_row = [&](int j, const double* col, int stride, int cnt) {   // row j <- col[i * stride], i < cnt
      for (int i = 0; i < cnt; ++i)
        for (int cidx = 0; cidx < m; ++cidx) coef[((size_t)j * nslot + slot_of[i]) * m + cidx] = col[(size_t)i * stride];
    };
    for (int j = 0; j < Gs; ++j) fill_row(j, rinv.data() + j, Gs, Gs);
    c->sweep_coef.ensure(coef.size());
    HIPCHK(hipMemcpyAsync(c->sweep_coef.p, coef.data(), sizeof(double) * (size_t)Gs * nslot * m,
                          hipMemcpyHostToDevice, st));
    // Z <- [Z, U R^-1]: block j = sum_i rinv[i][j] U_i, with its squared norm
    const bool combined = sweep_combine_ok(m, nslot, Gs);
    if (combined) {
      // all blocks and their norms in two launches (K4s)
      c->sweep_part.ensure(sweep_combine_partial_len(nv, m, Gs));
      launch_sweep_combine(st, nv, m, nslot, Gs, ubase, nm, c->sweep_coef.p, c->Z.p, c->zld, c->zc,
                           c->sweep_part.p, c->nrm2.p);
    } else {
      for (int j = 0; j < Gs; ++j) {
        launch_cols_update(st, nv, m, nslot, ubase, nm, c->sweep_coef.p + (size_t)j * nslot * m, 1.0,
                           nullptr, nullptr, c->sweep_t.p);
        launch_copy_cols(st, nv, m, c->sweep_t.p, m, 0, c->Z.p, c->zld, c->zc + j * m, 1.0);
        col_norms2(c, c->sweep_t.p, nv, m, c->nrm2.p + (size_t)j * m);
      }
    }
    hn.resize((size_t)Gs * m);
    HIPCHK(hipMemcpyAsync(hn.data(), c->nrm2.p, sizeof(double) * Gs * m, hipMemcpyDeviceToHost, st));
    std::vector<double> rwords;
    if (shard && words_fit) {
      // the ranks' status words, one strided copy out of the gathered buffer
      rwords.assign((size_t)2 * world, 0.0);
      HIPCHK(hipMemcpy2DAsync(rwords.data(), sizeof(double) * 2, c->xrecv + (size_t)nv * m,
                              sizeof(double) * nm * per_rank, sizeof(double) * 2, world, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    for (int r = 0; r < (int)rwords.size() / 2; ++r)
      if (rwords[(size_t)2 * r] != 0.0)
        throw HipError{r == rank && !fail.empty() ? fail
                                                  : "rank " + std::to_string(r) + " failed in its share of an ADI sweep"};
    // The block norms steer the stopping decisions, which must not differ between the ranks.  The fused
    // recombination sums in a fixed order (sweep_combine_kernel): every rank gets the same bits from the same
    // gathered panels and decides alone.  The per-block fallback hands round rank 0's values.
    if (!combined)
      for (int o = 0; o < Gs * m; o += RICADI_XCTL / 8)
        values_of_rank0(c, hn.data() + o, std::min(RICADI_XCTL / 8, Gs * m - o));
    // the reference's rule, block by block; blocks behind the stopping step are dropped
    int kept = Gs;
    bool stop = false;
    for (int j = 0; j < Gs; ++j) {
      double b2 = 0.0;
      for (int cc = 0; cc < m; ++cc) b2 += hn[(size_t)j * m + cc];
      znorm2 += b2;
      const double relj = znorm2 > 0.0 ? std::sqrt(b2 / znorm2) : 0.0;
      const int pos = (steps + j) % ns;
      rel_h2[pos] = rel_h1[pos];
      rel_h1[pos] = relj;
      stt.rel = relj;
      if (narrow_tail && relj < prm.adi_newZ_reltol) {
        kept = j + 1;
        stop = true;
        break;
      }
    }
    if (!narrow_tail) {
      // sweep granularity (RICADI_FULL_SWEEPS=1): mean block norm of the sweep
      double n2 = 0.0;
      for (int j = 0; j < Gs * m; ++j) n2 += hn[j];
      stt.rel = znorm2 > 0.0 ? std::sqrt(n2 / Gs / znorm2) : 0.0;
      stop = stt.rel < prm.adi_newZ_reltol;
    }
    // W <- W + E (U C^-1 1) over the blocks that are KEPT: every U_g was solved against the same W, so
    // the first `kept` solutions are the sweep of the first `kept` shifts, whose Cauchy data differ only
    // in C^-1 1 (R^-1 of the leading block is the leading block of R^-1) -- W stays the residual factor
    // of the truncated Z, and ||W^T W|| the residual norm that is reported
    const double* cw = cinv1.data();
    if (kept < Gs) {
      cinv_kept.assign(kept, 0.0);
      rdummy.assign((size_t)kept * kept, 0.0);
      if (cauchy_data(ps.data(), kept, rdummy.data(), cinv_kept.data()) != RICADI_OK)
        throw HipError{"Cauchy matrix of a truncated ADI sweep is numerically singular"};
      cw = cinv_kept.data();
    }
    fill_row(Gs, cw, 1, kept);
    HIPCHK(hipMemcpyAsync(c->sweep_coef.p + (size_t)Gs * nslot * m, coef.data() + (size_t)Gs * nslot * m,
                          sizeof(double) * (size_t)nslot * m, hipMemcpyHostToDevice, st));
    launch_cols_update(st, nv, m, nslot, ubase, nm, c->sweep_coef.p + (size_t)Gs * nslot * m, 1.0,
                       nullptr, nullptr, c->sweep_t.p);
    launch_spmm(st, nv, c->E.rp.p, c->E.ci.p, c->E.v.p, c->sweep_t.p, m, nullptr, dW, m, dW, m, 1.0,
                1.0, nullptr, m);
    HIPCHK(hipStreamSynchronize(st));     // `coef` is reused by the next sweep
    c->zc += kept * m;
    steps += kept;
    stt.steps = steps;
    stt.sweeps = sw + 1;
    lap(c->t_recomb);
    static const bool dbg = getenv("RICADI_DEBUG_SWEEPS") != nullptr;
    if (prm.verbose || dbg) {
      int its = 0;
      for (int k = 0; k < nmine; ++k) its = std::max(its, res[k].iters);
      if (dbg) {
        double wf = 0.0;
        DScalar::gram_norms(c, dW, c->nv, m, &wf, nullptr);
        fprintf(stderr, "[ricadi rank %d] sweep %d: Gs %d kept %d per_rank %d nmine %d  ||W^T W|| %.6e  znorm2 %.6e  its", rank, sw + 1,
                Gs, kept, per_rank, nmine, wf, znorm2);
        for (int k = 0; k < nmine; ++k) fprintf(stderr, " %d", res[k].iters);
        fprintf(stderr, "\n");
      }
      fprintf(stderr, "[ricadi] ADI sweep %3d (steps %d..%d): rel new Z %9.3e, gmres its <= %d%s\n",
              sw + 1, steps - kept + 1, steps, stt.rel, its, shard ? " (this rank)" : "");
    }
    if (stop) break;
    if (steps >= prm.adi_max_steps) break;
    if (prm.compress_cols > 0 && c->zc - zc_last >= prm.compress_cols) {
      if (sync_recompress) {
        factor_recompress(c);
      } else {
        // splice in what the helper finished during the last sweeps, hand it the next prefix
        job.finish();
        job.start();
      }
      zc_last = c->zc;
      lap(c->t_compress);
    }
  }
  job.finish();
  lap(c->t_compress);
  stt.gmres_iters = c->total_iters - it0;
  if (shard) {
    // a rank has only seen its own solves
    double v[4] = {(double)stt.gmres_iters, (double)stt.shift_solves, (double)stt.nonconverged, stt.worst_relres};
    reduce_over_ranks(c, v, 3, 1);
    stt.gmres_iters = (long)(v[0] + 0.5);
    stt.shift_solves = (long)(v[1] + 0.5);
    stt.nonconverged = (long)(v[2] + 0.5);
    stt.worst_relres = v[3];
  }
  DScalar::gram_norms(c, dW, c->nv, m, &stt.res_fro, nullptr);
  return true;
}

// Depth of the recycling ring inside the ADI drivers (RICADI_RECYCLE=d; 0 switches it off)
// (cfg2, same-call A/B: depth 0 / 2 / 3 / 5 / 8 -> 63.1 / 56.3 / 55.3 / 53.4 / 52.7 iterations per solve,
// 436.7 / 405.9 / 405.9 / 401.9 / 409.8 ms per step.)  Every stored pair costs n x m doubles per shift:
// 5 where that is small, 3 beyond n = 2e5 (cfg5: 128 shifts x 3 x 64 MB).
static int adi_recycle_depth(const ricadi_ctx* c) {
  const char* e = getenv("RICADI_RECYCLE");     // read per call: tests toggle it
  return e ? std::max(0, std::min(8, atoi(e))) : (c->n <= 200000 ? 5 : 3);
}

static AdiStats lyap_adi_dev(ricadi_ctx* c, const double* shifts, int ns, double* dW, int m,
                             const ricadi_adi_params& prm) {
  AdiStats stt;
  Restore<int> keep_rec(c->rec_depth);
  c->rec_depth = std::max(c->rec_user_depth, adi_recycle_depth(c));
  if (prm.sweep_width > 1 && lyap_adi_sweeps_dev(c, shifts, ns, dW, m, prm, stt)) return stt;
  stt = AdiStats();
  hipStream_t st = c->st;
  ensure_work(c, m);
  // per-shift data of the whole shift cycle (and of the projection) up front: the coarse
  // inverses then come out of one batched factorisation instead of one at a time
  prefetch_setup(c, shifts, std::min(ns, prm.adi_max_steps), prm.project_w != 0);
  if (prm.project_w) project_panel(c, dW, m);
  const long it0 = c->total_iters;
  double znorm2 = 0.0;
  int zc_last = c->zc;
  for (int step = 1; step <= prm.adi_max_steps; ++step) {
    const double p = shifts[(step - 1) % ns];
    ShiftData* sd = get_shift(c, p, 1.0);
    load_rhs(c, dW, m, c->bvec.p);
    GmresResult r = gmres_solve(c, sd, c->bvec.p, c->xs.p, m, true, nullptr);
    if (!r.converged) {
      stt.nonconverged++;
      stt.worst_relres = std::max(stt.worst_relres, r.max_relres);
      if (prm.verbose)
        fprintf(stderr, "[ricadi] ADI step %d shift %g: GMRES stopped at relres %.2e after %d its\n",
                step, p, r.max_relres, r.iters);
    }
    stt.shift_solves++;
    // W <- W - 2 p E V
    launch_spmm(st, c->nv, c->E.rp.p, c->E.ci.p, c->E.v.p, c->xs.p, m, nullptr, dW, m, dW, m,
                -2.0 * p, 1.0, nullptr, m);
    // Z <- [Z, sqrt(-2p) V]
    launch_copy_cols(st, c->nv, m, c->xs.p, m, 0, c->Z.p, c->zld, c->zc, std::sqrt(-2.0 * p));
    double n2 = 0.0;
    col_norms2(c, c->xs.p, c->nv, m, c->nrm2.p);
    HIPCHK(hipMemcpyAsync(c->h_resid, c->nrm2.p, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < m; ++j) n2 += c->h_resid[j];
    n2 *= -2.0 * p;
    znorm2 += n2;
    c->zc += m;
    stt.steps = step;
    stt.rel = znorm2 > 0.0 ? std::sqrt(n2 / znorm2) : 0.0;
    if (prm.verbose)
      fprintf(stderr, "[ricadi] ADI step %3d: shift %10.3e rel new Z %9.3e gmres its %d\n", step,
              p, stt.rel, r.iters);
    if (stt.rel < prm.adi_newZ_reltol) break;
    if (prm.compress_cols > 0 && c->zc - zc_last >= prm.compress_cols) {
      factor_recompress(c);
      zc_last = c->zc;
    }
  }
  stt.gmres_iters = c->total_iters - it0;
  DScalar::gram_norms(c, dW, c->nv, m, &stt.res_fro, nullptr);
  return stt;
}

static void factor_reserve(ricadi_ctx* c, int ld) {
  if ((size_t)c->nv * ld > c->Z.n) c->Z.alloc((size_t)c->nv * ld);
  c->zld = ld;
  c->zc = 0;
}

// ---- compression: Gram matrix on the matrix cores, eigendecomposition, Z * V_k -----
// dZ: NV x cz (ld = ldz).  Returns k and writes Zc (NV x k, ld = k) into dOut
// (which must hold NV*cz doubles).  Singular values (descending) to sv_host.
static Exec main_exec(ricadi_ctx* c) {
  Exec ex;
  ex.st = c->st;
  ex.rb = c->rb;
  ex.pool = &c->pool;
  ex.info = c->info.p;
  return ex;
}

// Gram route of the compression on the given execution resources:  G = Z^T Z on the FP64
// matrix cores, symmetric eigendecomposition (rocSOLVER), Zc = Z V_k.  Returns k; dOut is
// NV x k (ld = k).  Synchronises ex.st before it returns.
static int compress_gram_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz,
                              double thresh, int kmax, bool thresh_relative, double* dOut,
                              std::vector<double>* sv_host) {
  hipStream_t st = ex.st;
  if (cz == 0) return 0;
  TArr<double> G(*ex.pool, (size_t)cz * cz), ev(*ex.pool, cz), work(*ex.pool, cz), sel(*ex.pool);
  HIPCHK(hipMemsetAsync(G.p, 0, sizeof(double) * cz * cz, st));
  launch_gemm_tn(st, c->nv, cz, cz, dZ, ldz, dZ, ldz, G.p, cz);
  RBCHK(rocsolver_dsyevd(ex.rb, rocblas_evect_original, rocblas_fill_upper, cz, G.p, cz, ev.p,
                         work.p, ex.info));
  std::vector<double> lam(cz);
  HIPCHK(hipMemcpyAsync(lam.data(), ev.p, sizeof(double) * cz, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  // eigenvalues ascending; singular values descending
  std::vector<double> sv(cz);
  for (int i = 0; i < cz; ++i) sv[i] = std::sqrt(std::max(lam[cz - 1 - i], 0.0));
  int k = std::min(cz, c->nv);
  if (thresh >= 0.0) {
    const double t = thresh_relative ? thresh * sv[0] : thresh;
    int cnt = 0;
    while (cnt < cz && sv[cnt] > t) ++cnt;
    k = std::min(k, cnt);
  }
  if (kmax > 0) k = std::min(k, kmax);
  if (sv_host) *sv_host = sv;
  if (k == 0) return 0;
  // row-major view of syevd's output: row j = eigenvector j (ascending); the cz x k
  // selection of the k largest is formed on the device
  sel.alloc((size_t)cz * k);
  launch_select_evecs(st, cz, k, G.p, sel.p);
  launch_gemm_nn(st, c->nv, cz, k, dZ, ldz, sel.p, k, dOut, k, 1.0, 0.0);
  HIPCHK(hipStreamSynchronize(st));
  return k;
}

// Recompression without an eigensolver (round 3; the route of the INTERNAL recompressions, which need
// no singular values -- only Zc Zc^T = Z Z^T to rounding):
//   G = Z^T Z (MFMA);  pivoted Cholesky  G ~ R^T R,  R k x cz, stopped at rel^2 of the first pivot
//   (the error of a stopped pivoted Cholesky is the remaining Schur complement, <= its trace);
//   then the rows of R are orthonormalised: with H = R R^T = L L^T the matrix V^T = L^-1 R has orthonormal
//   rows spanning the row space of R, and Zc = Z V, Zc Zc^T = Z (V V^T) Z^T is Z Z^T up to that Schur
//   complement.  chol(H) and the triangular solve are ONE more pivoted Cholesky, of the augmented matrix
//   [H | R] (its pivoting also drops what the first pass kept beyond the tolerance: the final column
//   count equals the eigensolver route's, measured +-1).  Even where L is ill-conditioned the product
//   V V^T is the projector to rounding (the CholQR argument: the error is that of H = L L^T, eps ||H||).
// rocSOLVER's dsyevd on the same Gram matrix was ~4000 launches (12-16 ms) per call; this is ~60.
// Returns k; dOut is NV x k (ld k); synchronises ex.st.  Returns -1 when the matrix is too wide for the
// panel kernel (the caller then takes the eigensolver route).
static int compress_pchol_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz, double rel,
                               double* dOut) {
  hipStream_t st = ex.st;
  if (cz == 0) return 0;
  const int kcap = std::min(cz, c->nv);
  if (pchol_block(cz) == 0 || pchol_block(cz + kcap) == 0) return -1;
  const double tol = rel * rel;
  TArr<double> G(*ex.pool, (size_t)cz * cz), R(*ex.pool, (size_t)kcap * cz), stt(*ex.pool, 8);
  TArr<int> done(*ex.pool, (size_t)cz + kcap);
  PcholState* s1 = reinterpret_cast<PcholState*>(stt.p);
  PcholState* s2 = s1 + 1;
  HIPCHK(hipMemsetAsync(G.p, 0, sizeof(double) * cz * cz, st));
  HIPCHK(hipMemsetAsync(stt.p, 0, sizeof(double) * 8, st));
  HIPCHK(hipMemsetAsync(done.p, 0, sizeof(int) * ((size_t)cz + kcap), st));
  launch_gemm_tn(st, c->nv, cz, cz, dZ, ldz, dZ, ldz, G.p, cz);
  {
    const int nb = pchol_block(cz);
    for (int r0 = 0; r0 < kcap; r0 += nb) {
      launch_pchol_panel(st, G.p, cz, cz, cz, tol, kcap, s1, R.p, cz, done.p);
      launch_pchol_trail(st, G.p, cz, cz, cz, s1, R.p, cz);
    }
  }
  PcholState h1;
  HIPCHK(hipMemcpyAsync(&h1, s1, sizeof(PcholState), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int k1 = h1.rank;
  if (k1 <= 0) return 0;
  // [H | R] with H = R R^T  (gemm_tn wants the tall operand: R^T, cz x k1)
  const int nc2 = k1 + cz;
  TArr<double> Rt(*ex.pool, (size_t)cz * k1), A2(*ex.pool, (size_t)k1 * nc2), R2(*ex.pool, (size_t)k1 * nc2);
  launch_transpose(st, k1, cz, R.p, cz, Rt.p, k1);
  HIPCHK(hipMemsetAsync(A2.p, 0, sizeof(double) * (size_t)k1 * nc2, st));
  launch_gemm_tn(st, cz, k1, k1, Rt.p, k1, Rt.p, k1, A2.p, nc2);
  launch_copy_cols(st, k1, cz, R.p, cz, 0, A2.p, nc2, k1, 1.0);
  {
    const int nb = pchol_block(nc2);
    int* done2 = done.p + cz;
    for (int r0 = 0; r0 < k1; r0 += nb) {
      launch_pchol_panel(st, A2.p, nc2, k1, nc2, tol, k1, s2, R2.p, nc2, done2);
      launch_pchol_trail(st, A2.p, nc2, k1, nc2, s2, R2.p, nc2);
    }
  }
  PcholState h2;
  HIPCHK(hipMemcpyAsync(&h2, s2, sizeof(PcholState), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int k = h2.rank;
  if (k <= 0) return 0;
  // V = (rows 0..k of the carried part)^T: cz x k;  Zc = Z V
  TArr<double> V(*ex.pool, (size_t)cz * k);
  launch_transpose(st, k, cz, R2.p + k1, nc2, V.p, k);
  launch_gemm_nn(st, c->nv, cz, k, dZ, ldz, V.p, k, dOut, k, 1.0, 0.0);
  HIPCHK(hipStreamSynchronize(st));
  return k;
}

// The internal recompressions: pivoted-Cholesky route unless RICADI_RECOMPRESS_EIG=1 (or the factor is too
// wide for it), then the Gram + eigensolver route.
static int recompress_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz, double rel,
                           double* dOut) {
  const bool eig = false;
  if (!eig) {
    const int k = compress_pchol_exec(c, ex, dZ, cz, ldz, rel, dOut);
    if (k >= 0) return k;
  }
  return compress_gram_exec(c, ex, dZ, cz, ldz, rel, 0, true, dOut, nullptr);
}

static int compress_dev(ricadi_ctx* c, const double* dZ, int cz, int ldz, double thresh, int kmax,
                        bool thresh_relative, double* dOut, std::vector<double>* sv_host,
                        bool use_qr) {
  hipStream_t st = c->st;
  if (cz == 0) return 0;
  if (use_qr && cz <= c->nv) {
    // Z = Q R (TSQR panels), R^T = U' S V'^T (rocSOLVER, column-major view of the
    // row-major R), right singular vectors of R = U'; Zc = Z V_k.
    TArr<double> Q(c->pool, (size_t)c->nv * cz), R(c->pool, (size_t)cz * cz), S(c->pool, cz),
        U(c->pool, (size_t)cz * cz), E5(c->pool, cz);
    block_qr_dev(c, dZ, ldz, c->nv, cz, Q.p, R.p);
    RBCHK(rocsolver_dgesvd(c->rb, rocblas_svect_all, rocblas_svect_none, cz, cz, R.p, cz, S.p, U.p, cz,
                           nullptr, 1, E5.p, rocblas_outofplace, c->info.p));
    std::vector<double> sv(cz), Uh((size_t)cz * cz);
    HIPCHK(hipMemcpyAsync(sv.data(), S.p, sizeof(double) * cz, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(Uh.data(), U.p, sizeof(double) * cz * cz, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int k = std::min(cz, c->nv);
    if (thresh >= 0.0) {
      const double t = thresh_relative ? thresh * sv[0] : thresh;
      int cnt = 0;
      while (cnt < cz && sv[cnt] > t) ++cnt;
      k = std::min(k, cnt);
    }
    if (kmax > 0) k = std::min(k, kmax);
    if (sv_host) *sv_host = sv;
    if (k == 0) return 0;
    // row jj of the row-major view of U' = right singular vector jj of R
    std::vector<double> Ch((size_t)cz * k);
    for (int jj = 0; jj < k; ++jj)
      for (int i = 0; i < cz; ++i) Ch[(size_t)i * k + jj] = Uh[(size_t)jj * cz + i];
    TArr<double> sel(c->pool, (size_t)cz * k);
    HIPCHK(hipMemcpyAsync(sel.p, Ch.data(), sizeof(double) * cz * k, hipMemcpyHostToDevice, st));
    launch_gemm_nn(st, c->nv, cz, k, dZ, ldz, sel.p, k, dOut, k, 1.0, 0.0);
    HIPCHK(hipStreamSynchronize(st));
    return k;
  }
  return compress_gram_exec(c, main_exec(c), dZ, cz, ldz, thresh, kmax, thresh_relative, dOut, sv_host);
}

// ---- K5: Householder TSQR tree and block QR ----------------------------------------
// Q (n x w, leading dimension ldq) and R (w x w upper, row-major, written with leading
// dimension ldr) of the n x w panel P (ldp), w <= 32.
static void tsqr_dev(ricadi_ctx* c, const double* P, int ldp, int n, int w, double* Q, int ldq,
                     double* R, int ldr) {
  hipStream_t st = c->st;
  std::vector<int> rows;        // rows of the matrix factorised at each level
  rows.push_back(n);
  while (tsqr_num_blocks(rows.back()) > 1) rows.push_back(tsqr_num_blocks(rows.back()) * 32);
  const int L = (int)rows.size();
  std::vector<TArr<double>> qloc, rst, qfin;
  for (int l = 0; l < L; ++l) {
    qloc.emplace_back(c->pool, (size_t)rows[l] * 32);
    rst.emplace_back(c->pool, (size_t)tsqr_num_blocks(rows[l]) * 32 * 32);
    qfin.emplace_back(c->pool);
  }
  for (int l = 0; l < L; ++l) {
    launch_tsqr_local(st, rows[l], w, l == 0 ? P : rst[l - 1].p, l == 0 ? ldp : 32, qloc[l].p,
                      rst[l].p);
  }
  // R of the top level; Q on the way down
  launch_copy_cols(st, w, w, rst[L - 1].p, 32, 0, R, ldr, 0, 1.0);
  const double* upper = qloc[L - 1].p;       // explicit Q of the top level (one block)
  if (L == 1) {
    launch_copy_cols(st, n, w, qloc[0].p, 32, 0, Q, ldq, 0, 1.0);
  } else {
    for (int l = L - 2; l >= 0; --l) {
      double* dst;
      int ld;
      if (l == 0) {
        dst = Q;
        ld = ldq;
      } else {
        qfin[l].alloc((size_t)rows[l] * 32);
        dst = qfin[l].p;
        ld = 32;
      }
      // intermediate levels keep all 32 columns (ld 32); the final Q only w
      launch_tsqr_apply(st, rows[l], l == 0 ? w : 32, qloc[l].p, upper, dst, ld);
      upper = dst;
    }
    // tsqr_apply writes all 32 columns; columns >= w of Q are exact zeros
  }
  // no synchronisation: the temporaries go back to the context's pool and are reused in
  // stream order
}

// One panel by Cholesky QR, twice (CholQR2): Gram matrices and Q = P T on the MFMA GEMMs,
// the 32 x 32 Cholesky / triangular inverse in cholqr_small_kernel.  Raises c->flag[1] when
// the panel is too ill-conditioned for it (the caller then redoes the factorisation with the
// Householder TSQR tree).
static void panel_cholqr2(ricadi_ctx* c, const double* P, int n, int w, double* Q, int ldq, double* R,
                          int ldr) {
  hipStream_t st = c->st;
  TArr<double> G(c->pool, 1024), T1(c->pool, 1024), R1(c->pool, 1024), T2(c->pool, 1024),
      R2(c->pool, 1024), Q1(c->pool, (size_t)n * 32);
  int* flag = c->flag.p + 1;
  HIPCHK(hipMemsetAsync(G.p, 0, sizeof(double) * 1024, st));
  launch_gemm_tn(st, n, 32, 32, P, 32, P, 32, G.p, 32);
  launch_cholqr_small(st, w, G.p, nullptr, T1.p, R1.p, flag);
  launch_gemm_nn(st, n, 32, 32, P, 32, T1.p, 32, Q1.p, 32, 1.0, 0.0);
  HIPCHK(hipMemsetAsync(G.p, 0, sizeof(double) * 1024, st));
  launch_gemm_tn(st, n, 32, 32, Q1.p, 32, Q1.p, 32, G.p, 32);
  launch_cholqr_small(st, w, G.p, R1.p, T2.p, R2.p, flag);
  launch_gemm_nn(st, n, 32, w, Q1.p, 32, T2.p, 32, Q, ldq, 1.0, 0.0);
  launch_copy_cols(st, w, w, R2.p, 32, 0, R, ldr, 0, 1.0);
}

// One panel of up to 128 columns by CholQR2 (round 3): both Gram matrices and both products Q = P T on the MFMA
// GEMMs, Cholesky factor + triangular inverse of the 128 x 128 Gram matrix in one workgroup
// (cholqr_wide_kernel).  P: n x w (ld ldp); Q1: scratch n x w (ld ldp); Q (ld ldq), R (ld ldr).
static void panel_cholqr2_wide(ricadi_ctx* c, const double* P, int ldp, int n, int w, double* Q1, double* Q, int ldq,
                               double* R, int ldr, double* G, double* T1, double* R1, double* T2, double* R2) {
  hipStream_t st = c->st;
  int* flag = c->flag.p + 1;
  HIPCHK(hipMemsetAsync(G, 0, sizeof(double) * 128 * 128, st));
  launch_gemm_tn(st, n, w, w, P, ldp, P, ldp, G, 128);
  launch_cholqr_wide(st, w, G, 128, T1, R1, flag);
  launch_gemm_nn(st, n, w, w, P, ldp, T1, 128, Q1, ldp, 1.0, 0.0);
  HIPCHK(hipMemsetAsync(G, 0, sizeof(double) * 128 * 128, st));
  launch_gemm_tn(st, n, w, w, Q1, ldp, Q1, ldp, G, 128);
  launch_cholqr_wide(st, w, G, 128, T2, R2, flag);
  launch_gemm_nn(st, n, w, w, Q1, ldp, T2, 128, Q, ldq, 1.0, 0.0);
  launch_gemm_nn(st, w, w, w, R2, 128, R1, 128, R, ldr, 1.0, 0.0);        // R = R_2 R_1
}

// D = Q R for a tall n x kk matrix (ldd): block classical Gram-Schmidt with
// re-orthogonalisation between panels (both passes on the FP64 MFMA GEMMs).  Inside a panel:
// CholQR2 on the matrix cores -- panels of 128 columns (panel_cholqr2_wide; round 2: 32 columns,
// ~20 dependent launches per panel, RICADI_QR_PANEL=32 restores it) when the panel allows it --
// checked once, after the last panel -- else the whole factorisation is redone with 32-column panels
// through the Householder TSQR tree (numerically rank-deficient panels, e.g. raw
// ADI blocks; RICADI_TSQR_HOUSEHOLDER=1 forces it).  Q: n x kk (ld kk), R: kk x kk
// row-major upper triangular.  No panel straddles column `split` (the update norm factorises [Z_new, Z_old]).
static void block_qr_dev(ricadi_ctx* c, const double* D, int ldd, int n, int kk, double* Q,
                         double* R, int split) {
  hipStream_t st = c->st;
  const bool hh_only = false;
  const int pw_env = 128;
  const int PWF = pw_env <= 32 ? 32 : 128;          // panel width of the fast path
  TArr<double> P(c->pool, (size_t)n * PWF), C1(c->pool, (size_t)kk * PWF), C2(c->pool, (size_t)kk * PWF);
  TArr<double> Q1(c->pool), Gw(c->pool), Tw(c->pool);
  if (PWF == 128) {
    Q1.alloc((size_t)n * 128);
    Gw.alloc(128 * 128);
    Tw.alloc(4 * 128 * 128);
  }
  for (int attempt = hh_only ? 1 : 0; attempt < 2; ++attempt) {
    const bool fast = attempt == 0;
    const int PW = fast ? PWF : 32;
    if (fast) HIPCHK(hipMemsetAsync(c->flag.p + 1, 0, sizeof(int), st));
    HIPCHK(hipMemsetAsync(R, 0, sizeof(double) * kk * kk, st));
    for (int c0 = 0, wnext = 0; c0 < kk; c0 += wnext) {
      int w = std::min(PW, kk - c0);
      if (c0 < split && c0 + w > split) w = split - c0;       // no panel straddles `split`
      wnext = w;
      if (PW == 32) HIPCHK(hipMemsetAsync(P.p, 0, sizeof(double) * (size_t)n * 32, st));   // 32-wide kernels read all 32
      launch_copy_cols(st, n, w, D, ldd, c0, P.p, PW, 0, 1.0);
      if (c0 > 0) {
        for (int pass = 0; pass < 2; ++pass) {
          double* C = pass == 0 ? C1.p : C2.p;
          HIPCHK(hipMemsetAsync(C, 0, sizeof(double) * c0 * w, st));
          launch_gemm_tn(st, n, c0, w, Q, kk, P.p, PW, C, w);
          launch_gemm_nn(st, n, c0, w, Q, kk, C, w, P.p, PW, -1.0, 1.0);
        }
        launch_axpby(st, (size_t)c0 * w, 1.0, C2.p, 1.0, C1.p);
        launch_copy_cols(st, c0, w, C1.p, w, 0, R, kk, c0, 1.0);
      }
      if (fast && PW == 128)
        panel_cholqr2_wide(c, P.p, PW, n, w, Q1.p, Q + c0, kk, R + (size_t)c0 * kk + c0, kk, Gw.p, Tw.p,
                           Tw.p + 16384, Tw.p + 2 * 16384, Tw.p + 3 * 16384);
      else if (fast)
        panel_cholqr2(c, P.p, n, w, Q + c0, kk, R + (size_t)c0 * kk + c0, kk);
      else
        tsqr_dev(c, P.p, 32, n, w, Q + c0, kk, R + (size_t)c0 * kk + c0, kk);
    }
    if (!fast) break;
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, c->flag.p + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (!flag) break;
    if (c->opts.verbose) fprintf(stderr, "[ricadi] block QR: ill-conditioned panel, Householder TSQR instead\n");
  }
}

// || Z1 Z1^T - Z0 Z0^T ||_F  via an LQ factorisation of [Z1, Z0]^T (Householder,
// rocSOLVER) -- no squaring, so updates far below 1e-8 relative are resolved.
static double diff_zzt_fnorm(ricadi_ctx* c, const double* dZ1, int k1, const double* dZ0, int k0,
                             double* x1norm) {
  // D = [Z1, Z0] = Q R  (Householder TSQR panels, no squaring of the condition
  // number);  D S D^T = Q (R S R^T) Q^T with S = diag(I_k1, -I_k0), so the norm
  // is that of the small matrix R S R^T -- updates far below 1e-8 are resolved.
  hipStream_t st = c->st;
  const int kk = k1 + k0, nv = c->nv;
  TArr<double> D(c->pool, (size_t)nv * kk), Q(c->pool, (size_t)nv * kk), R(c->pool, (size_t)kk * kk),
      Rt(c->pool, (size_t)kk * kk), Rts(c->pool, (size_t)kk * kk), T(c->pool, (size_t)kk * kk);
  launch_copy_cols(st, nv, k1, dZ1, k1, 0, D.p, kk, 0, 1.0);
  if (k0 > 0) launch_copy_cols(st, nv, k0, dZ0, k0, 0, D.p, kk, k1, 1.0);
  // panels never hold columns of both factors: Z1 ~ Z0 at convergence, and near-duplicate columns inside one
  // panel would send the factorisation to the Householder fallback
  block_qr_dev(c, D.p, kk, nv, kk, Q.p, R.p, k1);
  std::vector<double> Th((size_t)kk * kk);
  auto fro_of = [&](double sneg) {
    // (S R^T)^T (R^T) = R S R^T  with the transposes formed explicitly (kk x kk)
    launch_transpose_sign(st, kk, kk, 1.0, R.p, Rt.p);
    launch_transpose_sign(st, kk, k1, sneg, R.p, Rts.p);
    HIPCHK(hipMemsetAsync(T.p, 0, sizeof(double) * kk * kk, st));
    launch_gemm_tn(st, kk, kk, kk, Rts.p, kk, Rt.p, kk, T.p, kk);
    HIPCHK(hipMemcpyAsync(Th.data(), T.p, sizeof(double) * kk * kk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double f = 0.0;
    for (double v : Th) f += v * v;
    return std::sqrt(f);
  };
  if (x1norm) *x1norm = fro_of(0.0);     // S1 = diag(I, 0): || Z1 Z1^T ||_F
  return fro_of(-1.0);
}

// K = E * (Z * (Z^T B))  (device);  dK is NV x nb
static void gain_dev(ricadi_ctx* c, const DevCsr& Mt, const double* dZ, int cz, int ldz,
                     const double* dB, int nb, double* dK) {
  hipStream_t st = c->st;
  TArr<double> ZtB(c->pool, (size_t)std::max(cz, 1) * nb), T(c->pool, (size_t)c->nv * nb);
  HIPCHK(hipMemsetAsync(ZtB.p, 0, sizeof(double) * std::max(cz, 1) * nb, st));
  launch_gemm_tn(st, c->nv, cz, nb, dZ, ldz, dB, nb, ZtB.p, nb);
  launch_gemm_nn(st, c->nv, cz, nb, dZ, ldz, ZtB.p, nb, T.p, nb, 1.0, 0.0);
  launch_spmm(st, c->nv, Mt.rp.p, Mt.ci.p, Mt.v.p, T.p, nb, nullptr, dK, nb, nullptr, 0, 1.0, 0.0,
              nullptr, nb);
  HIPCHK(hipStreamSynchronize(st));
}

}  // namespace ricadi

// =====================================================================================
//                                      C  A B I
// =====================================================================================
#define API_BEGIN try {
#define API_END                                                   \
  }                                                               \
  catch (const ricadi::HipError& e) {                             \
    ricadi::set_error(e.msg);                                     \
    return RICADI_EHIP;                                           \
  }                                                               \
  catch (const std::exception& e) {                               \
    ricadi::set_error(e.what());                                  \
    return RICADI_EHIP;                                           \
  }                                                               \
  catch (...) {                                                   \
    ricadi::set_error("unknown C++ exception");                   \
    return RICADI_EHIP;                                           \
  }                                                               \
  return RICADI_OK;

#define REQUIRE(cond, code, msg)     \
  do {                               \
    if (!(cond)) {                   \
      ricadi::set_error(msg);        \
      return code;                   \
    }                                \
  } while (0)

extern "C" {

const char* ricadi_last_error(void) { return ricadi::g_err.c_str(); }
int ricadi_version(void) { return 400; }
int ricadi_sizeof_opts(void) { return (int)sizeof(ricadi_opts); }
int ricadi_sizeof_adi_params(void) { return (int)sizeof(ricadi_adi_params); }
// field types in declaration order (d = double, i = int); keep in step with include/ricadi.h
const char* ricadi_struct_signature(void) { return "ricadi_opts:diiiiiiiiii;ricadi_adi_params:ididdiiii"; }

void ricadi_default_opts(ricadi_opts* o) {
  if (!o) return;
  o->gmres_tol = 1e-10;
  o->gmres_restart = 30;
  o->gmres_maxit = 3000;
  o->bj_block = 32;
  o->agg_v = 16;
  o->agg_p = 24;
  o->coarse_max = 4096;
  o->use_coarse = 1;
  o->max_levels = 3;
  o->verbose = 0;
  o->compress_qr = 1;
}

void ricadi_default_adi_params(ricadi_adi_params* p) {
  if (!p) return;
  // /root/reference/optcont_main.py:122-131
  p->adi_max_steps = 200;
  p->adi_newZ_reltol = 1e-8;
  p->nwtn_max_steps = 16;
  p->nwtn_upd_reltol = 5e-8;
  p->nwtn_upd_abstol = 1e-7;
  p->project_w = 1;
  p->verbose = 0;
  p->compress_cols = 0;
  p->sweep_width = 1;
}

int ricadi_create(int device_id, ricadi_ctx** out) {
  REQUIRE(out, RICADI_EINVAL, "ricadi_create: ctx is NULL");
  *out = nullptr;
  API_BEGIN
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) throw ricadi::HipError{"no HIP device visible (this library has no CPU fallback)"};
  if (device_id < 0 || device_id >= ndev) throw ricadi::HipError{"bad device id"};
  HIPCHK(hipSetDevice(device_id));
  std::unique_ptr<ricadi_ctx> c(new ricadi_ctx);
  c->dev = device_id;
  ricadi_default_opts(&c->opts);
  c->precond32 = getenv("RICADI_PRECOND64") == nullptr;
  c->timing = getenv("RICADI_TIMING") != nullptr;
  if (const char* e = getenv("RICADI_SMW")) c->smw = e[0] != '0';
  HIPCHK(hipStreamCreate(&c->st));
  RBCHK(rocblas_create_handle(&c->rb));
  RBCHK(rocblas_set_stream(c->rb, c->st));
  c->flag.alloc(4);
  c->info.alloc(4);
  *out = c.release();
  API_END
}

int ricadi_destroy(ricadi_ctx* ctx) {
  if (!ctx) return RICADI_OK;
  API_BEGIN
  (void)hipSetDevice(ctx->dev);
  (void)hipStreamSynchronize(ctx->st);
  delete ctx;
  API_END
}

int ricadi_set_opts(ricadi_ctx* c, const ricadi_opts* o) {
  REQUIRE(c && o, RICADI_EINVAL, "ricadi_set_opts: NULL argument");
  REQUIRE(o->gmres_restart >= 2 && o->gmres_restart <= 400, RICADI_EINVAL, "gmres_restart out of range");
  REQUIRE(o->gmres_tol > 0 && o->gmres_maxit > 0, RICADI_EINVAL, "bad gmres_tol / gmres_maxit");
  const bool structural = c->has_op && (o->bj_block != c->opts.bj_block || o->agg_v != c->opts.agg_v ||
                                        o->agg_p != c->opts.agg_p || o->coarse_max != c->opts.coarse_max ||
                                        o->max_levels != c->opts.max_levels ||
                                        o->use_coarse != c->opts.use_coarse);
  REQUIRE(!structural, RICADI_ESTATE, "preconditioner options must be set before ricadi_set_operator");
  c->opts = *o;
  return RICADI_OK;
}

void* ricadi_stream(ricadi_ctx* c) { return c ? (void*)c->st : nullptr; }

int ricadi_synchronize(ricadi_ctx* c) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  API_BEGIN
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_set_operator(ricadi_ctx* c, int nv, int np, const int32_t* a_rp, const int32_t* a_ci,
                        const double* a_v, const int32_t* e_rp, const int32_t* e_ci,
                        const double* e_v, const int32_t* j_rp, const int32_t* j_ci,
                        const double* j_v) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  REQUIRE(nv > 0 && np >= 0, RICADI_EINVAL, "bad sizes");
  REQUIRE(a_rp && e_rp, RICADI_EINVAL, "NULL matrix");
  REQUIRE((a_ci && a_v) || a_rp[nv] == 0, RICADI_EINVAL, "NULL matrix arrays");
  REQUIRE((e_ci && e_v) || e_rp[nv] == 0, RICADI_EINVAL, "NULL matrix arrays");
  REQUIRE(np == 0 || (j_rp && j_ci && j_v), RICADI_EINVAL, "NULL J");
  API_BEGIN
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t st = c->st;
  HostCsr A = make_csr(nv, nv, a_rp, a_ci, a_v);
  HostCsr E = make_csr(nv, nv, e_rp, e_ci, e_v);
  HostCsr J;
  if (np > 0) {
    J = make_csr(np, nv, j_rp, j_ci, j_v);
  } else {
    J.nrows = 0;
    J.ncols = nv;
    J.rp.assign(1, 0);
  }
  for (size_t k = 0; k < A.nnz(); ++k)
    if (A.ci[k] < 0 || A.ci[k] >= nv) throw ricadi::HipError{"A: column index out of range"};
  for (size_t k = 0; k < E.nnz(); ++k)
    if (E.ci[k] < 0 || E.ci[k] >= nv) throw ricadi::HipError{"E: column index out of range"};
  for (size_t k = 0; k < J.nnz(); ++k)
    if (J.ci[k] < 0 || J.ci[k] >= nv) throw ricadi::HipError{"J: column index out of range"};
  HostSetup hs;
  if (!c->borrowed)
    c->levels = std::max(2, c->opts.max_levels);
  // smoothed aggregation of the velocity prolongation (two-level setups, folded preconditioner cycle only);
  // RICADI_SA=0 switches it off, RICADI_SA=<omega> sets the damping
  double sa_omega = getenv("RICADI_SA") ? atof(getenv("RICADI_SA")) : 0.5;
  if (c->borrowed || np == 0 || c->opts.bj_block != 32) sa_omega = 0.0;
  build_setup(A, E, J, c->opts, hs, c->levels, sa_omega);
  if (hs.sa) {
    // the folded first sweep takes per-block dense slices of S*P of at most 64 columns
    int kmax = 0;
    std::vector<int> tmp;
    for (int b = 0; b < hs.nbv; ++b) {
      tmp.clear();
      for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q)
        for (int kk = hs.sy_rp[hs.bv_rows[q]]; kk < hs.sy_rp[hs.bv_rows[q] + 1]; ++kk) tmp.push_back(hs.sy_ci[kk]);
      std::sort(tmp.begin(), tmp.end());
      kmax = std::max(kmax, (int)(std::unique(tmp.begin(), tmp.end()) - tmp.begin()));
    }
    if (kmax > 64 || !block_apply2_ok(hs.bs, 64)) {
      if (c->opts.verbose)
        fprintf(stderr, "[ricadi] smoothed aggregation off: a velocity block touches %d coarse columns\n", kmax);
      hs = HostSetup();
      build_setup(A, E, J, c->opts, hs, c->levels, 0.0);
    }
  }
  c->sa = hs.sa;
  c->cache.clear();
  c->child.reset();
  if (hs.multilevel) {
    std::unique_ptr<ricadi_ctx> ch(new ricadi_ctx);
    ch->dev = c->dev;
    ch->st = c->st;
    ch->rb = c->rb;
    ch->borrowed = true;
    ch->opts = c->opts;
    // aggregates of the child level (in units of ITS dofs = this level's aggregates); they double
    // until the last level's dense inverse fits coarse_max
    ch->opts.agg_v = 2;
    ch->opts.agg_p = 1;
    ch->opts.coarse_max = c->opts.coarse_max + c->opts.coarse_max / 8;   // pairs do not always pair up
    ch->levels = 2;
    ch->precond32 = c->precond32;
    ch->smw = c->smw;
    ch->flag.alloc(4);
    ch->info.alloc(4);
    const int rc = ricadi_set_operator(ch.get(), hs.kcv, hs.kcp, hs.l1A.rp.data(), hs.l1A.ci.data(), hs.l1A.v.data(),
                                       hs.l1E.rp.data(), hs.l1E.ci.data(), hs.l1E.v.data(), hs.l1J.rp.data(),
                                       hs.l1J.ci.data(), hs.l1J.v.data());
    if (rc != RICADI_OK) throw ricadi::HipError{std::string("child level: ") + ricadi_last_error()};
    c->child = std::move(ch);
  }
  c->nv = nv;
  c->np = np;
  c->n = nv + np;
  c->bs = hs.bs;
  c->nbv = hs.nbv;
  c->nbp = hs.nbp;
  c->kc = hs.kc;
  c->snnz = hs.s_ci.size();
  c->s_rp.upload(hs.s_rp, st);
  c->s_ci.upload(hs.s_ci, st);
  c->srcA.upload(hs.s_srcA, st);
  c->srcE.upload(hs.s_srcE, st);
  c->srcJ.upload(hs.s_srcJ, st);
  c->A.upload(A, st);
  c->E.upload(E, st);
  c->J.upload(J, st);
  HostCsr JT = transpose(J);
  c->JT.upload(JT, st);
  {
    // rectangular last sweep: pressure dofs touched by every velocity block, dense J^T slices
    c->gt_ok = false;
    if (np > 0 && hs.nbv > 0) {
      std::vector<int> gptr(hs.nbv + 1, 0), gcols;
      int kmax = 0;
      std::vector<int> tmp;
      for (int b = 0; b < hs.nbv; ++b) {
        tmp.clear();
        for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q) {
          const int row = hs.bv_rows[q];
          for (int k = JT.rp[row]; k < JT.rp[row + 1]; ++k) tmp.push_back(JT.ci[k]);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        gcols.insert(gcols.end(), tmp.begin(), tmp.end());
        gptr[b + 1] = (int)gcols.size();
        kmax = std::max(kmax, (int)tmp.size());
      }
      const int ks = kmax <= 32 ? 32 : (kmax <= 64 ? 64 : (kmax <= 128 ? 128 : 0));
      if (ks > 0 && block_apply_rect_ok(hs.bs, ks)) {
        std::vector<double> jtd((size_t)hs.nbv * hs.bs * ks, 0.0);
        for (int b = 0; b < hs.nbv; ++b) {
          const int* cb = gcols.data() + gptr[b];
          const int nc = gptr[b + 1] - gptr[b];
          for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q) {
            const int row = hs.bv_rows[q], il = q - hs.bv_ptr[b];
            for (int k = JT.rp[row]; k < JT.rp[row + 1]; ++k) {
              const int jl = (int)(std::lower_bound(cb, cb + nc, JT.ci[k]) - cb);
              jtd[((size_t)b * hs.bs + il) * ks + jl] += JT.v[k];
            }
          }
        }
        c->gt_ptr.upload(gptr, st);
        c->gt_cols.upload(gcols, st);
        c->gt_jtd.upload(jtd, st);
        c->gt_ks = ks;
        c->gt_ok = true;
        if (c->opts.verbose)
          fprintf(stderr, "[ricadi] last velocity sweep in rectangular form: <= %d pressure dofs per block (slice width %d)\n",
                  kmax, ks);
      }
    }
  }
  c->bv_ptr.upload(hs.bv_ptr, st);
  c->bv_rows.upload(hs.bv_rows, st);
  c->bp_ptr.upload(hs.bp_ptr, st);
  c->bp_rows.upload(hs.bp_rows, st);
  c->bvA.upload(hs.bv_A, st);
  c->bvE.upload(hs.bv_E, st);
  c->jd_ptr.upload(hs.jd_ptr, st);
  c->jd_vblk.upload(hs.jd_vblk, st);
  c->jd_val.upload(hs.jd_val, st);
  c->agg_ptr.upload(hs.agg_ptr, st);
  c->agg_rows.upload(hs.agg_rows, st);
  c->aggof.upload(hs.aggof, st);
  if (hs.sa) {
    c->pt_rp.upload(hs.pt_rp, st);
    c->pt_ci.upload(hs.pt_ci, st);
    c->pt_v.upload(hs.pt_v, st);
  }
  c->synnz = hs.sy_ci.size();
  c->sy_chunk = (c->synnz <= (size_t)10 * std::max(c->n, 1)) ? 8 : 16;
  if (c->opts.verbose)
    fprintf(stderr, "[ricadi] prolongated operator S*Y: %.1f entries per row\n",
            (double)c->synnz / std::max(c->n, 1));
  c->sy_rp.upload(hs.sy_rp, st);
  c->sy_ci.upload(hs.sy_ci, st);
  c->sy_A.upload(hs.sy_A, st);
  c->sy_E.upload(hs.sy_E, st);
  c->sy_J.upload(hs.sy_J, st);
  {
    // dense slices of S*Y per velocity block (first sweep with the coarse residual folded in)
    c->ady_ok = false;
    if (hs.kc > 0 && np > 0 && hs.nbv > 0 && !hs.sy_rp.empty()) {
      std::vector<int> cptr(hs.nbv + 1, 0), ccols, tmp;
      int kmax = 0;
      for (int b = 0; b < hs.nbv; ++b) {
        tmp.clear();
        for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q) {
          const int row = hs.bv_rows[q];
          for (int kk = hs.sy_rp[row]; kk < hs.sy_rp[row + 1]; ++kk) tmp.push_back(hs.sy_ci[kk]);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        ccols.insert(ccols.end(), tmp.begin(), tmp.end());
        cptr[b + 1] = (int)ccols.size();
        kmax = std::max(kmax, (int)tmp.size());
      }
      const int ks = kmax <= 32 ? 32 : (kmax <= 64 ? 64 : 0);
      if (ks > 0 && block_apply2_ok(hs.bs, ks)) {
        const size_t tot = (size_t)hs.nbv * hs.bs * ks;
        std::vector<double> dA(tot, 0.0), dE(tot, 0.0), dJ(tot, 0.0), dT(hs.sa ? tot : 0, 0.0);
        for (int b = 0; b < hs.nbv; ++b) {
          const int* cb = ccols.data() + cptr[b];
          const int nc = cptr[b + 1] - cptr[b];
          for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q) {
            const int row = hs.bv_rows[q], il = q - hs.bv_ptr[b];
            for (int kk = hs.sy_rp[row]; kk < hs.sy_rp[row + 1]; ++kk) {
              const int jl = (int)(std::lower_bound(cb, cb + nc, hs.sy_ci[kk]) - cb);
              const size_t at = ((size_t)b * hs.bs + il) * ks + jl;
              dA[at] += hs.sy_A[kk];
              dE[at] += hs.sy_E[kk];
              dJ[at] += hs.sy_J[kk];
            }
            if (hs.sa)       // (P - Y)[row, :]: its columns are among those of (S P)[row, :] (S has a diagonal)
              for (int kk = hs.pd_rp[row]; kk < hs.pd_rp[row + 1]; ++kk) {
                const int* f = std::lower_bound(cb, cb + nc, hs.pd_ci[kk]);
                if (f == cb + nc || *f != hs.pd_ci[kk]) throw ricadi::HipError{"smoothed prolongation: column outside the block's list"};
                dT[((size_t)b * hs.bs + il) * ks + (int)(f - cb)] += hs.pd_v[kk];
              }
          }
        }
        if (hs.sa) c->cy_dT.upload(dT, st);
        c->cy_ptr.upload(cptr, st);
        c->cy_cols.upload(ccols, st);
        c->cy_dA.upload(dA, st);
        c->cy_dE.upload(dE, st);
        c->cy_dJ.upload(dJ, st);
        c->ady_ks = ks;
        c->ady_ok = true;
      }
    }
  }
  c->syb_ok = hs.kc > 0 && hs.sb_nblk > 0 && hs.syb_max_cols > 0;
  c->syb_max_cols = hs.syb_max_cols;
  if (c->syb_ok) {
    const int nb = hs.sb_nblk, mc = hs.syb_max_cols;
    std::vector<int> rp2((size_t)nb * 33, 0), cols2((size_t)nb * mc, -1);
    for (int b = 0; b < nb; ++b) {
      const int q0 = hs.sb_rowptr[b], nr = hs.sb_rowptr[b + 1] - q0;
      for (int q = 0; q <= 32; ++q) rp2[(size_t)b * 33 + q] = hs.syb_rp[q0 + std::min(q, nr)];
      const int c0 = hs.syb_cptr[b], nc = hs.syb_cptr[b + 1] - c0;
      for (int j = 0; j < nc; ++j) cols2[(size_t)b * mc + j] = hs.syb_cols[c0 + j];
    }
    c->syb_rp2.upload(rp2, st);
    c->syb_cols2.upload(cols2, st);
    c->syb_perm.upload(hs.syb_perm, st);
    c->syb_lidx.upload(hs.syb_lidx, st);
  }
  c->E0.upload(hs.E0, st);
  c->EM.upload(hs.EM, st);
  c->EJ.upload(hs.EJ, st);
  c->ones.upload(std::vector<double>((size_t)c->n, 1.0), st);
  c->sb_nblk = hs.sb_nblk;
  c->sb_max_cols = hs.sb_max_cols;
  c->sb_max_nnz = hs.sb_max_nnz;
  {
    const int nb = hs.sb_nblk, mc = std::max(hs.sb_max_cols, 1);
    std::vector<int> rows2((size_t)nb * 32, -1), rp2((size_t)nb * 33, 0), cols2((size_t)nb * mc, -1),
        colsm2((size_t)nb * mc, -1);
    for (int b = 0; b < nb; ++b) {
      const int q0 = hs.sb_rowptr[b], nr = hs.sb_rowptr[b + 1] - q0;
      for (int q = 0; q <= 32; ++q) rp2[(size_t)b * 33 + q] = hs.sb_rp[q0 + std::min(q, nr)];
      for (int q = 0; q < nr; ++q) rows2[(size_t)b * 32 + q] = hs.sb_rows[q0 + q];
      const int c0 = hs.sb_cptr[b], nc = hs.sb_cptr[b + 1] - c0;
      for (int j = 0; j < nc; ++j) {
        cols2[(size_t)b * mc + j] = hs.sb_cols[c0 + j];
        colsm2[(size_t)b * mc + j] = hs.kc > 0 ? hs.aggof[hs.sb_cols[c0 + j]] : -1;
      }
    }
    c->sb_rows2.upload(rows2, st);
    c->sb_rp2.upload(rp2, st);
    c->sb_cols2.upload(cols2, st);
    c->sb_colsm2.upload(colsm2, st);
  }
  c->sb_perm.upload(hs.sb_perm, st);
  c->sb_lidx.upload(hs.sb_lidx, st);
  c->sb_ok = hs.sb_nblk > 0 && hs.sb_max_cols < 65536;
  if (const char* e = getenv("RICADI_MS_SPMM")) {
    c->ms_spmm = e[0] != '0';
    c->ms_force = e[0] == '2';
  }
  // multi-shift kernel operands: vAJ = A part + J part (disjoint supports) and vE in tile
  // order; velocity-velocity flag in bit 15 of the local index
  auto ms_arrays = [&](const std::vector<int>& rp, const std::vector<int>& ci, const std::vector<double>& a,
                       const std::vector<double>& e, const std::vector<double>& j, const std::vector<int>& perm,
                       const std::vector<uint16_t>& lidx, int ncol_v, DArr<double>& dAJ, DArr<double>& dE,
                       DArr<uint16_t>& dl) {
    const size_t nnz = perm.size();
    std::vector<int> rowof(ci.size());
    for (int i = 0; i + 1 < (int)rp.size(); ++i)
      for (int k = rp[i]; k < rp[i + 1]; ++k) rowof[k] = i;
    std::vector<double> aj(nnz), ee(nnz);
    std::vector<uint16_t> lm(nnz);
    for (size_t kb = 0; kb < nnz; ++kb) {
      const int k = perm[kb];
      aj[kb] = a[k] + j[k];
      ee[kb] = e[k];
      const bool vv = rowof[k] < nv && ci[k] < ncol_v;
      lm[kb] = (uint16_t)(lidx[kb] | (vv ? 0x8000 : 0));
    }
    dAJ.upload(aj, st);
    dE.upload(ee, st);
    dl.upload(lm, st);
  };
  if (c->sb_ok && hs.sb_max_cols <= 160)
    ms_arrays(hs.s_rp, hs.s_ci, hs.s_srcA, hs.s_srcE, hs.s_srcJ, hs.sb_perm, hs.sb_lidx, nv, c->sbAJ, c->sbE,
              c->sb_lidx_ms);
  else
    c->ms_spmm = false;
  if (c->syb_ok && hs.syb_max_cols <= 160 && c->ms_spmm)
    ms_arrays(hs.sy_rp, hs.sy_ci, hs.sy_A, hs.sy_E, hs.sy_J, hs.syb_perm, hs.syb_lidx, hs.kcv, c->sybAJ, c->sybE,
              c->syb_lidx_ms);
  HIPCHK(hipStreamSynchronize(st));
  c->q = 0;
  c->wcols = 0;  // workspaces depend on n
  c->zc = 0;
  c->has_op = true;
  if (c->opts.verbose)
    fprintf(stderr, "[ricadi] operator nv=%d np=%d nnz(S)=%zu | BJ blocks %d+%d (bs=%d) | coarse %d (%d+%d) | "
            "SpMM row blocks %d (max %d distinct cols, %d nnz; mean %.0f cols)\n",
            nv, np, c->snnz, c->nbv, c->nbp, c->bs, c->kc, hs.kcv, hs.kcp, hs.sb_nblk,
            hs.sb_max_cols, hs.sb_max_nnz, hs.sb_nblk ? (double)hs.sb_cols.size() / hs.sb_nblk : 0.0);
  API_END
}

int ricadi_clear_cache(ricadi_ctx* c) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  API_BEGIN
  HIPCHK(hipStreamSynchronize(c->st));
  for (ricadi_ctx* l = c; l; l = l->child.get())
    for (auto& kv : l->cache) {
      kv.second->valid = false;   // buffers stay
      for (auto& r : kv.second->rec) r->serial = -1;
    }
  for (auto& e : c->rec_ring) e->serial = -1;
  API_END
}

int ricadi_set_recycle(ricadi_ctx* c, int depth) {
  REQUIRE(c && depth >= 0 && depth <= 8, RICADI_EINVAL, "recycling depth must be in [0, 8]");
  c->rec_user_depth = c->rec_depth = depth;
  return RICADI_OK;
}

// the exchange state of a context back to "none" (a communicator the library created is destroyed)
static void exchange_reset(ricadi_ctx* c) {
  if (c->xcomm && c->xcomm_owned) (void)ncclCommDestroy(c->xcomm);
  c->xcomm = nullptr;
  c->xcomm_owned = false;
  c->xforce = false;
  c->xsend_own.release();
  c->xrecv_own.release();
  c->xrank = 0;
  c->xworld = 1;
  c->xfn = nullptr;
  c->xuser = nullptr;
  c->xsend = c->xrecv = nullptr;
  c->xcap = 0;
}

int ricadi_set_exchange(ricadi_ctx* c, int rank, int world, ricadi_allgather_fn fn, void* user,
                        void* send_dev, void* recv_dev, int64_t send_capacity) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  API_BEGIN
  exchange_reset(c);
  if (world <= 1 || !fn) return RICADI_OK;
  REQUIRE(rank >= 0 && rank < world && world <= 64, RICADI_EINVAL, "bad rank / world size");
  REQUIRE(send_dev && recv_dev && send_capacity >= 2 * RICADI_XCTL, RICADI_EINVAL, "exchange buffers missing or too small");
  c->xrank = rank;
  c->xworld = world;
  c->xfn = fn;
  c->xuser = user;
  c->xsend = static_cast<double*>(send_dev);
  c->xrecv = static_cast<double*>(recv_dev);
  c->xcap = (size_t)send_capacity;
  API_END
}

int ricadi_rccl_unique_id(void* id_out, int bytes) {
  REQUIRE(id_out && bytes >= (int)sizeof(ncclUniqueId), RICADI_EINVAL, "id buffer of at least 128 bytes required");
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) {
    ricadi::set_error(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    return RICADI_EHIP;
  }
  std::memcpy(id_out, &id, sizeof(id));
  return RICADI_OK;
}

int ricadi_set_exchange_rccl(ricadi_ctx* c, int rank, int world, const void* unique_id, void* comm,
                             int64_t send_capacity) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, RICADI_EINVAL, "bad rank / world size");
  const bool resize = !unique_id && !comm;      // keep the communicator, new buffer sizes
  REQUIRE(!resize || (c->xcomm && c->xrank == rank && c->xworld == world), RICADI_EINVAL,
          "a unique id (ricadi_rccl_unique_id) or a communicator is required");
  REQUIRE(send_capacity >= 2 * RICADI_XCTL, RICADI_EINVAL, "send_capacity too small");
  API_BEGIN
  HIPCHK(hipSetDevice(c->dev));
  if (resize) {
    HIPCHK(hipStreamSynchronize(c->st));
  } else if (comm) {
    exchange_reset(c);
    c->xcomm = static_cast<ncclComm_t>(comm);
  } else {
    exchange_reset(c);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&c->xcomm, world, id, rank);
    if (r != ncclSuccess) {
      c->xcomm = nullptr;
      throw HipError{std::string("ncclCommInitRank: ") + ncclGetErrorString(r)};
    }
    c->xcomm_owned = true;
  }
  const size_t cap = ((size_t)send_capacity + 7) / 8;
  c->xsend_own.alloc(cap);
  c->xrecv_own.alloc(cap * world);
  HIPCHK(hipMemsetAsync(c->xsend_own.p, 0, cap * sizeof(double), c->st));
  HIPCHK(hipMemsetAsync(c->xrecv_own.p, 0, cap * world * sizeof(double), c->st));
  c->xrank = rank;
  c->xworld = world;
  c->xforce = world == 1;
  c->xsend = c->xsend_own.p;
  c->xrecv = c->xrecv_own.p;
  c->xcap = cap * sizeof(double);
  API_END
}

int ricadi_exchange_count(ricadi_ctx* c, int64_t* count_out) {
  REQUIRE(c && count_out, RICADI_EINVAL, "NULL argument");
  *count_out = (int64_t)c->xcount;
  return RICADI_OK;
}

int ricadi_set_dims(ricadi_ctx* c, int nv) {
  REQUIRE(c && nv > 0, RICADI_EINVAL, "bad argument");
  c->cache.clear();
  for (auto& e : c->rec_ring) e->serial = -1;
  c->has_op = false;
  c->nv = nv;
  c->np = 0;
  c->n = nv;
  c->q = 0;
  c->zc = 0;
  c->wcols = 0;
  return RICADI_OK;
}

int ricadi_set_lowrank(ricadi_ctx* c, const double* U, const double* V, int q) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(q >= 0 && q <= 64, RICADI_EINVAL, "low-rank width must be in [0, 64]");
  REQUIRE(q == 0 || (U && V), RICADI_EINVAL, "NULL low-rank factor");
  API_BEGIN
  c->q = q;
  ++c->lr_epoch;
  if (q > 0) {
    const size_t cnt = (size_t)c->nv * q;
    c->U.ensure(cnt);
    c->V.ensure(cnt);
    HIPCHK(hipMemcpyAsync(c->U.p, U, cnt * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCHK(hipMemcpyAsync(c->V.p, V, cnt * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  API_END
}

static int check_panel(ricadi_ctx* c, int m) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  (void)hipSetDevice(c->dev);   // host worker threads start on device 0
  REQUIRE(m >= 1 && m <= RICADI_MAX_M, RICADI_EINVAL, "panel width must be in [1, 128]");
  return RICADI_OK;
}

int ricadi_spmm_dev(ricadi_ctx* c, double alpha, double beta, const double* dX, int m, double* dY) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dX && dY, RICADI_EINVAL, "NULL panel");
  API_BEGIN
  ShiftData* sd = get_shift(c, alpha, beta);
  ensure_work(c, m);
  op_apply(c, sd, dX, dY, m, true);
  API_END
}

int ricadi_spmm(ricadi_ctx* c, double alpha, double beta, const double* X, int m, double* Y) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(X && Y, RICADI_EINVAL, "NULL panel");
  API_BEGIN
  const size_t nm = (size_t)c->n * m;
  ShiftData* sd = get_shift(c, alpha, beta);
  ensure_work(c, m);
  HIPCHK(hipMemcpyAsync(c->pw1.p, X, nm * sizeof(double), hipMemcpyHostToDevice, c->st));
  op_apply(c, sd, c->pw1.p, c->pw2.p, m, true);
  HIPCHK(hipMemcpyAsync(Y, c->pw2.p, nm * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_precond_apply(ricadi_ctx* c, double alpha, double beta, const double* R, int m, double* Z) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(R && Z, RICADI_EINVAL, "NULL panel");
  API_BEGIN
  const size_t nm = (size_t)c->n * m;
  ShiftData* sd = get_shift(c, alpha, beta);
  ensure_work(c, m);
  HIPCHK(hipMemcpyAsync(c->pw1.p, R, nm * sizeof(double), hipMemcpyHostToDevice, c->st));
  precond_apply(c, sd, c->pw1.p, c->pw2.p, m);
  HIPCHK(hipMemcpyAsync(Z, c->pw2.p, nm * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_shift_solve_dev(ricadi_ctx* c, double alpha, double beta, const double* dR, int m,
                           double* dX, int* iters_out, double* relres_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dR && dX, RICADI_EINVAL, "NULL panel");
  int status = RICADI_OK;
  try {
    ShiftData* sd = get_shift(c, alpha, beta);
    ensure_work(c, m);
    load_rhs(c, dR, m, c->bvec.p);
    GmresResult r = gmres_solve(c, sd, c->bvec.p, dX, m, true, relres_out);
    if (iters_out) *iters_out = r.iters;
    if (!r.converged) {
      ricadi::set_error("GMRES did not reach the tolerance");
      status = RICADI_ENOCONV;
    }
  } catch (const ricadi::HipError& e) {
    ricadi::set_error(e.msg);
    return RICADI_EHIP;
  } catch (const std::exception& e) {
    ricadi::set_error(e.what());
    return RICADI_EHIP;
  } catch (...) {
    ricadi::set_error("unknown C++ exception");
    return RICADI_EHIP;
  }
  return status;
}

int ricadi_shift_solve_batch_dev(ricadi_ctx* c, int ng, const double* alphas, const double* betas,
                                 const double* dR, int64_t r_stride, int m, double* dX,
                                 int* iters_out, double* relres_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dR && dX && alphas && betas, RICADI_EINVAL, "NULL argument");
  REQUIRE(ng >= 1 && ng <= RICADI_MAX_GROUPS && (size_t)ng * m <= 2048, RICADI_EINVAL,
          "1 <= ng <= 16 and ng*m <= 2048 required");
  REQUIRE(r_stride == 0 || r_stride >= (int64_t)c->nv * m, RICADI_EINVAL, "bad r_stride");
  int status = RICADI_OK;
  try {
    std::vector<ShiftData*> sds(ng);
    get_shifts(c, alphas, betas, ng, sds.data());
    ensure_work(c, m, ng);
    const size_t nm = (size_t)c->n * m;
    const int nload = r_stride == 0 ? 1 : ng;
    for (int g = 0; g < nload; ++g) load_rhs(c, dR + (size_t)g * r_stride, m, c->bvec.p + (size_t)g * nm);
    std::vector<GmresResult> res(ng);
    solve_batch(c, sds.data(), ng, c->bvec.p, r_stride == 0 ? 0 : nm, dX, m, true, relres_out,
                res.data());
    for (int g = 0; g < ng; ++g) {
      if (iters_out) iters_out[g] = res[g].iters;
      if (!res[g].converged) {
        ricadi::set_error("GMRES did not reach the tolerance");
        status = RICADI_ENOCONV;
      }
    }
  } catch (const ricadi::HipError& e) {
    ricadi::set_error(e.msg);
    return RICADI_EHIP;
  } catch (const std::exception& e) {
    ricadi::set_error(e.what());
    return RICADI_EHIP;
  } catch (...) {
    ricadi::set_error("unknown C++ exception");
    return RICADI_EHIP;
  }
  return status;
}

int ricadi_shift_solve(ricadi_ctx* c, double alpha, double beta, const double* R, const double* Rp,
                       int m, double* X_out, int* iters_out, double* relres_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(R && X_out, RICADI_EINVAL, "NULL panel");
  int status = RICADI_OK;
  try {
    const size_t nm = (size_t)c->n * m, nvm = (size_t)c->nv * m;
    ShiftData* sd = get_shift(c, alpha, beta);
    ensure_work(c, m);
    HIPCHK(hipMemcpyAsync(c->bvec.p, R, nvm * sizeof(double), hipMemcpyHostToDevice, c->st));
    if (c->np > 0) {
      if (Rp)
        HIPCHK(hipMemcpyAsync(c->bvec.p + nvm, Rp, (nm - nvm) * sizeof(double), hipMemcpyHostToDevice, c->st));
      else
        HIPCHK(hipMemsetAsync(c->bvec.p + nvm, 0, (nm - nvm) * sizeof(double), c->st));
    }
    GmresResult r = gmres_solve(c, sd, c->bvec.p, c->xs.p, m, true, relres_out);
    HIPCHK(hipMemcpyAsync(X_out, c->xs.p, nm * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    if (iters_out) *iters_out = r.iters;
    if (!r.converged) {
      ricadi::set_error("GMRES did not reach the tolerance");
      status = RICADI_ENOCONV;
    }
  } catch (const ricadi::HipError& e) {
    ricadi::set_error(e.msg);
    return RICADI_EHIP;
  } catch (const std::exception& e) {
    ricadi::set_error(e.what());
    return RICADI_EHIP;
  } catch (...) {
    ricadi::set_error("unknown C++ exception");
    return RICADI_EHIP;
  }
  return status;
}

int ricadi_apply_e_dev(ricadi_ctx* c, double coef, const double* dV, int m, double* dW) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dV && dW, RICADI_EINVAL, "NULL panel");
  API_BEGIN
  launch_spmm(c->st, c->nv, c->E.rp.p, c->E.ci.p, c->E.v.p, dV, m, nullptr, dW, m, dW, m, coef, 1.0,
              nullptr, m);
  API_END
}

int ricadi_lincomb_dev(ricadi_ctx* c, int nrows, int m, int nvec, const double* dBasis,
                       int64_t stride, const double* coef, double* dOut) {
  REQUIRE(c && dBasis && coef && dOut, RICADI_EINVAL, "NULL argument");
  REQUIRE(nrows > 0 && m >= 1 && m <= RICADI_MAX_M && nvec >= 1 && nvec <= 64, RICADI_EINVAL,
          "bad sizes");
  API_BEGIN
  std::vector<double> h((size_t)nvec * m);
  for (int i = 0; i < nvec; ++i)
    for (int j = 0; j < m; ++j) h[(size_t)i * m + j] = coef[i];
  c->scratch.ensure((size_t)nvec * m + 64);
  HIPCHK(hipMemcpyAsync(c->scratch.p, h.data(), sizeof(double) * nvec * m, hipMemcpyHostToDevice, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  launch_cols_update(c->st, nrows, m, nvec, dBasis, (size_t)stride, c->scratch.p, 1.0, nullptr, nullptr,
                     dOut);
  API_END
}

int ricadi_sweep_recombine_slots_dev(ricadi_ctx* c, int nslot, int G, const double* dU, int m,
                                     const double* coefz, const double* coefw, double* dZ, double* dW,
                                     double* n2_out, double* block_n2_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dU && coefz && coefw && dZ && dW && n2_out, RICADI_EINVAL, "NULL argument");
  REQUIRE(G >= 1 && G <= 64 && nslot >= 1 && nslot <= 128 && G * m <= 2048, RICADI_EINVAL,
          "1 <= G <= 64, 1 <= nslot <= 128 and G*m <= 2048 required");
  API_BEGIN
  hipStream_t st = c->st;
  const int nv = c->nv;
  const size_t nvm = (size_t)nv * m;
  ensure_work(c, m, std::min(G, RICADI_MAX_GROUPS));
  c->sweep_t.ensure(nvm);
  c->sweep_coef.ensure((size_t)(G + 1) * nslot * m);
  c->scratch.ensure((size_t)G * m + 64);
  // coefficient rows replicated over the m columns: G columns of coefz, then coefw
  std::vector<double> coef((size_t)(G + 1) * nslot * m);
  for (int j = 0; j <= G; ++j)
    for (int i = 0; i < nslot; ++i) {
      const double v = j < G ? coefz[(size_t)i * G + j] : coefw[i];
      for (int cc = 0; cc < m; ++cc) coef[((size_t)j * nslot + i) * m + cc] = v;
    }
  HIPCHK(hipMemcpyAsync(c->sweep_coef.p, coef.data(), sizeof(double) * coef.size(),
                        hipMemcpyHostToDevice, st));
  // Z-block j = sum_i coefz[i][j] U_i  (columns j*m .. of dZ, leading dimension G*m)
  for (int j = 0; j < G; ++j) {
    launch_cols_update(st, nv, m, nslot, dU, nvm, c->sweep_coef.p + (size_t)j * nslot * m, 1.0, nullptr,
                       nullptr, c->sweep_t.p);
    launch_copy_cols(st, nv, m, c->sweep_t.p, m, 0, dZ, G * m, j * m, 1.0);
    col_norms2(c, c->sweep_t.p, nv, m, c->scratch.p + (size_t)j * m);
  }
  // W += E (sum_i coefw[i] U_i)
  launch_cols_update(st, nv, m, nslot, dU, nvm, c->sweep_coef.p + (size_t)G * nslot * m, 1.0, nullptr,
                     nullptr, c->sweep_t.p);
  launch_spmm(st, nv, c->E.rp.p, c->E.ci.p, c->E.v.p, c->sweep_t.p, m, nullptr, dW, m, dW, m, 1.0, 1.0,
              nullptr, m);
  std::vector<double> nr((size_t)G * m);
  HIPCHK(hipMemcpyAsync(nr.data(), c->scratch.p, sizeof(double) * G * m, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));   // also keeps `coef` alive until its upload has run
  double n2 = 0.0;
  for (double v : nr) n2 += v;
  *n2_out = n2;
  if (block_n2_out)
    for (int j = 0; j < G; ++j) {
      double b2 = 0.0;
      for (int cc = 0; cc < m; ++cc) b2 += nr[(size_t)j * m + cc];
      block_n2_out[j] = b2;
    }
  API_END
}

int ricadi_sweep_recombine_dev(ricadi_ctx* c, int G, const double* dU, int m, const double* rinv,
                               const double* cinv1, double* dZ, double* dW, double* n2_out) {
  return ricadi_sweep_recombine_slots_dev(c, G, G, dU, m, rinv, cinv1, dZ, dW, n2_out, nullptr);
}

int ricadi_gain_dev(ricadi_ctx* c, double coef, const double* dZ, int cz, int ldz, const double* dB,
                    int nb, double* dK) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(dZ && dB && dK && cz > 0 && ldz >= cz && nb >= 1 && nb <= RICADI_MAX_M, RICADI_EINVAL,
          "bad argument");
  API_BEGIN
  gain_dev(c, c->E, dZ, cz, ldz, dB, nb, dK);
  if (coef != 1.0) launch_axpby(c->st, (size_t)c->nv * nb, coef, dK, 0.0, dK);
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_panel_norms_dev(ricadi_ctx* c, const double* dW, int nrows, int m, double* gram_fro,
                           double* nrm2) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dW && nrows > 0, RICADI_EINVAL, "bad panel");
  API_BEGIN
  DScalar::gram_norms(c, dW, nrows, m, gram_fro, nrm2);
  API_END
}

int ricadi_time_spmm_dev(ricadi_ctx* c, double alpha, double beta, const double* dX, int m,
                         double* dY, int reps, double* ms_per_launch) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dX && dY && reps > 0 && ms_per_launch, RICADI_EINVAL, "bad argument");
  API_BEGIN
  ShiftData* sd = get_shift(c, alpha, beta);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  // plain assembled-CSR saddle SpMM only (no low-rank term): the roofline kernel
  HIPCHK(hipEventRecord(e0, c->st));
  const Batch bt = make_batch(c, sd, m);
  for (int i = 0; i < reps; ++i) saddle_spmm(c, bt, dX, bt.gs, nullptr, dY, bt.gs, nullptr, 0, 1.0, 0.0);
  HIPCHK(hipEventRecord(e1, c->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  API_END
}

int ricadi_time_spmm_batch_dev(ricadi_ctx* c, int ng, const double* alphas, const double* betas,
                               const double* dX, int m, double* dY, int reps, double* ms_per_launch) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dX && dY && alphas && betas && reps > 0 && ms_per_launch, RICADI_EINVAL, "bad argument");
  REQUIRE(ng >= 1 && ng <= RICADI_MAX_GROUPS, RICADI_EINVAL, "1 <= ng <= 16 required");
  API_BEGIN
  std::vector<ShiftData*> sds(ng);
  get_shifts(c, alphas, betas, ng, sds.data());
  const Batch bt = make_batch(c, sds.data(), ng, m);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  // the saddle SpMM exactly as the batched GMRES launches it (no low-rank term; on the FP32-stored vector
  // when the iteration does so)
  DArr<float> x32;
  if (operator_reads_x32(c, m) && ms_pays(c, ng, c->snnz)) {
    x32.alloc(bt.gs * ng);
    for (int g = 0; g < ng; ++g)
      launch_to_f32(c->st, c->n, m, dX + (size_t)g * bt.gs, m, x32.p + (size_t)g * bt.gs, m);
  }
  saddle_spmm(c, bt, dX, bt.gs, nullptr, dY, bt.gs, nullptr, 0, 1.0, 0.0, LowRankArgs(), x32.p);
  HIPCHK(hipEventRecord(e0, c->st));
  for (int i = 0; i < reps; ++i)
    saddle_spmm(c, bt, dX, bt.gs, nullptr, dY, bt.gs, nullptr, 0, 1.0, 0.0, LowRankArgs(), x32.p);
  HIPCHK(hipEventRecord(e1, c->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  API_END
}

// One launch (or launch pair: the dot kernels come with their partial-sum reduction) of a
// hot-path kernel class exactly as the batched GMRES issues it, timed with HIP events on
// the context stream.  Operands are the solver's own workspace buffers, filled with finite
// values; results are discarded.
int ricadi_time_kernel_dev(ricadi_ctx* c, int which, int ng, const double* alphas, const double* betas,
                           int m, int nvec, int reps, double* ms_per_launch) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(alphas && betas && reps > 0 && ms_per_launch, RICADI_EINVAL, "bad argument");
  REQUIRE(ng >= 1 && ng <= RICADI_MAX_GROUPS && (size_t)ng * m <= 2048, RICADI_EINVAL,
          "1 <= ng <= 16 and ng*m <= 2048 required");
  REQUIRE(nvec >= 1 && nvec <= c->opts.gmres_restart, RICADI_EINVAL, "1 <= nvec <= gmres_restart required");
  API_BEGIN
  hipStream_t st = c->st;
  std::vector<ShiftData*> sds(ng);
  get_shifts(c, alphas, betas, ng, sds.data());
  ensure_work(c, m, ng, 0);
  Batch bt = make_batch(c, sds.data(), ng, m);
  const int n = c->n, restart = c->opts.gmres_restart;
  const size_t nm = bt.gs, vs = nm * ng;
  const size_t gsh = (size_t)(restart + 2) * m;
  const size_t gspart = (size_t)dots_num_blocks(n) * (restart + 2) * m;
  // finite fill: byte 0x3C -> 1.5e-18 (FP64), 1.06 (FP16), 0.0115 (FP32)
  HIPCHK(hipMemsetAsync(c->wv.p, 0x3C, sizeof(double) * vs, st));
  HIPCHK(hipMemsetAsync(c->zv.p, 0x3C, sizeof(double) * vs, st));
  if (c->zbasisf.p) HIPCHK(hipMemsetAsync(c->zbasisf.p, 0x3C, sizeof(float) * vs, st));
  HIPCHK(hipMemsetAsync(c->r2.p, 0x3C, sizeof(double) * vs, st));
  HIPCHK(hipMemsetAsync(c->h1.p, 0x3C, sizeof(double) * gsh * ng, st));
  HIPCHK(hipMemsetAsync(c->h2.p, 0x3C, sizeof(double) * gsh * ng, st));
  HIPCHK(hipMemsetAsync(c->scale.p, 0x3C, sizeof(double) * (size_t)ng * m, st));
  HIPCHK(hipMemsetAsync(c->resid.p, 0x3C, sizeof(double) * 2 * c->wcols, st));
  HIPCHK(hipMemsetAsync(c->bnorm2.p, 0x3C, sizeof(double) * (size_t)ng * m, st));
  HIPCHK(hipMemsetAsync(c->g.p, 0x3C, sizeof(double) * (size_t)ng * m * (restart + 1), st));
  HIPCHK(hipMemsetAsync(c->cs.p, 0x3C, sizeof(double) * (size_t)ng * m * restart, st));
  HIPCHK(hipMemsetAsync(c->sn.p, 0x3C, sizeof(double) * (size_t)ng * m * restart, st));
  if (c->kc > 0) {
    HIPCHK(hipMemsetAsync(c->rc.p, 0x3C, sizeof(double) * bt.gsc * ng, st));
    HIPCHK(hipMemsetAsync(c->ec.p, 0x3C, sizeof(double) * bt.gsc * ng, st));
  }
  if (c->np > 0) HIPCHK(hipMemsetAsync(c->tp.p, 0x3C, sizeof(double) * bt.gsp * ng, st));
  const bool b16 = c->basis16, b32 = c->basis32 && !b16;
  const size_t basis_bytes = (size_t)(nvec + 1) * vs * (b16 ? 2 : b32 ? 4 : 8);
  if (c->basis32) {
    HIPCHK(hipMemsetAsync(c->basisf.p, 0x3C, basis_bytes, st));
    HIPCHK(hipMemsetAsync(c->vcur.p, 0x3C, sizeof(double) * vs, st));
  } else {
    HIPCHK(hipMemsetAsync(c->basis.p, 0x3C, basis_bytes, st));
  }
  _Float16* Vh = reinterpret_cast<_Float16*>(c->basisf.p);
  float* Vf = c->basisf.p;
  double* V = c->basis.p;
  const GroupTab& gt = bt.tab;
  const GroupPtrs ones = same_ptr(c->ones.p);
  auto launch = [&]() {
    switch (which) {
      case 0:
        saddle_spmm(c, bt, c->zv.p, nm, nullptr, c->wv.p, nm, nullptr, 0, 1.0, 0.0, LowRankArgs(),
                    operator_reads_x32(c, m) && ms_pays(c, ng, c->snnz) && c->zbasisf.p ? c->zbasisf.p : nullptr);
        break;
      case 1:
        if (c->precond32)
          launch_block_apply_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinvf, c->r2.p, m, nm,
                               c->zv.p, m, nm, m, 0);
        else
          launch_block_apply_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinv, c->r2.p, m, nm,
                               c->zv.p, m, nm, m, 0);
        break;
      case 2:
        if (c->nbp <= 0) throw HipError{"no pressure block"};
        if (c->precond32)
          launch_block_apply_b(st, gt, c->bs, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinvf, c->tp.p, m,
                               bt.gsp, c->zv.p + (size_t)c->nv * m, m, nm, m, 0);
        else
          launch_block_apply_b(st, gt, c->bs, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinv, c->tp.p, m,
                               bt.gsp, c->zv.p + (size_t)c->nv * m, m, nm, m, 0);
        break;
      case 3:
        if (c->kc <= 0) throw HipError{"no coarse level"};
        {
          // the dense inverse lives on the last level
          ricadi_ctx* lc = c;
          Batch lb = bt;
          while (lc->child) {
            Batch t = *lb.sub;
            t.tab = gt;
            lb = t;
            lc = lc->child.get();
          }
          if (c->precond32)
            launch_dense_apply_b(st, gt, lc->kc, m, lb.einvf, (lc->kc + 3) & ~3, lc->rc.p, lc->ec.p);
          else
            launch_dense_apply_b(st, gt, lc->kc, m, lb.einv, lc->rc.p, lc->ec.p);
        }
        break;
      case 4:
        if (!c->syb_ok) throw HipError{"no tiled S*Y"};
        if (ms_pays(c, gt.ng, c->snnz) && spmm_blocked_ms_ok(m, c->syb_max_cols, (size_t)c->kc))
          launch_spmm_blocked_ms(st, gt, bt.alpha, bt.beta, c->sb_nblk, c->sb_rows2.p, c->syb_rp2.p,
                                 c->syb_cols2.p, c->syb_lidx_ms.p, c->sybAJ.p, c->sybE.p, c->ec.p,
                                 m, bt.gsc, c->r2.p, m, nm, c->wv.p, m, nm, -1.0, 1.0, m, c->syb_max_cols);
        else
        launch_spmm_blocked_b(st, gt, c->sb_nblk, c->sb_rows2.p, c->syb_rp2.p, c->syb_cols2.p,
                              c->syb_lidx.p, bt.syvalb, c->ec.p, m, bt.gsc, c->r2.p, m, nm, c->wv.p, m, nm,
                              -1.0, 1.0, m, c->syb_max_cols);
        break;
      case 5:
        if (b16) launch_cols_dots_b(st, gt, n, m, nvec, Vh, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart, c->h1.p, gsh);
        else if (b32) launch_cols_dots_b(st, gt, n, m, nvec, Vf, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart, c->h1.p, gsh);
        else launch_cols_dots_b(st, gt, n, m, nvec, V, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart, c->h1.p, gsh);
        break;
      case 6:
        set_update_dots_nostore(update_dots_keeps_w(m, b16, restart));   // as the iteration launches it
        if (b16) launch_cols_update_dots_b(st, gt, n, m, nvec, Vh, vs, nm, c->h1.p, gsh, c->wv.p, nm, c->partial.p, gspart, c->h2.p, gsh);
        else if (b32) launch_cols_update_dots_b(st, gt, n, m, nvec, Vf, vs, nm, c->h1.p, gsh, c->wv.p, nm, c->partial.p, gspart, c->h2.p, gsh);
        else launch_cols_update_dots_b(st, gt, n, m, nvec, V, vs, nm, c->h1.p, gsh, c->wv.p, nm, c->partial.p, gspart, c->h2.p, gsh);
        set_update_dots_nostore(false);
        break;
      case 7:
        if (b16 && update_hess_fused_ok(m, b16))     // as the iteration launches it: with the Hessenberg update
          launch_cols_update16_hess_b(st, gt, n, nvec, Vh, vs, nm, c->h1.p, c->h2.p, gsh, update_dots_keeps_w(m, b16, restart) ? 1 : 0,
                                      c->wv.p, nm, precond_reads_h16(c, m) ? nullptr : c->vcur.p, nm, Vh + (size_t)nvec * vs, nm,
                                      nvec - 1, restart, c->H.p, c->cs.p, c->sn.p, c->g.p, c->resid.p, c->resid.p + c->wcols,
                                      c->bnorm2.p, c->opts.gmres_tol, nullptr);
        else if (b16) launch_cols_update_b(st, gt, n, m, nvec, Vh, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm, c->scale.p, precond_reads_h16(c, m) ? nullptr : c->vcur.p, nm, Vh + (size_t)nvec * vs, nm);
        else if (b32) launch_cols_update_b(st, gt, n, m, nvec, Vf, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm, c->scale.p, c->vcur.p, nm, Vf + (size_t)nvec * vs, nm);
        else launch_cols_update_b(st, gt, n, m, nvec, V, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm, c->scale.p, V + (size_t)nvec * vs, nm);
        break;
      case 8:
        if (b16 && precond_reads_h16(c, m))
          precond_apply(c, bt, c->wv.p, nm, c->zv.p, c->zbasisf.p, nm, false, Vh);
        else
          precond_apply(c, bt, c->wv.p, nm, c->zv.p, c->zbasisf.p, nm);
        break;
      case 9:
        if (c->kc <= 0) throw HipError{"no coarse level"};
        launch_spmm_b(st, gt, c->kc, c->agg_ptr.p, c->agg_rows.p, ones, c->wv.p, m, nm, nullptr, c->rc.p, m,
                      bt.gsc, nullptr, 0, 0, 1.0, 0.0, m);
        break;
      case 10: case 11: case 12: case 13: case 14: case 15: case 16: {
        // ONE stage of the preconditioner application, through the launcher precond_apply itself uses
        Restore<int> keep(c->pc_stage);
        c->pc_stage = which - 10;
        if (b16 && precond_reads_h16(c, m))
          precond_apply(c, bt, c->wv.p, nm, c->zv.p, c->zbasisf.p, nm, false, Vh);
        else
          precond_apply(c, bt, c->wv.p, nm, c->zv.p, c->zbasisf.p, nm);
        break;
      }
      default:
        throw HipError{"unknown kernel class"};
    }
  };
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  launch();   // warm-up (code object load, caches)
  HIPCHK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  API_END
}

int ricadi_qr(ricadi_ctx* c, const double* Z, int cz, double* Q_out, double* R_out) {
  REQUIRE(c && c->nv > 0, RICADI_ESTATE, "set the operator (or the dimensions) first");
  REQUIRE(Z && R_out && cz > 0 && cz <= c->nv, RICADI_EINVAL, "bad argument");
  API_BEGIN
  (void)hipSetDevice(c->dev);
  const int nv = c->nv;
  DArr<double> dZ, dQ, dR;
  dZ.alloc((size_t)nv * cz);
  dQ.alloc((size_t)nv * cz);
  dR.alloc((size_t)cz * cz);
  HIPCHK(hipMemcpyAsync(dZ.p, Z, sizeof(double) * nv * cz, hipMemcpyHostToDevice, c->st));
  block_qr_dev(c, dZ.p, cz, nv, cz, dQ.p, dR.p);
  HIPCHK(hipMemcpyAsync(R_out, dR.p, sizeof(double) * cz * cz, hipMemcpyDeviceToHost, c->st));
  if (Q_out) HIPCHK(hipMemcpyAsync(Q_out, dQ.p, sizeof(double) * nv * cz, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_setup_info(ricadi_ctx* c, int* out, int nout) {
  REQUIRE(c && out && nout >= 8, RICADI_EINVAL, "bad argument");
  out[0] = c->nv;
  out[1] = c->np;
  out[2] = c->nbv;
  out[3] = c->nbp;
  out[4] = c->bs;
  out[5] = c->kc;
  out[6] = c->sb_nblk;
  out[7] = c->sb_max_cols;
  for (int i = 8; i < nout; ++i) out[i] = 0;
  // [8]: levels in use; [9]: size of the dense inverse on the last level
  int lv = c->kc > 0 ? 2 : 1;
  const ricadi_ctx* lc = c;
  for (; lc->child; lc = lc->child.get()) ++lv;
  if (nout > 8) out[8] = lv;
  if (nout > 9) out[9] = lc->kc;
  // [10]: 1 if the iteration reads the current vector from the FP16 basis (no FP64 copy written), 16-column panels
  if (nout > 10) out[10] = (c->has_op && (getenv("RICADI_BASIS64") == nullptr) && (getenv("RICADI_BASIS32") == nullptr) &&
                            c->n <= (1 << 21) && precond_reads_h16_static(c)) ? 1 : 0;
  // [11], [12]: padded widths of the dense rectangles of the last / first velocity sweep (0: sweep not in that form);
  // [13]: pressure dofs per Schur block list entry count (np), [14]: nnz(J), [15]: nnz of the pressure rows of S*Y
  if (nout > 11) out[11] = c->gt_ok ? c->gt_ks : 0;
  if (nout > 12) out[12] = (c->ady_ok && c->kc > 0) ? c->ady_ks : 0;
  if (nout > 13) out[13] = c->np;
  if (nout > 14) out[14] = (int)c->J.ci.n;
  if (nout > 15) out[15] = c->kc > 0 && c->np > 0 ? (int)(c->synnz) : 0;
  // [16]: entries of the restriction (rows of P^T with smoothed aggregation; else one per dof)
  if (nout > 16) out[16] = c->kc > 0 ? (c->sa ? (int)c->pt_ci.n : c->n) : 0;
  // [17]: route of the last batch of dense coarse inverses on the last level (0 block Gauss-Jordan, 1 rocSOLVER with
  // partial pivoting; -1 none yet); [18]: kernel of the last saddle SpMM launch (0 CSR, 1 LDS-tiled per
  // group, 2 LDS-tiled multi-shift, +4: FP32 x input; -1 none yet)
  if (nout > 17) out[17] = lc->coarse_route;
  if (nout > 18) out[18] = c->k1_variant;
  return RICADI_OK;
}

int ricadi_dense_inverse_batch(ricadi_ctx* c, int k, int nb, double* A, int* route_out) {
  REQUIRE(c && A && k >= 1 && nb >= 1 && nb <= 4 * RICADI_MAX_GROUPS, RICADI_EINVAL, "bad argument");
  API_BEGIN
  hipStream_t st = c->st;
  const size_t kk = (size_t)k * k;
  DArr<double> dA, dA0;
  dA.alloc(kk * nb);
  dA0.alloc(kk * nb);
  HIPCHK(hipMemcpyAsync(dA.p, A, sizeof(double) * kk * nb, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dA0.p, dA.p, sizeof(double) * kk * nb, hipMemcpyDeviceToDevice, st));
  std::vector<double*> hp(nb);
  for (int i = 0; i < nb; ++i) hp[i] = dA.p + kk * i;
  std::vector<int> info(nb, 0);
  const int route = invert_dense_batch(c, hp, k, info, [&] {
    HIPCHK(hipMemcpyAsync(dA.p, dA0.p, sizeof(double) * kk * nb, hipMemcpyDeviceToDevice, st));
  });
  if (route_out) *route_out = route;
  for (int i = 0; i < nb; ++i)
    if (info[i] != 0) throw HipError{"matrix " + std::to_string(i) + " singular (getrf/getri info " + std::to_string(info[i]) + ")"};
  HIPCHK(hipMemcpyAsync(A, dA.p, sizeof(double) * kk * nb, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  API_END
}

int ricadi_time_qr_dev(ricadi_ctx* c, const double* dZ, int cz, int reps, double* ms_per_call) {
  REQUIRE(c && c->nv > 0 && dZ && cz > 0 && cz <= c->nv && reps > 0 && ms_per_call, RICADI_EINVAL,
          "bad argument");
  API_BEGIN
  (void)hipSetDevice(c->dev);
  DArr<double> Q, R;
  Q.alloc((size_t)c->nv * cz);
  R.alloc((size_t)cz * cz);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, c->st));
  for (int i = 0; i < reps; ++i) block_qr_dev(c, dZ, cz, c->nv, cz, Q.p, R.p);
  HIPCHK(hipEventRecord(e1, c->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_call = (double)ms / reps;
  API_END
}

int ricadi_time_gram_dev(ricadi_ctx* c, const double* dZ, int cz, double* dG, int reps,
                         double* ms_per_launch) {
  REQUIRE(c && c->nv > 0 && dZ && dG && cz > 0 && reps > 0 && ms_per_launch, RICADI_EINVAL,
          "bad argument");
  API_BEGIN
  (void)hipSetDevice(c->dev);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipMemsetAsync(dG, 0, sizeof(double) * cz * cz, c->st));
  HIPCHK(hipEventRecord(e0, c->st));
  for (int i = 0; i < reps; ++i) launch_gemm_tn(c->st, c->nv, cz, cz, dZ, cz, dZ, cz, dG, cz);
  HIPCHK(hipEventRecord(e1, c->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  API_END
}

int ricadi_lyap_adi(ricadi_ctx* c, const double* shifts, int ns, const double* W, int m,
                    const ricadi_adi_params* prm, double* Z_out, int* c_out, double* stats_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(shifts && ns > 0 && W && prm, RICADI_EINVAL, "bad argument");
  for (int i = 0; i < ns; ++i) REQUIRE(shifts[i] < 0.0, RICADI_EINVAL, "ADI shifts must be negative");
  REQUIRE(prm->adi_max_steps > 0, RICADI_EINVAL, "adi_max_steps must be positive");
  API_BEGIN
  ensure_work(c, m);
  const long esc0 = c->escalations;
  factor_reserve(c, prm->adi_max_steps * m);
  DArr<double> dW;
  dW.alloc((size_t)c->nv * m);
  HIPCHK(hipMemcpyAsync(dW.p, W, sizeof(double) * c->nv * m, hipMemcpyHostToDevice, c->st));
  if (c->timing) c->t_setup = c->t_solve = c->t_recomb = c->t_compress = c->t_proj = c->t_cyc = c->t_iter = c->t_guess = 0;
  Tick tka;
  AdiStats s = lyap_adi_dev(c, shifts, ns, dW.p, m, *prm);
  if (c->timing) {
    (void)hipStreamSynchronize(c->st);
    fprintf(stderr, "[ricadi timing] lyap_adi: total %.1f ms = setup %.1f + projection %.1f + solves %.1f (Arnoldi iterations %.1f, "
            "restart-cycle bookkeeping %.1f, recycled guesses %.1f) + recombination %.1f + recompression %.1f (+ rest)\n",
            1e3 * tka.lap(), 1e3 * c->t_setup, 1e3 * c->t_proj, 1e3 * c->t_solve, 1e3 * c->t_iter, 1e3 * c->t_cyc,
            1e3 * c->t_guess, 1e3 * c->t_recomb, 1e3 * c->t_compress);
  }
  if (c_out) *c_out = c->zc;
  if (Z_out && c->zc > 0) {
    HIPCHK(hipMemcpy2DAsync(Z_out, sizeof(double) * c->zc, c->Z.p, sizeof(double) * c->zld,
                            sizeof(double) * c->zc, c->nv, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  if (stats_out) {
    stats_out[0] = s.steps;
    stats_out[1] = s.rel;
    stats_out[2] = (double)s.gmres_iters;
    stats_out[3] = (double)s.shift_solves;
    stats_out[4] = s.res_fro;
    stats_out[5] = (double)s.nonconverged;
    stats_out[6] = s.worst_relres;
    stats_out[7] = (double)(c->escalations - esc0);
  }
  API_END
}

}  // extern "C"

// Newton-Kleinman iteration on DEVICE operands (all panels row-major, their own width as leading dimension):
// dB nv x nb, dW nv x mw, dZ0 nv x c0 (or c0 = 0), dOld nv x nb or NULL (`oldB`: whether it is given).  The new
// iterate is left in the context's factor (c->Z, c->zc).
static void ric_newtonadi_run(ricadi_ctx* c, const double* shifts, int ns, const double* dB, int nb, const double* dW,
                              int mw, const double* dZ0, int c0, const double* dOld, const ricadi_adi_params* prm,
                              double* stats_out) {
  const bool oldB = dOld != nullptr;
  hipStream_t st = c->st;
  const int nv = c->nv;
  const int mfull = mw + nb;
  // The Newton loop installs its own low-rank term (K_k - old) B^T in the context; whatever
  // way this function is left -- also by an exception -- no stale term may stay behind for
  // later ricadi_lyap_adi / ricadi_shift_solve calls.
  struct LowRankReset {
    ricadi_ctx* c;
    ~LowRankReset() {
      c->q = 0;
      ++c->lr_epoch;
    }
  } lowrank_reset{c};
  ensure_work(c, mfull);
  TArr<double> dWm(c->pool, (size_t)nv * mw), dK(c->pool, (size_t)nv * nb), dKall(c->pool, (size_t)nv * nb),
      dRhs(c->pool, (size_t)nv * mfull), Zown(c->pool), Znew(c->pool);
  // W is projected in place below: private copy; B, the old gain and Z0 are only read
  HIPCHK(hipMemcpyAsync(dWm.p, dW, sizeof(double) * nv * mw, hipMemcpyDeviceToDevice, st));
  const double* zk = c0 > 0 ? dZ0 : nullptr;       // current (compressed) iterate Z_k, nv x kk (ld kk)
  int kk = c0;
  // the rhs factor W is projected once here; the K_k part is in range(P^T) already
  ricadi_adi_params p2 = *prm;
  Tick tk0;
  prefetch_setup(c, shifts, std::min(ns, prm->adi_max_steps), prm->project_w != 0);
  const double t_pre = c->timing ? ((void)hipStreamSynchronize(st), tk0.lap()) : 0.0;
  if (prm->project_w) project_panel(c, dWm.p, mw);
  if (c->timing) {
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "[ricadi timing] per-shift setup of %d shifts + projection operator %.1f ms, projection solve %.1f ms\n",
            std::min(ns, prm->adi_max_steps), 1e3 * t_pre, 1e3 * tk0.lap());
  }
  p2.project_w = 0;
  if (p2.compress_cols <= 0) {
    // columns the factor may grow by before it is recompressed (RICADI_COMPRESS_COLS overrides): rocSOLVER's
    // tridiagonalisation is launch bound at these sizes (~32 us per column), so fewer, larger eigenproblems are cheaper
    p2.compress_cols = 512;
  }
  double upd = 0, updrel = 0;
  long adi_total = 0, gm_total = 0, sol_total = 0, nonconv = 0, sweep_total = 0;
  const long esc0 = c->escalations;
  double worst = 0.0, last_res = 0.0, last_rhs = 0.0;
  int steps = 0;
  for (steps = 1; steps <= prm->nwtn_max_steps; ++steps) {
    int m = mw;
    Tick tkn;
    if (c->timing) c->t_setup = c->t_solve = c->t_recomb = c->t_compress = c->t_updnorm = c->t_proj = c->t_gain = c->t_cyc = c->t_iter = c->t_guess = 0;
    if (kk > 0) {
      gain_dev(c, c->E, zk, kk, kk, dB, nb, dK.p);
      if (c->timing) c->t_gain += tkn.lap();
      m = mfull;
    } else {
      HIPCHK(hipMemsetAsync(dK.p, 0, sizeof(double) * nv * nb, st));
    }
    // closed loop  cal A - (K_k - old) B^T
    HIPCHK(hipMemcpyAsync(dKall.p, dK.p, sizeof(double) * nv * nb, hipMemcpyDeviceToDevice, st));
    if (oldB) launch_axpby(st, (size_t)nv * nb, -1.0, dOld, 1.0, dKall.p);
    const bool lr = (kk > 0) || oldB;
    c->q = lr ? nb : 0;
    ++c->lr_epoch;
    if (lr) {
      c->U.ensure((size_t)nv * nb);
      c->V.ensure((size_t)nv * nb);
      HIPCHK(hipMemcpyAsync(c->U.p, dKall.p, sizeof(double) * nv * nb, hipMemcpyDeviceToDevice, st));
      HIPCHK(hipMemcpyAsync(c->V.p, dB, sizeof(double) * nv * nb, hipMemcpyDeviceToDevice, st));
    }
    // rhs = [W, K_k]
    launch_copy_cols(st, nv, mw, dWm.p, mw, 0, dRhs.p, m, 0, 1.0);
    if (m > mw) launch_copy_cols(st, nv, nb, dK.p, nb, 0, dRhs.p, m, mw, 1.0);
    factor_reserve(c, prm->adi_max_steps * m);
    DScalar::gram_norms(c, dRhs.p, nv, m, &last_rhs, nullptr);
    // without mtxoldb the low-rank factor U = K_k is the last nb columns of the rhs itself
    c->lr_ucol = (lr && !oldB && m > mw) ? mw : -1;
    AdiStats s = lyap_adi_dev(c, shifts, ns, dRhs.p, m, p2);
    c->lr_ucol = -1;
    last_res = s.res_fro;
    adi_total += s.steps;
    gm_total += s.gmres_iters;
    sol_total += s.shift_solves;
    sweep_total += s.sweeps;
    nonconv += s.nonconverged;
    worst = std::max(worst, s.worst_relres);
    // compressed copy of the new iterate (truncation at the Gram noise floor)
    const int zraw = c->zc;
    Tick tkc;
    factor_recompress(c);
    if (c->timing) c->t_compress += tkc.lap();
    Znew.alloc((size_t)nv * c->zc);
    const int knew = c->zc;
    launch_copy_cols(st, nv, knew, c->Z.p, c->zld, 0, Znew.p, knew, 0, 1.0);
    double x1 = 0.0;
    upd = diff_zzt_fnorm(c, Znew.p, knew, zk, kk, &x1);
    updrel = x1 > 0.0 ? upd / x1 : 0.0;
    {
      double dec[2] = {upd, updrel};       // the stopping decision is rank 0's
      values_of_rank0(c, dec, 2);
      upd = dec[0];
      updrel = dec[1];
    }
    if (c->timing) {
      c->t_updnorm += tkc.lap();
      fprintf(stderr, "[ricadi timing] inside the solves: Arnoldi iterations %.1f ms, restart-cycle bookkeeping %.1f, recycled guesses %.1f\n",
              1e3 * c->t_iter, 1e3 * c->t_cyc, 1e3 * c->t_guess);
      fprintf(stderr, "[ricadi timing] Newton step %d: total %.1f ms = setup %.1f + projection %.1f + solves %.1f + "
              "recombination %.1f + recompression %.1f + update norm %.1f + gain %.1f (+ rest); %d raw columns at the end\n",
              steps, 1e3 * tkn.lap(), 1e3 * c->t_setup, 1e3 * c->t_proj, 1e3 * c->t_solve, 1e3 * c->t_recomb,
              1e3 * c->t_compress, 1e3 * c->t_updnorm, 1e3 * c->t_gain, zraw);
    }
    if (prm->verbose)
      fprintf(stderr, "[ricadi] Newton step %2d: |upd| %9.3e rel %9.3e (%d ADI steps, %d -> %d columns)\n",
              steps, upd, updrel, s.steps, c->zc, knew);
    Zown.swap(Znew);
    zk = Zown.p;
    kk = knew;
    if (upd < prm->nwtn_upd_abstol || updrel < prm->nwtn_upd_reltol) break;
  }
  if (steps > prm->nwtn_max_steps) steps = prm->nwtn_max_steps;
  if (stats_out) {
    stats_out[0] = steps;
    stats_out[1] = upd;
    stats_out[2] = updrel;
    stats_out[3] = (double)adi_total;
    stats_out[4] = (double)gm_total;
    stats_out[5] = (double)sol_total;
    stats_out[6] = (double)nonconv;
    stats_out[7] = worst;
    stats_out[8] = last_res;
    stats_out[9] = last_rhs;
    stats_out[10] = (double)(c->escalations - esc0);
    stats_out[11] = (double)sweep_total;
  }
}

extern "C" {

int ricadi_ric_newtonadi(ricadi_ctx* c, const double* shifts, int ns, const double* B, int nb,
                         const double* W, int mw, const double* Z0, int c0, const double* oldB,
                         const ricadi_adi_params* prm, double* Z_out, int zcap, int* c_out,
                         double* stats_out) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(shifts && ns > 0 && B && W && prm, RICADI_EINVAL, "bad argument");
  REQUIRE(nb >= 1 && nb <= 64 && mw >= 1 && mw + nb <= RICADI_MAX_M, RICADI_EINVAL, "bad widths");
  REQUIRE(c0 == 0 || Z0, RICADI_EINVAL, "Z0 is NULL");
  for (int i = 0; i < ns; ++i) REQUIRE(shifts[i] < 0.0, RICADI_EINVAL, "ADI shifts must be negative");
  API_BEGIN
  hipStream_t st = c->st;
  const int nv = c->nv;
  TArr<double> dB(c->pool, (size_t)nv * nb), dWm(c->pool, (size_t)nv * mw), dOld(c->pool), dZ0(c->pool);
  HIPCHK(hipMemcpyAsync(dB.p, B, sizeof(double) * nv * nb, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dWm.p, W, sizeof(double) * nv * mw, hipMemcpyHostToDevice, st));
  if (oldB) {
    dOld.alloc((size_t)nv * nb);
    HIPCHK(hipMemcpyAsync(dOld.p, oldB, sizeof(double) * nv * nb, hipMemcpyHostToDevice, st));
  }
  if (c0 > 0) {
    dZ0.alloc((size_t)nv * c0);
    HIPCHK(hipMemcpyAsync(dZ0.p, Z0, sizeof(double) * nv * c0, hipMemcpyHostToDevice, st));
  }
  ric_newtonadi_run(c, shifts, ns, dB.p, nb, dWm.p, mw, dZ0.p, c0, oldB ? dOld.p : nullptr, prm, stats_out);
  if (c_out) *c_out = c->zc;
  if (Z_out && c->zc > 0) {
    if (c->zc > zcap) throw ricadi::HipError{"Z_out capacity too small"};
    HIPCHK(hipMemcpy2DAsync(Z_out, sizeof(double) * c->zc, c->Z.p, sizeof(double) * c->zld,
                            sizeof(double) * c->zc, nv, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  API_END
}

// The same with every panel ALREADY ON THE DEVICE (no PCIe traffic inside the call): dB, dW, dZ0, dOldB are device
// pointers (row-major, leading dimension = width; dZ0 / dOldB may be NULL with c0 = 0).  The new iterate stays in
// the context's factor: ricadi_factor_cols, ricadi_factor_get (host) / ricadi_factor_get_dev (device).
int ricadi_ric_newtonadi_dev(ricadi_ctx* c, const double* shifts, int ns, const double* dB, int nb,
                             const double* dW, int mw, const double* dZ0, int c0, const double* dOldB,
                             const ricadi_adi_params* prm, int* c_out, double* stats_out) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(shifts && ns > 0 && dB && dW && prm, RICADI_EINVAL, "bad argument");
  REQUIRE(nb >= 1 && nb <= 64 && mw >= 1 && mw + nb <= RICADI_MAX_M, RICADI_EINVAL, "bad widths");
  REQUIRE(c0 == 0 || dZ0, RICADI_EINVAL, "Z0 is NULL");
  for (int i = 0; i < ns; ++i) REQUIRE(shifts[i] < 0.0, RICADI_EINVAL, "ADI shifts must be negative");
  API_BEGIN
  ric_newtonadi_run(c, shifts, ns, dB, nb, dW, mw, dZ0, c0, dOldB, prm, stats_out);
  if (c_out) *c_out = c->zc;
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

// Copy of the device-resident factor into a DEVICE buffer (nv x cz row-major, ld cz; cz = ricadi_factor_cols)
int ricadi_factor_get_dev(ricadi_ctx* c, double* dZ_out, int cz) {
  REQUIRE(c && dZ_out, RICADI_EINVAL, "NULL argument");
  REQUIRE(cz == c->zc && cz > 0, RICADI_EINVAL, "column count differs from the resident factor");
  API_BEGIN
  launch_copy_cols(c->st, c->nv, cz, c->Z.p, c->zld, 0, dZ_out, cz, 0, 1.0);
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_compress(ricadi_ctx* c, const double* Z, int cz, double thresh, int kmax, double* Zc_out,
                    int* k_out, double* sv_out) {
  REQUIRE(c && c->nv > 0, RICADI_ESTATE, "set the operator (or the dimensions) first");
  REQUIRE(Zc_out && k_out, RICADI_EINVAL, "NULL output");
  API_BEGIN
  const double* dZ;
  int ld;
  DArr<double> tmp, out;
  if (Z) {
    REQUIRE(cz > 0, RICADI_EINVAL, "bad column count");
    tmp.alloc((size_t)c->nv * cz);
    HIPCHK(hipMemcpyAsync(tmp.p, Z, sizeof(double) * c->nv * cz, hipMemcpyHostToDevice, c->st));
    dZ = tmp.p;
    ld = cz;
  } else {
    REQUIRE(c->zc > 0, RICADI_ESTATE, "no device-resident factor");
    dZ = c->Z.p;
    cz = c->zc;
    ld = c->zld;
  }
  out.alloc((size_t)c->nv * cz);
  std::vector<double> sv;
  // the reference's route -- thin QR, then SVD of R ("QR ... SVD", optcont_main.py:133-134) -- up to 1024 columns
  // (the factors the Newton iteration returns are recompressed to a few hundred); raw factors beyond that take the
  // Gram route (singular values resolved to sqrt(eps) sigma_1 instead of eps sigma_1): an O(n c^2) block QR with
  // re-orthogonalisation of thousands of columns costs seconds
  const bool qr_route = c->opts.compress_qr != 0 && cz <= 1024;
  int k = compress_dev(c, dZ, cz, ld, thresh, kmax, false, out.p, &sv, qr_route);
  *k_out = k;
  if (k > 0) {
    HIPCHK(hipMemcpyAsync(Zc_out, out.p, sizeof(double) * c->nv * k, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  if (sv_out) std::memcpy(sv_out, sv.data(), sizeof(double) * std::min<size_t>(sv.size(), (size_t)std::min(cz, c->nv)));
  API_END
}

int ricadi_recompress(ricadi_ctx* c, const double* Z, int cz, double rel, double* Zc_out, int* k_out) {
  REQUIRE(c && c->nv > 0, RICADI_ESTATE, "set the operator (or the dimensions) first");
  REQUIRE(Z && Zc_out && k_out && cz > 0, RICADI_EINVAL, "NULL argument or bad column count");
  API_BEGIN
  DArr<double> tmp, out;
  tmp.alloc((size_t)c->nv * cz);
  out.alloc((size_t)c->nv * cz);
  HIPCHK(hipMemcpyAsync(tmp.p, Z, sizeof(double) * c->nv * cz, hipMemcpyHostToDevice, c->st));
  const int k = recompress_exec(c, main_exec(c), tmp.p, cz, cz, rel > 0.0 ? rel : kInternalRelThresh, out.p);
  *k_out = k;
  if (k > 0) {
    HIPCHK(hipMemcpyAsync(Zc_out, out.p, sizeof(double) * c->nv * k, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  API_END
}

int ricadi_gain(ricadi_ctx* c, const int32_t* mt_rp, const int32_t* mt_ci, const double* mt_v,
                const double* Z, int cz, const double* B, int nb, double* K_out) {
  REQUIRE(c && c->nv > 0, RICADI_ESTATE, "set the operator (or the dimensions) first");
  REQUIRE(mt_rp || c->has_op, RICADI_ESTATE, "no cal E in the context: pass mt_* explicitly");
  REQUIRE(B && K_out && nb >= 1 && nb <= RICADI_MAX_M, RICADI_EINVAL, "bad argument");
  API_BEGIN
  const int nv = c->nv;
  DArr<double> dZ, dB, dK;
  const double* z;
  int ld;
  if (Z) {
    REQUIRE(cz > 0, RICADI_EINVAL, "bad column count");
    dZ.alloc((size_t)nv * cz);
    HIPCHK(hipMemcpyAsync(dZ.p, Z, sizeof(double) * nv * cz, hipMemcpyHostToDevice, c->st));
    z = dZ.p;
    ld = cz;
  } else {
    REQUIRE(c->zc > 0, RICADI_ESTATE, "no device-resident factor");
    z = c->Z.p;
    cz = c->zc;
    ld = c->zld;
  }
  dB.alloc((size_t)nv * nb);
  dK.alloc((size_t)nv * nb);
  HIPCHK(hipMemcpyAsync(dB.p, B, sizeof(double) * nv * nb, hipMemcpyHostToDevice, c->st));
  if (mt_rp) {
    HostCsr Mt = make_csr(nv, nv, mt_rp, mt_ci, mt_v);
    DevCsr dMt;
    dMt.upload(Mt, c->st);
    gain_dev(c, dMt, z, cz, ld, dB.p, nb, dK.p);
  } else {
    gain_dev(c, c->E, z, cz, ld, dB.p, nb, dK.p);
  }
  HIPCHK(hipMemcpyAsync(K_out, dK.p, sizeof(double) * nv * nb, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_lyap_res_norm(ricadi_ctx* c, const double* Z, int cz, const double* W, int m,
                         double* res2_out) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(Z && W && res2_out && cz > 0 && m > 0, RICADI_EINVAL, "bad argument");
  API_BEGIN
  hipStream_t st = c->st;
  const int nv = c->nv, wtot = 2 * cz + m;
  DArr<double> dZ, S, chunk, G;
  dZ.alloc((size_t)nv * cz);
  S.alloc((size_t)nv * wtot);       // [cal A_eff Z, cal E Z, W], ld = wtot
  HIPCHK(hipMemcpyAsync(dZ.p, Z, sizeof(double) * nv * cz, hipMemcpyHostToDevice, st));
  {
    DArr<double> dWh;
    dWh.alloc((size_t)nv * m);
    HIPCHK(hipMemcpyAsync(dWh.p, W, sizeof(double) * nv * m, hipMemcpyHostToDevice, st));
    launch_copy_cols(st, nv, m, dWh.p, m, 0, S.p, wtot, 2 * cz, 1.0);
    HIPCHK(hipStreamSynchronize(st));
  }
  // cal A Z and cal E Z in column chunks of <= 64
  const int CH = 64;
  chunk.alloc((size_t)c->n * CH * 2);
  double* in = chunk.p;
  double* out = chunk.p + (size_t)c->n * CH;
  for (int c0 = 0; c0 < cz; c0 += CH) {
    const int w = std::min(CH, cz - c0);
    launch_copy_cols(st, nv, w, dZ.p, cz, c0, in, w, 0, 1.0);
    launch_spmm(st, nv, c->A.rp.p, c->A.ci.p, c->A.v.p, in, w, nullptr, out, w, nullptr, 0, 1.0, 0.0, nullptr, w);
    if (c->q > 0) {
      c->scratch.ensure((size_t)c->q * w + 64);
      HIPCHK(hipMemsetAsync(c->scratch.p, 0, sizeof(double) * c->q * w, st));
      launch_gemm_tn(st, nv, c->q, w, c->V.p, c->q, in, w, c->scratch.p, w);
      launch_gemm_nn(st, nv, c->q, w, c->U.p, c->q, c->scratch.p, w, out, w, -1.0, 1.0);
    }
    launch_copy_cols(st, nv, w, out, w, 0, S.p, wtot, c0, 1.0);
    launch_spmm(st, nv, c->E.rp.p, c->E.ci.p, c->E.v.p, in, w, nullptr, out, w, nullptr, 0, 1.0, 0.0, nullptr, w);
    launch_copy_cols(st, nv, w, out, w, 0, S.p, wtot, cz + c0, 1.0);
  }
  // project every column: P^T s
  Restore<int> keep_q(c->q);
  for (int c0 = 0; c0 < wtot; c0 += CH) {
    const int w = std::min(CH, wtot - c0);
    launch_copy_cols(st, nv, w, S.p, wtot, c0, in, w, 0, 1.0);
    project_panel(c, in, w);
    launch_copy_cols(st, nv, w, in, w, 0, S.p, wtot, c0, 1.0);
  }
  G.alloc((size_t)wtot * wtot);
  HIPCHK(hipMemsetAsync(G.p, 0, sizeof(double) * wtot * wtot, st));
  launch_gemm_tn(st, nv, wtot, wtot, S.p, wtot, S.p, wtot, G.p, wtot);
  std::vector<double> Gh((size_t)wtot * wtot);
  HIPCHK(hipMemcpyAsync(Gh.data(), G.p, sizeof(double) * wtot * wtot, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  // residual = U S U^T with U = [G, H, Wp], S swaps the first two blocks;
  // ||.||_F^2 = trace(S Gram S Gram)
  auto perm = [&](int i) { return i < cz ? i + cz : (i < 2 * cz ? i - cz : i); };
  double tr = 0.0;
  for (int i = 0; i < wtot; ++i)
    for (int j = 0; j < wtot; ++j)
      tr += Gh[(size_t)perm(i) * wtot + j] * Gh[(size_t)perm(j) * wtot + i];
  *res2_out = tr;
  API_END
}

int ricadi_factor_cols(ricadi_ctx* c, int* c_out) {
  REQUIRE(c && c_out, RICADI_EINVAL, "NULL argument");
  *c_out = c->zc;
  return RICADI_OK;
}

int ricadi_factor_get(ricadi_ctx* c, double* Z_out, int cz) {
  REQUIRE(c && Z_out, RICADI_EINVAL, "NULL argument");
  REQUIRE(cz == c->zc && cz > 0, RICADI_EINVAL, "column count does not match the device factor");
  API_BEGIN
  HIPCHK(hipMemcpy2DAsync(Z_out, sizeof(double) * cz, c->Z.p, sizeof(double) * c->zld,
                          sizeof(double) * cz, c->nv, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_factor_set(ricadi_ctx* c, const double* Z, int cz) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(Z && cz > 0, RICADI_EINVAL, "bad argument");
  API_BEGIN
  factor_reserve(c, cz);
  HIPCHK(hipMemcpyAsync(c->Z.p, Z, sizeof(double) * c->nv * cz, hipMemcpyHostToDevice, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  c->zc = cz;
  API_END
}

}  // extern "C"
