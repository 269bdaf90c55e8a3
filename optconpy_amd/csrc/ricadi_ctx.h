// ricadi_ctx.h -- the context of libricadi_hip.so: device arrays, per-shift data, workspaces (not installed).
#pragma once
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

extern thread_local std::string g_err;   // ricadi_solver.hip

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
  // BF16 copies (bit patterns) of the four block operands above for the record-driven sweeps of the FP32 cycle
  DArr<uint16_t> bvinvh, bpinvh, gtmh, adymh;
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
  DArr<int> ps_meta;          // fused pressure step: {row, J range, (S Y) range} per (Schur block, row), stride 5
  // velocity sweeps: fixed-stride record per block (layout: ProlongArgs::bmeta); offsets of the two input lists
  DArr<int> sw_meta;
  int sw_stride = 0, sw_in_rect = 0, sw_in_two = 0;
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
  bool w32 = true;            // RICADI_W32=0: the operator's output inside the Arnoldi iteration stays an FP64 panel
  bool x32_always = true;     // RICADI_X32=0: the operator reads the FP32 Z_j only where the multi-shift SpMM runs
  bool blocks16 = true;       // RICADI_BLOCKS16=0: the sweeps apply the FP32 copies of the per-shift blocks
  bool rowwave = true;        // RICADI_ROWWAVE=0: the restriction through the 16-lanes-per-row CSR kernel
  bool mid32 = true;          // RICADI_MID32=0: the velocity part between the sweeps of a cycle stays an FP64 panel
  bool sweep_mfma32 = false;  // RICADI_SWEEP32=1: first velocity sweep on the FP32 matrix cores (experimental, unmeasured: DESIGN 10a)
  bool coarse_mfma32 = false; // RICADI_COARSE32=1: coarse apply on the FP32 matrix cores (experimental, unmeasured: DESIGN 10a)
  int w32_last = -1;          // the last operator launch of an iteration / timing class wrote the FP32 panel (1) or FP64 (0)
  int mid32_last = -1;        // what the last preconditioner application did (1 FP32 panel, 0 FP64; -1 none yet)
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
  DArr<float> wv32;              // w = S z_j of the hot path as an FP32 panel (round 4; the FP64 wv serves the restarts)
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

