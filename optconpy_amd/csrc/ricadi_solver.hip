// ricadi_solver.hip -- context, two-level preconditioner, panel GMRES, low-rank
// ADI, Newton-Kleinman, compression, gain; and the C-ABI of include/ricadi.h.
//
// New code (the reference has no native source).  The
// algorithms restate what the reference *calls* -- see include/ricadi.h for the
// reference call site behind each entry point.
#include "ricadi_ctx.h"

namespace ricadi {
thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace ricadi

// One translation unit, split by concern (the pieces share file-local helpers and the context type):
namespace ricadi {
#include "solver_setup.inl"
#include "solver_precond.inl"
#include "solver_gmres.inl"
#include "solver_adi.inl"
#include "solver_dense.inl"
}  // namespace ricadi
#include "solver_newton.inl"
#include "solver_capi.inl"
