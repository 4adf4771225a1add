// solver_f64.hip -- Solver<double>: every kernel and launch sequence of the double path (one of the two heavy translation units of
// libnmfx.so; __graft_entry__.build compiles them side by side).
#include <hip/hip_runtime.h>

#include "all_impl.hpp"

namespace nmfx {
template class Solver<double>;
SolverBase *make_solver_f64(int64_t p, int64_t n_local, int64_t k, int device) { return new Solver<double>(p, n_local, k, device); }
}  // namespace nmfx
