// solver_f32.hip -- Solver<float>: every kernel and launch sequence of the float path (one of the two heavy translation units of
// libnmfx.so; __graft_entry__.build compiles them side by side).
#include <hip/hip_runtime.h>

#include "all_impl.hpp"

namespace nmfx {
template class Solver<float>;
SolverBase *make_solver_f32(int64_t p, int64_t n_local, int64_t k, int device) { return new Solver<float>(p, n_local, k, device); }
}  // namespace nmfx
