// projals_impl.hpp -- ProjectedALS (src/projals.jl:76-107) kernel sequence.
#pragma once
#include "solver.hpp"
namespace nmfx {
template <typename T> void Solver<T>::enqueue_projals(const nmfx_opts &, long long) {
    throw StatusError{NMFX_ERR_UNSUPPORTED, "projals: not built yet"};
}
}  // namespace nmfx
