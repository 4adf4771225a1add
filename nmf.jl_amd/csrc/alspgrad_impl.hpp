// alspgrad_impl.hpp -- ALSPGrad (src/alspgrad.jl) kernel sequence.
#pragma once
#include "solver.hpp"
namespace nmfx {
template <typename T> void Solver<T>::run_alspgrad(const nmfx_opts &, nmfx_result *, double *) {
    throw StatusError{NMFX_ERR_UNSUPPORTED, "alspgrad: not built yet"};
}
template <typename T> void Solver<T>::subsolve(int, const nmfx_opts &, nmfx_result *) {
    throw StatusError{NMFX_ERR_UNSUPPORTED, "alspgrad sub-solvers: not built yet"};
}
}  // namespace nmfx
