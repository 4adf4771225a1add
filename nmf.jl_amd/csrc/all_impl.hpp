// all_impl.hpp -- every member definition of Solver<T>: included by the two translation units that instantiate the class
// (solver_f32.hip, solver_f64.hip; compiled side by side, nmfx_api.hip holds the C entry points only).
#pragma once
#include "solver_impl.hpp"
#include "projals_impl.hpp"
#include "alspgrad_impl.hpp"
#include "frontend_impl.hpp"
#include "cd_impl.hpp"
#include "rsvd_impl.hpp"
#include "pipeline_impl.hpp"
#include "spa_impl.hpp"
#include "smallk_impl.hpp"
