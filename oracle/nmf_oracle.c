/* CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Independent second restatement (the first is oracle/nmf_oracle.py) of the
 * NMF.jl hot path: nmf_skeleton! (src/common.jl:45-111) driving the update_wh!
 * bodies of MultUpdate MSE/Div (src/multupd.jl:83-116,150-193), ProjectedALS
 * (src/projals.jl:76-107 + src/utils.jl:15-84) and ALSPGrad
 * (src/alspgrad.jl:9-19,63-67,86-191,218-222,242-347,400-425), executing the
 * REFERENCE's operation sequence (6 GEMMs/iter for multmse through the p x n
 * product WH, etc.), with plain loops and no BLAS/LAPACK dependency.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  libnmfx.so (the product) does not link or call it.
 *
 * PARITY PINNING: the reference (Julia) cannot run in the build container.
 * Pinned by the reference's own known-answer tests (tests/test_oracle_kat.py)
 * and by agreement with the NumPy twin.  Result.objvalue is PARITY UNPINNED:
 * StatsBase.sqL2dist/gkldiv are un-vendored (compat 0.25-0.34, no Manifest) and
 * no reference test asserts objvalue; restated from their published definition
 * (term in T, running sum in Float64).  ProjectedALS has no dedicated reference
 * test: PARITY UNPINNED beyond the interface smoke run.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int maxiter, update_H, track_objective, maxsubiter, traceiter, _pad;
    double tol, lambda_w, lambda_h, delta, tolg, beta, sigma;   /* all resolved by the caller */
    double l1_w, l2_w, l1_h, l2_h;                              /* CoordinateDescentUpd (coorddesc.jl:62-82) */
} oracle_opts;

typedef struct {
    long niters;
    int converged, _pad;
    double objvalue;
    long inner_iters, backtracks;
    double final_tolg;
} oracle_result;

#define T float
#define SFX f32
#define T_EPS 1.1920928955078125e-07
#define SQRTT sqrtf
#define LOGT logf
#include "nmf_oracle_impl.inc"
#undef T
#undef SFX
#undef T_EPS
#undef SQRTT
#undef LOGT

#define T double
#define SFX f64
#define T_EPS 2.220446049250313e-16
#define SQRTT sqrt
#define LOGT log
#include "nmf_oracle_impl.inc"
