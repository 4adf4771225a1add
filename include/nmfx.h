/*
 * nmfx.h -- C ABI of libnmfx.so, the MI355X (gfx950) replacement for the NMF.jl
 * iterative hot path.  Plain pointers and sizes only; no torch / HIP types.
 *
 * This is what a Julia `ccall` (or any FFI) binds to keep NMF.jl's
 *     NMF.solve!(alg, X, W, H) :: NMF.Result{T}
 * call surface while the inner loops run as hand-written HIP on the GPU.
 * The reference interface each entry point replaces is cited as
 * /root/reference/<file>:<line>.  INTEGRATION.md shows the Julia-side binding.
 *
 * Conventions (same as Julia Matrix{T}): column-major, contiguous,
 *   X is p x n (ld = p), W is p x k (ld = p), H is k x n (ld = k),
 *   T is float (NMFX_F32) or double (NMFX_F64).
 * Every function returns an nmfx_status; nmfx_last_error(ctx) gives the text.
 * A context may be used by one host thread at a time; the library never calls
 * back into the host language and keeps no global mutable state.
 *
 * Two path switches are read from the environment when a context is created (development / A-B aids; results agree to
 * rounding either way, tests/test_gpu_multupd.py, tests/test_gpu_projals_alspgrad.py):
 *   NMFX_SMALLK=0       MultUpdate-MSE with k <= 64 in Float32 stays on the general split-K path (default: the 4-launch
 *                       stripe kernels when p*n <= 4096^2)
 *   NMFX_CHOL_SLOTS=0   ProjectedALS factors between its big products instead of under them (default 8: block slots
 *                       the products leave to the factorisation stream)
 */
#ifndef NMFX_H
#define NMFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nmfx_ctx nmfx_ctx;

/* element type T of X, W, H (the reference is generic over Float32/Float64) */
enum { NMFX_F32 = 0, NMFX_F64 = 1 };

/* algorithm selector = the `alg` structs accepted by NMF.solve!:
 *   MultUpdate{T}(obj=:mse)  src/multupd.jl:9-52   (nnmf alg=:multmse, src/interf.jl:65-66)
 *   MultUpdate{T}(obj=:div)  src/multupd.jl:9-52   (alg=:multdiv,  src/interf.jl:67-68)
 *   ProjectedALS{T}          src/projals.jl:18-39  (alg=:projals,  src/interf.jl:61-62)
 *   ALSPGrad{T}              src/alspgrad.jl:352-383 (alg=:alspgrad, src/interf.jl:63-64)
 *   CoordinateDescent{T}     src/coorddesc.jl:23-51  (alg=:cd,       src/interf.jl:69-70)   -- SURVEY.md section 8f rank 2
 *   GreedyCD{T}              src/greedycd.jl:10-35   (alg=:greedycd, src/interf.jl:71-72; nnmf's default) */
enum { NMFX_ALG_MULTMSE = 0, NMFX_ALG_MULTDIV = 1, NMFX_ALG_PROJALS = 2, NMFX_ALG_ALSPGRAD = 3, NMFX_ALG_CD = 4, NMFX_ALG_GREEDYCD = 5 };

/* status codes; the host shim maps them to the reference's exceptions:
 *   BAD_ARG       -> ArgumentError       (src/multupd.jl:27-31, src/interf.jl:15-33)
 *   DIM_MISMATCH  -> DimensionMismatch   (src/common.jl:12, :30)
 *   NOT_POSDEF    -> PosDefException     (potrf! in src/utils.jl:68,78)
 *   ALPHA_NONFINITE -> ErrorException("alpha is not finite") (src/alspgrad.jl:140,296) */
typedef enum {
    NMFX_OK = 0,
    NMFX_ERR_BAD_ARG = 1,
    NMFX_ERR_DIM_MISMATCH = 2,
    NMFX_ERR_NOT_POSDEF = 3,
    NMFX_ERR_ALPHA_NONFINITE = 4,
    NMFX_ERR_HIP = 5,
    NMFX_ERR_RCCL = 6,
    NMFX_ERR_NO_DEVICE = 7,
    NMFX_ERR_STATE = 8,
    NMFX_ERR_UNSUPPORTED = 9
} nmfx_status;

/* Union of the reference option structs; every field is ALREADY RESOLVED by the
 * caller (the Julia structs hold concrete values after construction):
 *   MultUpdate   fields obj,maxiter,verbose,tol,update_H,lambda_w,lambda_h  src/multupd.jl:9-17
 *   ProjectedALS fields maxiter,verbose,tol,update_H,lambda_w,lambda_h      src/projals.jl:18-24
 *   ALSPGrad     fields maxiter,maxsubiter,tol,tolg,update_H,verbose        src/alspgrad.jl:352-358
 * delta = sqrt(eps(T)) is what solve! passes to the MultUpd updaters (src/multupd.jl:48,50);
 * traceiter/beta/sigma are the literals 20, 0.2, 0.01 of src/alspgrad.jl:407,417. */
typedef struct {
    int32_t maxiter;          /* outer iteration cap (nmf_skeleton!, src/common.jl:64) */
    int32_t update_H;         /* 0: H is left bit-identical (test/interf.jl:33-37) */
    int32_t track_objective;  /* 1: evaluate the objective at t=0 and after every iteration, like
                                 verbose=true does (src/common.jl:56,79); fills objv_trace */
    int32_t maxsubiter;       /* ALSPGrad.maxsubiter (200) */
    int32_t traceiter;        /* back-tracking cap (20) */
    int32_t check_every;      /* host polls the device-side stop flag every this many outer iterations; <= 0: adaptive
                                 (a window of 4 that doubles, up to 256, while a window takes < 1 ms of wall time).
                                 Results do not depend on it: iterations past the stop are no-ops */
    double tol;               /* stop_condition eps (src/common.jl:92-111) */
    double lambda_w, lambda_h;
    double delta;
    double tolg;              /* initial ALSPGradUpd.tolg (decays *0.1, src/alspgrad.jl:409-421) */
    double beta, sigma;
    /* CoordinateDescentUpd's resolved regularisation (src/coorddesc.jl:62-82): l1 = alpha*l1ratio, l2 = alpha*(1-l1ratio)
     * for W (regularization in {:both, :transformation}) and H ({:both, :components}); GreedyCD uses lambda_w / lambda_h
     * as its L1 coefficients (src/greedycd.jl:15-16). */
    double l1_w, l2_w, l1_h, l2_h;
    /* arithmetic of the two p*n*k products.  NMFX_PREC_FP32 (0, default): v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64,
     * the element type's own arithmetic.  NMFX_PREC_BF16X3 (1; f32 contexts with k >= 65, ignored otherwise): operands split
     * into bf16 pairs, three bf16 MFMA products per term, fp32 accumulation (csrc/gemm_bf16x3.hpp) -- ~2.2x faster launches,
     * GEMM error vs fp64 as small as the fp32 path's; opt-in because the inputs are rounded to 16 mantissa bits. */
    int32_t precision;
    /* CoordinateDescent(shuffle = ...), src/coorddesc.jl:130-134.  0: components are swept in order 1..k (shuffle = false).
     * != 0: shuffle = true -- every call of _update_coord_descent! (W side and H side of every outer iteration) sweeps in a
     * fresh random order.  The reference draws randperm(k) from Julia's global RNG, whose stream cannot be reproduced outside
     * Julia; here the order of call c = 2*(t-1) + side is the permutation that sorts the k keys
     * Philox4x32-10(counter = (i, c, 4, 0), key = cd_shuffle)[0], i = 0..k-1 (ties by index) -- documented, reproducible,
     * identical on every rank of a sharded run. */
    int32_t cd_shuffle;
    /* ALSPGrad: how the gradient of an inner iteration is formed.  The reference recomputes G = Gram*Z - B by a full product at the
     * top of EVERY inner iteration (src/alspgrad.jl:124-127, 280-283).  1 = exactly that ("exact").  n > 1: the running form
     * G += Gram*D(accepted step) -- Gram*D is what the accepted trial step just computed (alspgrad.jl:150-152) -- with a full product
     * every n-th inner iteration to bound the rounding drift of the running sum (one product less per inner iteration).
     * 0 = the library default: 64 for Float64, 16 for Float32.  Measured against the CPU oracle on six seeded problems per element
     * type (scripts/alspgrad_gradient_modes.py, profiles/r05_alspgrad_gradient_modes.jsonl): the inner-iteration and back-tracking
     * counters are the oracle's in every mode, and the exact form is NOT closer to the oracle's objective than the running form
     * (Float32: max relative deviation 2.1e-4 exact, 1.4e-4 period 16, 2.1e-4 period 64; Float64: <= 8.4e-13 all) -- the Float32
     * deviation is the products' summation order, not the gradient's form -- while it costs 11 % (Float32, 8192^2, k = 256) to 25 %
     * (the Float64 C5 shard) more time per outer iteration. */
    int32_t pg_refresh;
    /* ProjectedALS: how H = (W'W + lambda I) \ W'X is solved after potrf! (src/utils.jl:63-70).  NMFX_HSOLVE_AUTO (0): the library
     * default -- since round 5 the reference's potrs! route wherever the strip kernel exists (padded k <= 512 in Float32, 256 in Float64: csrc/chol.hpp,
     * potrs_strip_kernel: 29-36 us against 55 for the product form at 16384 columns, k = 256, Float32), the product form beyond;
     * NMFX_HSOLVE_PRODUCT (1): Uinv (Uinv' B), two products with the inverted factor; NMFX_HSOLVE_POTRS (2): forward and back
     * substitution with the factor itself (beyond the strip kernel: potrs_panel_kernel while its panel fits the LDS; where neither
     * kernel exists -- padded k > 512 in Float32 with a panel beyond the LDS -- the call fails with NMFX_ERR_UNSUPPORTED instead of
     * running another route silently). */
    int32_t h_solve;
    /* stop_condition's four sums per component (src/common.jl:95-104).  0 (default): the T-rounded terms summed in Float64 in a fixed
     * tree order, inside the update launches (free; differs from the reference's totals by <= ~eps(T) sqrt(length) relative, so the
     * decision can differ only when an iteration's relative change sits that close to tol).  1: the reference's own arithmetic -- every
     * sum accumulated SEQUENTIALLY in T, one chain per component, in index order -- by a pass of its own after each iteration
     * (measured +0.08-0.09 ms per iteration at 16384 x 16384, k = 256, Float32 since round 6 -- 0.42 before: one dependent add per element and
     * chain, a wave per kind of sum that does nothing else, k / 4 workgroups per factor in one launch); `niters`,
     * `converged` and the relchange column are then the
     * reference's bit for bit whenever W and H are.  One GPU only (the chains would have to run through the ranks in turn). */
    int32_t stop_sums;
    int32_t reserved0;        /* must be 0 */
} nmfx_opts;
enum { NMFX_PREC_FP32 = 0, NMFX_PREC_BF16X3 = 1 };
enum { NMFX_HSOLVE_AUTO = 0, NMFX_HSOLVE_PRODUCT = 1, NMFX_HSOLVE_POTRS = 2 };

/* NMF.Result{T} minus the matrices (src/common.jl:21-27), plus measurement fields */
typedef struct {
    int64_t niters;           /* Result.niters */
    int32_t converged;        /* Result.converged */
    int32_t status;           /* nmfx_status of the solve (NOT_POSDEF / ALPHA_NONFINITE raised on device) */
    double objvalue;          /* Result.objvalue, already rounded to T (src/common.jl:33) */
    double seconds_loop;      /* device time of the iteration loop (hipEvent), excludes H2D/D2H; iterations are enqueued a poll
                                 window ahead of the host, so after a stop it includes up to one window of no-op launches
                                 (with a communicator also their collectives; window <= 8 there) */
    int64_t inner_iters;      /* alspgrad: executed sub-solver iterations (H and W sides); greedycd: executed greedy steps */
    int64_t backtracks;       /* alspgrad: executed back-tracking steps */
    double final_tolg;        /* alspgrad: tolg after decay */
} nmfx_result;

/* ---- context ----------------------------------------------------------------
 * One context = one GPU's share of one problem: the column shard X[:, c0:c0+n_local)
 * with its H[:, c0:c0+n_local), and a full replica of W.  Single GPU: n_local = n.
 * `device` is the HIP device ordinal.  Replaces the state objects built by
 * prepare_state (src/multupd.jl:70-78,128-147; src/projals.jl:48-63;
 * src/alspgrad.jl:385-396): all temporaries live on the device for the context's life. */
int nmfx_create(nmfx_ctx **out, int dtype, int64_t p, int64_t n_local, int64_t k, int device);
void nmfx_destroy(nmfx_ctx *ctx);
const char *nmfx_last_error(const nmfx_ctx *ctx);     /* ctx may be NULL: last creation error */
const char *nmfx_version(void);

/* Upload X (host, column-major, leading dimension ldx >= p).  X is read-only
 * for solve! (src/common.jl:45-47) and re-used across `replicates` (src/interf.jl:85-101),
 * so it is uploaded once.
 * Device memory: the padded X (p and n rounded up to the tile sizes), and -- Float32 MultUpdate(:mse) on the general path and
 * one-GPU Float32 CoordinateDescent / GreedyCD / ProjectedALS (large problems), from the first such solve on -- a second, transposed image of it (+ p*n elements: 1 GiB of 288 at 16384 x 16384) that lets X*H'
 * contract over the contiguous index.  The image is an optimisation: it is taken only while enough memory stays free for the
 * buffers an iteration allocates behind it (otherwise, or with NMFX_XT=0 in the environment, X*H' runs on X as it is, ~5 % slower). */
int nmfx_set_X(nmfx_ctx *ctx, const void *X_host, int64_t ldx);
/* Same, but the source is already in device memory (bench / pipelines that keep X in HBM). */
int nmfx_set_X_device(nmfx_ctx *ctx, const void *X_dev, int64_t ldx);

/* Upload / download the factors (host, column-major, ldw = p, ldh = k). */
int nmfx_set_factors(nmfx_ctx *ctx, const void *W_host, const void *H_host);
int nmfx_get_factors(nmfx_ctx *ctx, void *W_host, void *H_host);

/* Run the iteration loop on the resident X, W, H (nmf_skeleton!, src/common.jl:45-89).
 * objv_trace: NULL or maxiter+1 doubles (entry t = objective after iteration t; entry 0 = initial). */
int nmfx_iterate(nmfx_ctx *ctx, int alg, const nmfx_opts *opts, nmfx_result *out, double *objv_trace);

/* The other two columns of the reference's verbose table (src/common.jl:54-59, :76-82) for the LAST solve run with
 * track_objective = 1: elapsed_s[t] = device time from the start of the loop to the end of iteration t (entry 0 = 0), and
 * relchange[t] = the `(W & H).relchange` value, stop_condition's devmax (src/common.jl:93, :106; entry 0 = NaN).
 * Either pointer may be NULL; *n_entries = number of entries written (= min(count, niters + 1); 0 when nothing was tracked). */
int nmfx_get_iter_trace(nmfx_ctx *ctx, double *elapsed_s, double *relchange, int count, int *n_entries);

/* NMF.solve!(alg, X, W, H) (src/multupd.jl:45-52, src/projals.jl:37-39, src/alspgrad.jl:381-383):
 * = nmfx_set_factors + nmfx_iterate + nmfx_get_factors; W and H are updated in place. */
int nmfx_solve(nmfx_ctx *ctx, int alg, const nmfx_opts *opts, void *W_host, void *H_host,
               nmfx_result *out, double *objv_trace);

/* Sub-solvers exported by the reference (src/alspgrad.jl:69-84, :225-240; test/alspgrad.jl:10-20):
 * which = 0 -> alspgrad_updateh!(X, W, H) updates H; which = 1 -> alspgrad_updatew! updates W.
 * Uses opts->maxsubiter as maxiter, opts->traceiter, tolg, beta, sigma.  out->niters = executed iterations. */
int nmfx_alspgrad_subsolve(nmfx_ctx *ctx, int which, const nmfx_opts *opts, void *W_host, void *H_host,
                           nmfx_result *out);

/* ---- nnmf() front end on the device: the parts of nnmf that touch the big arrays, done
 * next to the resident X so that `replicates` restarts never re-upload it -------------------------------------------
 *   nmfx_check_nonneg     all(t -> t >= zero(T), A) for A = X (which 0), W (1), H (2): the ArgumentError checks of
 *                         src/interf.jl:15, 28, 31 (NaN counts as a violation, as in the reference)
 *   nmfx_randinit         randinit(X, k; normalize, zeroh), src/initialization.jl:4-17: W = rand(T,p,k) with columns
 *                         scaled to sum 1 when normalize, H = rand(T,k,n) or zeros.  Julia's RNG stream cannot be
 *                         reproduced outside Julia; the generator is Philox4x32-10 keyed by `seed`, one call per element,
 *                         counter = the element's global column-major index (H columns offset by h_col_offset, the first
 *                         global column of this context's shard), so a sharded run draws the same global matrices.
 *   nmfx_solve_replicates solve_replicates!, src/interf.jl:85-101: replicate 1 solves from the given W, H; replicates
 *                         2..R from randinit(seed + r - 1, normalize = 1, zeroh); the result with the strictly smallest
 *                         objvalue wins and is returned in W, H / *out; *best_replicate (nullable) = its 1-based index.
 *   nmfx_nndsvd           the part of nndsvd() (src/initialization.jl:74-101) behind `U, s, V = ...`: _nndsvd! (:26-72) with
 *                         posnegnorm / scalepos! / scaleneg! (:103-137) on the device, filling the resident W (p x k) and
 *                         H (k x n_local; zeros when zeroh).  U is p x k (ld p), s has k entries, V is n_local x k (ld
 *                         n_local), all host, type T (or all three NULL: use the SVD left resident by nmfx_rsvd_finish):
 *                         the truncated SVD stays with the caller (the reference's
 *                         RandomizedLinAlg.rsvd, or its `initdata`).  variant 0/1/2 = :std / :a / :ar; :a and :ar fill with
 *                         mean(X) resp. mean(X)*0.01 (X must be resident; n_total = global column count for the mean),
 *                         :ar multiplies by one Philox uniform per component (Julia's rand stream is not reproducible). */
int nmfx_check_nonneg(nmfx_ctx *ctx, int which, int *all_nonneg);
int nmfx_randinit(nmfx_ctx *ctx, uint64_t seed, int normalize, int zeroh, int64_t h_col_offset);
int nmfx_solve_replicates(nmfx_ctx *ctx, int alg, const nmfx_opts *opts, int replicates, uint64_t seed, int zeroh,
                          int64_t h_col_offset, void *W_host, void *H_host, nmfx_result *out, int *best_replicate);
int nmfx_nndsvd(nmfx_ctx *ctx, const void *U_host, const void *s_host, const void *V_host, int variant, int zeroh, uint64_t seed,
                int64_t n_total);
/* rsvd(X, k) of src/initialization.jl:83 (RandomizedLinAlg.jl, un-vendored; randomized range finder + SVD of the projected
 * matrix) on the resident X.  Every p*n*k product is one of the hot path's GEMM launches: Y = X*Omega (the X*H' launch, Omega
 * Gaussian from Philox), Q = orth(Y) (Gram-Schmidt with re-orthogonalisation), B = Q'X (the W'X launch), C = B*B' (k x k).
 * The k x k symmetric eigenproblem stays with the host's LAPACK, like the reference's small svd:
 *   nmfx_rsvd_begin   runs the device half and returns C (k x k, column-major, type T).  power_iters = 0 is the plain
 *                     k-column sketch (no oversampling, no power iteration, like rsvd(X, k)'s defaults: good enough for an
 *                     initialiser, up to ~30x the optimal rank-k error); each power iteration Y = X (X'Q) costs two more
 *                     launches and brings the error to the optimum when the spectrum has a gap.
 *   nmfx_rsvd_finish  takes the eigenvectors Ub (k x k, columns ordered by DESCENDING eigenvalue) and s = sqrt(eigenvalues)
 *                     and forms U = Q*Ub (p x k) and V' = diag(1/s) Ub' B (k x n_local) on the device; U_out (p x k, ld p)
 *                     and Vt_out (k x n_local, ld k) are optional downloads.  The triple stays resident:
 *                     nmfx_nndsvd(ctx, NULL, NULL, NULL, ...) then runs _nndsvd! on it without a round trip.
 * Julia's randn stream cannot be reproduced: parity with the reference's rsvd output is unpinned by construction. */
int nmfx_rsvd_begin(nmfx_ctx *ctx, uint64_t seed, int64_t h_col_offset, int power_iters, void *C_host);
int nmfx_rsvd_finish(nmfx_ctx *ctx, const void *Ub_host, const void *s_host, void *U_out, void *Vt_out);

/* spa(X, k), src/spa.jl:38-63 -- nnmf's init = :spa (src/interf.jl:50-51) and, with the objective on top, alg = :spa
 * (src/interf.jl:73-77, src/spa.jl:66-75) -- on the resident X: fills the resident W = X[:, anchors] (p x k) and H (k x n).
 * The anchor search (k rounds of arg-max column norm + rank-1 projection of the p x n residual) is k HBM passes.  H =
 * nonneg_lsq(W, X, alg = :fnnls) comes from NonNegLeastSquares.jl in the reference (not vendored): the same published
 * active-set method (Bro & de Jong's fast NNLS on W'W, W'X; KKT tolerance 10 eps(T) ||W'W||_1, at most 30 k + 64 steps) runs
 * here with one workgroup per column of X, Float64 arithmetic inside, started from `warm_sweeps` coordinate sweeps
 * (CoordinateDescent's H sweep; 0 = cold start like the reference) -- a warm start changes the path, not the minimiser.
 * anchors_out (nullable): the k anchor column indices, 0-based, in selection order.  *unsolved_out (nullable): columns whose
 * solve hit the step cap or a Cholesky breakdown (W'W numerically singular on the passive set); they keep their last feasible
 * iterate.  Single GPU.  nmfx_get_factors / nmfx_iterate / nmfx_objective work on the result. */
int nmfx_spa_init(nmfx_ctx *ctx, int warm_sweeps, int64_t *anchors_out, int64_t *unsolved_out);

/* ---- multi-GPU (column-sharded X and H) -------------------------------------------------------------------
 * The reference has no distributed path; this is the build's data-parallel extension.
 * Rank r owns columns [c0, c0+n_local) of X and H; W is replicated between iterations.  Per outer iteration ONE exchange
 * step on the W side:
 *   NMFX_COMM_ROW_SHARDED (default): reduce-scatter of X_g H_g' by row blocks (+ a small all-reduce of H_g H_g', rowsum(H_g)
 *       and the H statistics) -- rank r then updates only ITS p/nranks rows of W (the update rules, projals' W = XH'(HH')^-1,
 *       alspgrad's W sub-problem and the cd / greedycd row sweeps are all row-separable) -- and an all-gather re-assembles W.  Same bytes on the wire as
 *       the all-reduce, but no replicated W-side work.
 *   NMFX_COMM_REPLICATED_W: one packed sum all-reduce of [X_g H_g' | H_g H_g' | rowsum(H_g)]; every rank applies the
 *       identical full W update (round-1 formulation; the fallback when p/nranks is not a whole number of 128-row tiles).
 *   NMFX_COMM_PIPELINED (opt-in): the row-sharded form with the exchange overlapped: the W side runs per row super-chunk, chunk
 *       c's reduce-scatter on a second stream under chunk c+1's X_g H_g' launch, its all-gather under the next iteration's
 *       W'X split-K part of chunk c+1.  MultUpdate-MSE only (other algorithms run as ROW_SHARDED); same results bit for bit
 *       up to the split-K grouping of the two big products.
 *   NMFX_COMM_REPLICAS: NOT a sharding of one solve but the fan-out of solve_replicates! (src/interf.jl:85-101) over the ranks.
 *       Every context holds the FULL X (n_local = n) and full W, H; nmfx_iterate / nmfx_solve run as on one GPU (no collective).
 *       nmfx_solve_replicates becomes collective: rank g runs the replicates r = g + 1, g + 1 + nranks, ... (replicate 1 -- the
 *       caller's W, H -- on rank 0; the others from randinit(seed + r - 1), the same Philox streams as the sequential loop), the
 *       objective values of all replicates are all-gathered, every rank replays the reference's scan (`if minobjv >
 *       tmp.objvalue`, in replicate order: ties and NaNs decide exactly as they do sequentially), and the winner's W, H are broadcast
 *       from the rank that computed them: every rank returns the same *out, *best_replicate and factors, bit-identical to the
 *       one-GPU call with the same seed (tests/test_gpu_localcomm.py).  Set the mode right after nmfx_comm_init*.
 * Two transports behind the same code path:
 *   one process per GPU (production, bench.py): RCCL over xGMI.  Rank 0 calls nmfx_comm_get_unique_id, the host
 *       broadcasts the 128 bytes by any means, every rank calls nmfx_comm_init.  nranks == 1 is valid.
 *   several contexts in ONE process, one host thread per context (contexts on different GPUs, or -- for testing the
 *       sharded path on a 1-GPU box -- on the same GPU): nmfx_local_group_create(nranks) once, then every context calls
 *       nmfx_comm_init_local(ctx, group, rank) from its own thread (the call is collective).  The collectives are
 *       hand-written kernels reading the peers' buffers, ordered by hipEvents; reductions add in rank order.
 * nmfx_comm_init* must precede nmfx_set_X when p is not a multiple of lcm(256, 128*nranks) (the row padding changes). */
#define NMFX_UNIQUE_ID_BYTES 128
enum { NMFX_COMM_ROW_SHARDED = 0, NMFX_COMM_REPLICATED_W = 1, NMFX_COMM_PIPELINED = 2, NMFX_COMM_REPLICAS = 3 };
typedef struct nmfx_local_group nmfx_local_group;
int nmfx_comm_get_unique_id(void *out_bytes /* NMFX_UNIQUE_ID_BYTES */);
int nmfx_comm_init(nmfx_ctx *ctx, const void *unique_id_bytes, int rank, int nranks);
int nmfx_local_group_create(nmfx_local_group **out, int nranks);   /* nranks <= 16 */
void nmfx_local_group_destroy(nmfx_local_group *group);            /* contexts still attached keep the group alive: freed when the last one is destroyed */
int nmfx_comm_init_local(nmfx_ctx *ctx, nmfx_local_group *group, int rank);
int nmfx_comm_set_mode(nmfx_ctx *ctx, int mode);
/* Peer-to-peer exchange (csrc/peer.hpp): the exchange step as a PUSH over directly mapped peer memory with device-side arrival
 * flags -- every rank stores piece q straight into rank q's window (hipIpc-mapped between processes, the pointer itself inside one
 * process), all xGMI links at once instead of a ring's one; the consumer adds the contributions in rank order (bit-identical to the
 * in-process transport).  Third transport behind the same code path:
 *   nmfx_comm_p2p_export  after any nmfx_comm_init* (and before nmfx_set_X like them): allocates this rank's window and returns
 *                         its handle (NMFX_P2P_HANDLE_BYTES).  The communicator the context already has becomes the FALLBACK for
 *                         collectives the windows cannot serve (the pipelined mode's second stream).
 *   nmfx_comm_p2p_attach  all_handles = the nranks handles in rank order (the host ships them by any means, like the unique id):
 *                         maps the peers' windows; from here on the exchange runs over them.  all_handles = NULL detaches:
 *                         the windows are unmapped and every collective goes to the wrapped transport again (what every rank
 *                         must do when mapping failed on ANY rank, so that the ranks keep routing by the same rule).
 *   nmfx_comm_init_p2p    a communicator with NO other transport (then export + attach as above): no RCCL involved; also works
 *                         with several processes on ONE device, where RCCL refuses duplicate GPUs (tests/test_gpu_peer.py).
 * Waits are bounded (NMFX_P2P_TIMEOUT_S, default 30): a rank that never arrives turns into NMFX_ERR_RCCL ("peer exchange timed
 * out") at the end of the solve instead of a hung GPU -- on EVERY rank: the rank whose wait expires raises the abort word in all
 * mapped windows.  The state is sticky: destroy the contexts and build a new communicator.  nmfx_comm_p2p_stats: collectives served by the windows / by the fallback. */
#define NMFX_P2P_HANDLE_BYTES 128
int nmfx_comm_init_p2p(nmfx_ctx *ctx, int rank, int nranks);
int nmfx_comm_p2p_export(nmfx_ctx *ctx, void *handle_out /* NMFX_P2P_HANDLE_BYTES */);
int nmfx_comm_p2p_attach(nmfx_ctx *ctx, const void *all_handles /* nranks x NMFX_P2P_HANDLE_BYTES */);
int nmfx_comm_p2p_stats(nmfx_ctx *ctx, int64_t *served_by_windows, int64_t *served_by_base);
/* Measurement aid (no reference counterpart, results are NOT a factorisation): "rank r of n" without peers -- collectives move
 * the bytes they would receive device-locally -- so the per-rank compute of the sharded path at an n-rank shard shape can be
 * timed on one GPU (bench.py --sim-ranks). */
int nmfx_comm_init_sim(nmfx_ctx *ctx, int rank, int nranks);

/* ---- standalone passes of the hot path (measurement + parity of the HBM-bound pieces) ----
 * Operate on the resident X, W, H; results are returned by value.
 *   objective: 0.5*sqL2dist(X, W*H) (alg != MULTDIV) or gkldiv(X, W*H) (alg == MULTDIV)
 *              evaluate_objv, src/multupd.jl:81,148; src/projals.jl:65-74; src/alspgrad.jl:398 */
int nmfx_objective(nmfx_ctx *ctx, int alg, const nmfx_opts *opts, double *out);

/* The SPD utilities ProjectedALS is built from (src/utils.jl:15-24 adddiag!, :34-41 projectnn!, :63-70 pdsolve!, :72-84
 * pdrsolve!; pinned by test/utils.jl:6-15, 29-34, 48-63), on the SAME device kernels the projals iteration runs (blocked
 * Cholesky + triangular inverse + MFMA products), for a context created with (p, n_local, k):
 *   nmfx_pdsolve   X (k x n_local) = inv(A + lambda I) B,   A k x k symmetric positive definite, B k x n_local
 *   nmfx_pdrsolve  X (p x k)       = A inv(B + lambda I),   A p x k, B k x k symmetric positive definite
 * lambda = 0 adds nothing (adddiag! skips it); project_nn != 0 clamps negative results to zero (projectnn!).
 * All matrices host, column-major, type T; A and B are not modified (the reference overwrites its arguments with the factor /
 * the inverse).  NMFX_ERR_NOT_POSDEF <-> PosDefException.  The resident W, H of the context are used as scratch. */
int nmfx_pdsolve(nmfx_ctx *ctx, const void *A_host, double lambda, const void *B_host, void *X_host, int project_nn);
int nmfx_pdrsolve(nmfx_ctx *ctx, const void *A_host, const void *B_host, double lambda, void *X_host, int project_nn);

/* Device/timing introspection used by bench.py (no reference counterpart). */
typedef struct {
    char name[64];
    double ms_total;      /* summed hipEvent time on the solver stream */
    int64_t launches;
    double flops;         /* algorithmic flops summed over launches */
    double bytes;         /* algorithmic HBM bytes summed over launches */
} nmfx_kernel_stat;
int nmfx_profile_enable(nmfx_ctx *ctx, int mode);  /* 0 off; 1 hipEvent pair around every launch (slow: ~10 us each);
                                                       2 only the dominant GEMM launches, every 8th (live roofline); 3: every 4th; 4: every 16th */
int nmfx_profile_get(nmfx_ctx *ctx, nmfx_kernel_stat *out, int max_entries, int *n_entries);
/* on = 1 (default): nmfx_iterate / nmfx_solve evaluate Result.objvalue after the loop, as nmf_skeleton! does (src/common.jl:85-87).
 * on = 0: they leave it NaN -- for a caller that times K iterations and asks nmfx_objective afterwards (the evaluation is one more
 * p*n*k product: 1 ms at 16384^2, k = 256, i.e. 2.5 % of a 20-iteration call). */
int nmfx_set_final_objective(nmfx_ctx *ctx, int on);
int nmfx_device_info(int device, char *name_out, int name_len, int *cu_count, int64_t *hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* NMFX_H */
