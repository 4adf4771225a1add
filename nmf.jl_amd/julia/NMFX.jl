# NMFX.jl -- the Julia side of the drop-in: NMF.solve! methods that run on MI355X through libnmfx.so.
#
# Host code stays in Julia (north_star): `nnmf`, argument validation, initialisation and `solve_replicates!`
# are NMF.jl's own code (src/interf.jl:3-101), untouched.  This file only adds `solve!`-compatible entry
# points that `ccall` the C ABI of include/nmfx.h and hand back an `NMF.Result{T}` holding the SAME W and H
# arrays, updated in place, exactly like `nmf_skeleton!` does (src/common.jl:88).
#
# The drop-in itself needs NO edit of NMF.jl: wrap the data matrix once, `Xd = NMFX.DeviceMatrix(X)`, and the unmodified
# `nnmf(Xd, k; alg=..., init=..., replicates=...)` / `NMF.solve!(alg, Xd, W, H)` run on the GPU -- `DeviceMatrix{T} <: AbstractMatrix{T}`
# (accepted by `nnmf`, src/interf.jl:3), and the methods of `NMF.solve!`, `NMF.nndsvd`, `NMF.spa`, `NMF.alspgrad_updateh!/w!` added
# below are more specific in X than the reference's untyped ones (src/multupd.jl:45, projals.jl:37, alspgrad.jl:381, coorddesc.jl:49,
# greedycd.jl:33), so Julia's dispatch picks them inside `solve_replicates!` (src/interf.jl:85-101) without anybody passing a keyword.
#
# NOTE: the build image has no Julia toolchain, so this shim is untested there; it is kept trivially thin
# (one ccall per C entry point, no logic beyond marshalling and status -> exception mapping).  The Python
# twin nmf.jl_amd/nmfx/api.py exercises the same C entry points in the test-suite, tests/test_julia_shim.py checks the
# struct layouts and every ccall signature of this file against include/nmfx.h, and julia/test/runtests.jl is what a maintainer
# with Julia and a GPU runs (ports of the reference's test/multupd.jl, test/alspgrad.jl, test/interf.jl through the wrapper).
module NMFX

using NMF
import LinearAlgebra
using Printf: @printf
using LinearAlgebra: PosDefException

const libnmfx = get(ENV, "NMFX_LIB", joinpath(@__DIR__, "..", "lib", "libnmfx.so"))

# struct nmfx_opts (include/nmfx.h) -- field order and types must match the C header
struct COpts
    maxiter::Int32
    update_H::Int32
    track_objective::Int32
    maxsubiter::Int32
    traceiter::Int32
    check_every::Int32
    tol::Float64
    lambda_w::Float64
    lambda_h::Float64
    delta::Float64
    tolg::Float64
    beta::Float64
    sigma::Float64
    l1_w::Float64      # CoordinateDescentUpd's resolved regularisation (src/coorddesc.jl:62-82)
    l2_w::Float64
    l1_h::Float64
    l2_h::Float64
    precision::Int32   # 0 = fp32 (default), 1 = bf16x3 (include/nmfx.h)
    cd_shuffle::Int32  # 0 = shuffle=false; != 0: shuffle=true, permutations keyed by this value (include/nmfx.h)
    pg_refresh::Int32  # ALSPGrad gradient form: 0 = library default, 1 = exact (src/alspgrad.jl:124-127), n = refresh period
    h_solve::Int32     # ProjectedALS H solve: 0 = library default, 1 = product form, 2 = potrs! route (src/utils.jl:63-70)
    stop_sums::Int32   # 1: stop_condition's sums sequentially in T like src/common.jl:95-104 (one GPU); 0: Float64 tree sums inside the update launches
    reserved0::Int32
end

# struct nmfx_result
struct CResult
    niters::Int64
    converged::Int32
    status::Int32
    objvalue::Float64
    seconds_loop::Float64
    inner_iters::Int64
    backtracks::Int64
    final_tolg::Float64
end

const ALG_MULTMSE, ALG_MULTDIV, ALG_PROJALS, ALG_ALSPGRAD, ALG_CD, ALG_GREEDYCD = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4), Int32(5)
dtype_code(::Type{Float32}) = Int32(0)
dtype_code(::Type{Float64}) = Int32(1)

last_error(h::Ptr{Cvoid}) = unsafe_string(ccall((:nmfx_last_error, libnmfx), Cstring, (Ptr{Cvoid},), h))

# status codes of include/nmfx.h -> the exceptions the reference throws
function check(status::Integer, h::Ptr{Cvoid}=C_NULL)
    status == 0 && return
    msg = last_error(h)
    status == 1 && throw(ArgumentError(msg))                     # src/multupd.jl:27-31
    status == 2 && throw(DimensionMismatch(msg))                 # src/common.jl:12
    status == 3 && throw(PosDefException(1))                     # potrf!, src/utils.jl:68,78
    status == 4 && error("α is not finite")                      # src/alspgrad.jl:140,296
    error("nmfx status $status: $msg")
end

"""
    Context{T}(X; device=0)

Device-resident copy of `X` plus all solver temporaries (replaces the `prepare_state` objects).
Create once and reuse across `replicates` (src/interf.jl:85-101 calls `solve!` repeatedly on the same `X`).
"""
mutable struct Context{T}
    h::Ptr{Cvoid}
    p::Int
    n::Int
    k::Int
    function Context{T}(X::Matrix{T}, k::Integer; device::Integer=0) where {T<:Union{Float32,Float64}}
        p, n = size(X)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:nmfx_create, libnmfx), Cint, (Ref{Ptr{Cvoid}}, Cint, Int64, Int64, Int64, Cint),
                    ref, dtype_code(T), p, n, k, device))
        ctx = new{T}(ref[], p, n, k)
        finalizer(c -> (c.h != C_NULL && ccall((:nmfx_destroy, libnmfx), Cvoid, (Ptr{Cvoid},), c.h); c.h = C_NULL), ctx)
        check(ccall((:nmfx_set_X, libnmfx), Cint, (Ptr{Cvoid}, Ptr{T}, Int64), ctx.h, X, stride(X, 2)), ctx.h)
        return ctx
    end
end

# the verbose table of nmf_skeleton! (src/common.jl:54-59, 76-82): header, the t = 0 line (elapsed 0, objective of the start
# factors), then per iteration t: elapsed seconds, objective, its change, and stop_condition's devmax -- printed after the solve from
# the per-iteration records the device kept (objective trace of nmfx_solve, time stamps / devmax of nmfx_get_iter_trace)
function print_verbose_table(io::IO, niters::Integer, objv::Vector{Float64}, elapsed::Vector{Float64}, relchange::Vector{Float64}, ::Type{T}) where T
    @printf(io, "%-5s    %-13s    %-13s    %-13s    %-13s\n", "Iter", "Elapsed time", "objv", "objv.change", "(W & H).relchange")
    @printf(io, "%5d    %13.6e    %13.6e\n", 0, 0.0, T(objv[1]))
    for t in 1:niters
        @printf(io, "%5d    %13.6e    %13.6e    %13.6e    %13.6e\n", t, elapsed[t + 1], T(objv[t + 1]), T(objv[t + 1]) - T(objv[t]), T(relchange[t + 1]))
    end
end

function run!(ctx::Context{T}, alg::Int32, o::COpts, W::Matrix{T}, H::Matrix{T}; verbose::Bool=false, io::IO=stdout) where T
    (size(W) == (ctx.p, ctx.k) && size(H) == (ctx.k, ctx.n)) ||
        throw(DimensionMismatch("Dimensions of X, W, and H are inconsistent."))   # nmf_checksize, src/common.jl:5-16
    res = Ref{CResult}()
    # verbose = true: the objective is evaluated at t = 0 and after every iteration (src/common.jl:56, :79) -- track_objective = 1,
    # and nmfx_solve fills a (maxiter + 1)-vector with it
    trace = verbose ? fill(NaN, Int(o.maxiter) + 1) : Float64[]
    st = ccall((:nmfx_solve, libnmfx), Cint,
               (Ptr{Cvoid}, Cint, Ref{COpts}, Ptr{T}, Ptr{T}, Ref{CResult}, Ptr{Float64}),
               ctx.h, alg, o, W, H, res, verbose ? trace : C_NULL)
    check(st, ctx.h)
    r = res[]
    if verbose
        el, rc = iter_trace(ctx, Int(r.niters))
        print_verbose_table(io, Int(r.niters), trace, el, rc, T)
    end
    return NMF.Result{T}(W, H, Int(r.niters), r.converged != 0, T(r.objvalue))     # src/common.jl:21-35
end

opts(T; maxiter, tol, update_H, lambda_w=0.0, lambda_h=0.0, maxsubiter=200, tolg=eps(T)^(1/4),
     l1_w=0.0, l2_w=0.0, l1_h=0.0, l2_h=0.0, precision=0, cd_shuffle=0, verbose=false, pg_refresh=0, h_solve=0, exact_stop=false) =
    COpts(maxiter, update_H, verbose ? 1 : 0, maxsubiter, 20, 0, tol, lambda_w, lambda_h, sqrt(eps(T)), tolg, T(0.2), T(0.01),
          l1_w, l2_w, l1_h, l2_h, precision, cd_shuffle, pg_refresh, h_solve, exact_stop ? 1 : 0, 0)

# ---- solve! methods: same signatures as src/multupd.jl:45, src/projals.jl:37, src/alspgrad.jl:381, with a
# ---- leading device Context.  `solve!(alg, X, W, H)` without a Context creates one for the call.
function solve!(ctx::Context{T}, alg::NMF.MultUpdate{T}, W::Matrix{T}, H::Matrix{T}) where T
    a = alg.obj == :mse ? ALG_MULTMSE : ALG_MULTDIV
    run!(ctx, a, opts(T; maxiter=alg.maxiter, tol=alg.tol, update_H=alg.update_H,
                      lambda_w=alg.lambda_w, lambda_h=alg.lambda_h, verbose=alg.verbose), W, H; verbose=alg.verbose)
end

solve!(ctx::Context{T}, alg::NMF.ProjectedALS{T}, W::Matrix{T}, H::Matrix{T}) where T =
    run!(ctx, ALG_PROJALS, opts(T; maxiter=alg.maxiter, tol=alg.tol, update_H=alg.update_H,
                                lambda_w=alg.lambda_w, lambda_h=alg.lambda_h, verbose=alg.verbose), W, H; verbose=alg.verbose)

solve!(ctx::Context{T}, alg::NMF.ALSPGrad{T}, W::Matrix{T}, H::Matrix{T}) where T =
    run!(ctx, ALG_ALSPGRAD, opts(T; maxiter=alg.maxiter, tol=alg.tol, update_H=alg.update_H,
                                 maxsubiter=alg.maxsubiter, tolg=alg.tolg, verbose=alg.verbose), W, H; verbose=alg.verbose)

# CoordinateDescent (src/coorddesc.jl:54-56): the l1/l2 pairs are resolved exactly like CoordinateDescentUpd's constructor
# (src/coorddesc.jl:62-82).  shuffle = true: the component orders come from the library's documented Philox generator
# (Julia's randperm stream is not reproducible on the device); the key is drawn here from Julia's RNG, so `Random.seed!`
# still makes a run repeatable.
function solve!(ctx::Context{T}, alg::NMF.CoordinateDescent{T}, W::Matrix{T}, H::Matrix{T}) where T
    u = NMF.CoordinateDescentUpd{T}(alg.α, alg.l₁ratio, alg.regularization, alg.shuffle, alg.update_H)
    key = alg.shuffle ? Int32(rand(1:typemax(Int32))) : Int32(0)
    run!(ctx, ALG_CD, opts(T; maxiter=alg.maxiter, tol=alg.tol, update_H=alg.update_H,
                           l1_w=u.l₁W, l2_w=u.l₂W, l1_h=u.l₁H, l2_h=u.l₂H, cd_shuffle=key, verbose=alg.verbose), W, H; verbose=alg.verbose)
end

# GreedyCD (src/greedycd.jl:34-35)
solve!(ctx::Context{T}, alg::NMF.GreedyCD{T}, W::Matrix{T}, H::Matrix{T}) where T =
    run!(ctx, ALG_GREEDYCD, opts(T; maxiter=alg.maxiter, tol=alg.tol, update_H=alg.update_H,
                                 lambda_w=alg.lambda_w, lambda_h=alg.lambda_h, verbose=alg.verbose), W, H; verbose=alg.verbose)

function solve!(alg::Union{NMF.MultUpdate{T},NMF.ProjectedALS{T},NMF.ALSPGrad{T},NMF.CoordinateDescent{T},NMF.GreedyCD{T}},
                X::Matrix{T}, W::Matrix{T}, H::Matrix{T}; device::Integer=0) where T
    ctx = Context{T}(X, size(W, 2); device=device)
    try
        return solve!(ctx, alg, W, H)
    finally
        finalize(ctx)
    end
end

# the `Elapsed time` and `(W & H).relchange` columns of the verbose table (src/common.jl:54-59, 76-82) of the last solve
# that ran with track_objective = 1
function iter_trace(ctx::Context, niters::Integer)
    el = zeros(Float64, niters + 1); rc = fill(NaN, niters + 1); m = Ref{Cint}(0)
    check(ccall((:nmfx_get_iter_trace, libnmfx), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Ref{Cint}),
                ctx.h, el, rc, niters + 1, m), ctx.h)
    el[1:m[]], rc[1:m[]]
end

get_factors!(ctx::Context{T}, W::Matrix{T}, H::Matrix{T}) where T =
    check(ccall((:nmfx_get_factors, libnmfx), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), ctx.h, W, H), ctx.h)

# alspgrad_updateh!(X, W, H) / alspgrad_updatew!(X, W, H) (src/alspgrad.jl:69-84, 225-240) on the resident X: which = 0 updates H,
# which = 1 updates W; returns the executed inner iterations
function alspgrad_subsolve!(ctx::Context{T}, which::Integer, W::Matrix{T}, H::Matrix{T}; maxiter::Integer=1000, traceiter::Integer=20,
                            tolg::Real=cbrt(eps(T)), beta::Real=T(0.2), sigma::Real=T(0.01)) where T
    o = COpts(1, 1, 0, maxiter, traceiter, 0, cbrt(eps(T)), 0.0, 0.0, sqrt(eps(T)), tolg, beta, sigma, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0)
    res = Ref(CResult(0, 0, 0, 0.0, 0.0, 0, 0, 0.0))
    check(ccall((:nmfx_alspgrad_subsolve, libnmfx), Cint, (Ptr{Cvoid}, Cint, Ref{COpts}, Ptr{T}, Ptr{T}, Ref{CResult}),
                ctx.h, which, o, W, H, res), ctx.h)
    Int(res[].niters)
end

# ---- DeviceMatrix: the zero-edit drop-in (see the header) ---------------------------------------------------------------
"""
    DeviceMatrix(X; device=0)

`X` as NMF.jl sees it (an `AbstractMatrix{T}` backed by the host array) plus its device-resident copies, created on first use --
one `Context` per component count `k` -- and kept for every later `solve!` / replicate / initialisation on the same matrix.
`NMFX.release!(Xd)` frees the device memory early (otherwise the finalizers do).
"""
mutable struct DeviceMatrix{T<:Union{Float32,Float64}} <: AbstractMatrix{T}
    X::Matrix{T}
    device::Int
    ctxs::Dict{Int,Context{T}}
end
DeviceMatrix(X::Matrix{T}; device::Integer=0) where {T<:Union{Float32,Float64}} = DeviceMatrix{T}(X, Int(device), Dict{Int,Context{T}}())
DeviceMatrix(X::AbstractMatrix{T}; device::Integer=0) where {T<:Union{Float32,Float64}} = DeviceMatrix(Matrix{T}(X); device=device)
Base.size(A::DeviceMatrix) = size(A.X)
Base.IndexStyle(::Type{<:DeviceMatrix}) = IndexLinear()
Base.getindex(A::DeviceMatrix, i::Int) = A.X[i]
Base.parent(A::DeviceMatrix) = A.X
context(A::DeviceMatrix{T}, k::Integer) where T = get!(() -> Context{T}(A.X, k; device=A.device), A.ctxs, Int(k))
function release!(A::DeviceMatrix)
    foreach(finalize, values(A.ctxs))
    empty!(A.ctxs)
    A
end

# NMF.solve!(alg, X, W, H) for every iterative algorithm (one method per type: a Union in the first argument would be ambiguous
# with the reference's methods, which are more specific there and less specific in X)
NMF.solve!(alg::NMF.MultUpdate{T}, X::DeviceMatrix{T}, W::Matrix{T}, H::Matrix{T}) where T = solve!(context(X, size(W, 2)), alg, W, H)
NMF.solve!(alg::NMF.ProjectedALS{T}, X::DeviceMatrix{T}, W::Matrix{T}, H::Matrix{T}) where T = solve!(context(X, size(W, 2)), alg, W, H)
NMF.solve!(alg::NMF.ALSPGrad{T}, X::DeviceMatrix{T}, W::Matrix{T}, H::Matrix{T}) where T = solve!(context(X, size(W, 2)), alg, W, H)
NMF.solve!(alg::NMF.CoordinateDescent{T}, X::DeviceMatrix{T}, W::Matrix{T}, H::Matrix{T}) where T = solve!(context(X, size(W, 2)), alg, W, H)
NMF.solve!(alg::NMF.GreedyCD{T}, X::DeviceMatrix{T}, W::Matrix{T}, H::Matrix{T}) where T = solve!(context(X, size(W, 2)), alg, W, H)

# nndsvd(X, k; zeroh, variant, initdata) (src/initialization.jl:74-101) next to the resident X: rsvd's sketch / orthogonalisation /
# projection and _nndsvd! run on the device, the k x k eigenproblem in Julia's LAPACK.  The sketch and the :ar fill are drawn by the
# library's Philox generator keyed by ONE draw from Julia's RNG, so `Random.seed!` still makes a run repeatable.
# (`randinit(X, k)` needs no method: it only asks for size and eltype, and keeps Julia's own random stream.)
function NMF.nndsvd(X::DeviceMatrix{T}, k::Integer; zeroh::Bool=false, variant::Symbol=:std, initdata=nothing) where T
    p, n = size(X)
    variant in (:std, :a, :ar) || throw(ArgumentError("Invalid value for variant"))
    ctx = context(X, k)
    seed = rand(UInt64)
    if initdata === nothing
        rsvd!(ctx, k; seed=seed)
        nndsvd_resident!(ctx; variant=variant, zeroh=zeroh, seed=seed, n_total=n)
    else
        nndsvd!(ctx, Matrix{T}(initdata.U[:, 1:k]), Vector{T}(initdata.S[1:k]), Matrix{T}(initdata.V[:, 1:k]);
                variant=variant, zeroh=zeroh, seed=seed, n_total=n)
    end
    W = Matrix{T}(undef, p, k)
    H = Matrix{T}(undef, k, n)
    get_factors!(ctx, W, H)
    return (W, H)
end

# spa(X, k) (src/spa.jl:41-63; the reference's method is typed on Matrix{T}, so `nnmf(Xd, k; init=:spa)` needs this one)
function NMF.spa(X::DeviceMatrix{T}, k::Integer; nnls_alg::Tuple{Symbol,Symbol}=(:pivot, :cache)) where T
    p, n = size(X)
    W = Matrix{T}(undef, p, k)
    H = Matrix{T}(undef, k, n)
    spa!(context(X, k), W, H)
    return W, H
end

# the exported sub-solvers (src/alspgrad.jl:69-84, 225-240; test/alspgrad.jl:10-20); like the reference they return (H, t) / (W, t)
function NMF.alspgrad_updateh!(X::DeviceMatrix{T}, W::Matrix{T}, H::Matrix{T}; maxiter::Int=1000, traceiter::Int=20, tolg::T=cbrt(eps(T)),
                               beta::T=convert(T, 0.2), sigma::T=convert(T, 0.01), verbose::Bool=false) where T
    t = alspgrad_subsolve!(context(X, size(W, 2)), 0, W, H; maxiter=maxiter, traceiter=traceiter, tolg=tolg, beta=beta, sigma=sigma)
    return (H, t)
end
function NMF.alspgrad_updatew!(X::DeviceMatrix{T}, W::Matrix{T}, H::Matrix{T}; maxiter::Int=1000, traceiter::Int=20, tolg::T=cbrt(eps(T)),
                               beta::T=convert(T, 0.2), sigma::T=convert(T, 0.01), verbose::Bool=false) where T
    t = alspgrad_subsolve!(context(X, size(W, 2)), 1, W, H; maxiter=maxiter, traceiter=traceiter, tolg=tolg, beta=beta, sigma=sigma)
    return (W, t)
end

# ---- nnmf front end on the device (include/nmfx.h; src/interf.jl:15,28,31,85-101; src/initialization.jl:4-17) ----------
# all(t -> t >= zero(T), A) for the resident X (which = 0), W (1), H (2)
function check_nonneg(ctx::Context, which::Integer)
    ok = Ref{Cint}(0)
    check(ccall((:nmfx_check_nonneg, libnmfx), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}), ctx.h, which, ok), ctx.h)
    ok[] != 0
end

# randinit(X, k; normalize, zeroh) into the resident W, H (Philox4x32-10 keyed by `seed`; Julia's own stream is not used)
randinit!(ctx::Context, seed::Integer; normalize::Bool=false, zeroh::Bool=false, h_col_offset::Integer=0) =
    check(ccall((:nmfx_randinit, libnmfx), Cint, (Ptr{Cvoid}, UInt64, Cint, Cint, Int64),
                ctx.h, seed, normalize, zeroh, h_col_offset), ctx.h)

# rsvd(X, k) (src/initialization.jl:83) on the resident X: the device sketches / orthogonalises / projects, Julia's LAPACK solves
# the k x k symmetric eigenproblem in between; the triple stays resident for nndsvd_resident!
function rsvd!(ctx::Context{T}, k::Integer; seed::Integer=0, power_iters::Integer=0) where T
    C = Matrix{T}(undef, k, k)
    check(ccall((:nmfx_rsvd_begin, libnmfx), Cint, (Ptr{Cvoid}, UInt64, Int64, Cint, Ptr{T}), ctx.h, seed, 0, power_iters, C), ctx.h)
    F = LinearAlgebra.eigen(LinearAlgebra.Symmetric((C + C') / 2))
    ord = sortperm(F.values; rev=true)
    s = T.(sqrt.(max.(F.values[ord], 0)))
    Ub = Matrix{T}(F.vectors[:, ord])
    check(ccall((:nmfx_rsvd_finish, libnmfx), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, Ub, s, C_NULL, C_NULL), ctx.h)
    s
end
nndsvd_resident!(ctx::Context{T}; variant::Symbol=:std, zeroh::Bool=false, seed::Integer=0, n_total::Integer) where T =
    check(ccall((:nmfx_nndsvd, libnmfx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, UInt64, Int64),
                ctx.h, C_NULL, C_NULL, C_NULL, variant == :std ? 0 : variant == :a ? 1 : 2, zeroh, seed, n_total), ctx.h)

# nndsvd(X, k; zeroh, variant, initdata) after its `U, s, V = ...` line (src/initialization.jl:83): the SVD stays in Julia
# (rsvd(X, k) or initdata), _nndsvd! runs on the device and fills the resident W, H
function nndsvd!(ctx::Context{T}, U::Matrix{T}, s::Vector{T}, V::Matrix{T}; variant::Symbol=:std, zeroh::Bool=false,
                 seed::Integer=0, n_total::Integer=size(V, 1)) where T
    ivar = variant == :std ? 0 : variant == :a ? 1 : variant == :ar ? 2 : throw(ArgumentError("Invalid value for variant"))
    check(ccall((:nmfx_nndsvd, libnmfx), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Cint, Cint, UInt64, Int64),
                ctx.h, U, s, V, ivar, zeroh, seed, n_total), ctx.h)
end

# solve_replicates!(alginst, X, W, H; replicates, initH): X stays on the device, only the winner comes back
function solve_replicates!(ctx::Context{T}, alg::Int32, o::COpts, W::Matrix{T}, H::Matrix{T};
                           replicates::Integer, initH::Bool, seed::Integer) where T
    res = Ref(CResult(0, 0, 0, 0.0, 0.0, 0, 0, 0.0))
    best = Ref{Cint}(0)
    st = ccall((:nmfx_solve_replicates, libnmfx), Cint,
               (Ptr{Cvoid}, Cint, Ref{COpts}, Cint, UInt64, Cint, Int64, Ptr{T}, Ptr{T}, Ref{CResult}, Ref{Cint}),
               ctx.h, alg, o, replicates, seed, !initH, 0, W, H, res, best)
    check(st, ctx.h)
    NMF.Result{T}(W, H, Int(res[].niters), res[].converged != 0, T(res[].objvalue))
end

# ---- spa(X, k) (src/spa.jl:38-63) on the resident X: anchors (1-based, selection order), W = X[:, anchors], H by the same
# active-set NNLS the reference takes from NonNegLeastSquares.fnnls (include/nmfx.h: nmfx_spa_init).  In NMF.jl:
#   spa(X::Matrix{T}, k) = (ctx = NMFX.Context{T}(X, k); NMFX.spa!(ctx, Matrix{T}(undef, size(X,1), k), Matrix{T}(undef, k, size(X,2)))[1:2])
function spa!(ctx::Context{T}, W::Matrix{T}, H::Matrix{T}; warm_sweeps::Integer=16) where T
    anchors = Vector{Int64}(undef, size(W, 2))
    unsolved = Ref{Int64}(0)
    check(ccall((:nmfx_spa_init, libnmfx), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ref{Int64}), ctx.h, warm_sweeps, anchors, unsolved), ctx.h)
    check(ccall((:nmfx_get_factors, libnmfx), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), ctx.h, W, H), ctx.h)
    unsolved[] == 0 || @warn "spa: $(unsolved[]) columns left the active-set solve at its step cap"
    W, H, anchors .+ 1
end

# ---- the SPD utilities of src/utils.jl on the device kernels ProjectedALS runs (include/nmfx.h: nmfx_pdsolve / nmfx_pdrsolve) ----
# pdsolve!(A, x) : x <- inv(A + lambda I) x   (adddiag! + pdsolve!, src/utils.jl:15-24, 63-70); A is k x k SPD, x is k x n
function pdsolve!(ctx::Context{T}, A::Matrix{T}, x::Matrix{T}; lambda::Real=0, projectnn::Bool=false) where T
    check(ccall((:nmfx_pdsolve, libnmfx), Cint, (Ptr{Cvoid}, Ptr{T}, Float64, Ptr{T}, Ptr{T}, Cint), ctx.h, A, lambda, x, x, projectnn), ctx.h)
    x
end
# pdrsolve!(A, B, x) : x <- A inv(B + lambda I)   (adddiag! + pdrsolve!, src/utils.jl:72-84); A, x are p x k, B is k x k SPD
function pdrsolve!(ctx::Context{T}, A::Matrix{T}, B::Matrix{T}, x::Matrix{T}; lambda::Real=0, projectnn::Bool=false) where T
    check(ccall((:nmfx_pdrsolve, libnmfx), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Float64, Ptr{T}, Cint), ctx.h, A, B, lambda, x, projectnn), ctx.h)
    x
end

# ---- multi-GPU from ONE Julia process: one task (thread) per Context, all attached to a local group (include/nmfx.h) --------
# g = local_group(nranks); Threads.@spawn per rank: ctx = Context{T}(...) on its device; attach!(ctx, g, rank) BEFORE the X upload
# changes anything (the row padding depends on nranks); then solve! as usual with the rank's column shard of X and H.
local_group(nranks::Integer) = (r = Ref{Ptr{Cvoid}}(C_NULL);
                                check(ccall((:nmfx_local_group_create, libnmfx), Cint, (Ref{Ptr{Cvoid}}, Cint), r, nranks)); r[])
free_local_group(g::Ptr{Cvoid}) = ccall((:nmfx_local_group_destroy, libnmfx), Cvoid, (Ptr{Cvoid},), g)
attach!(ctx::Context, g::Ptr{Cvoid}, rank::Integer) =
    check(ccall((:nmfx_comm_init_local, libnmfx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), ctx.h, g, rank), ctx.h)

# ---- peer-to-peer exchange (include/nmfx.h nmfx_comm_p2p_*; DESIGN.md section 4): one process per GPU, the handles travel through
# the host (e.g. MPI.Allgather); `p2p_attach!` after every rank's `p2p_export` and before the X upload.
p2p_init!(ctx::Context, rank::Integer, nranks::Integer) =
    check(ccall((:nmfx_comm_init_p2p, libnmfx), Cint, (Ptr{Cvoid}, Cint, Cint), ctx.h, rank, nranks), ctx.h)
function p2p_export(ctx::Context)
    h = Vector{UInt8}(undef, 128)
    check(ccall((:nmfx_comm_p2p_export, libnmfx), Cint, (Ptr{Cvoid}, Ptr{UInt8}), ctx.h, h), ctx.h)
    h
end
p2p_attach!(ctx::Context, handles::Vector{UInt8}) =
    check(ccall((:nmfx_comm_p2p_attach, libnmfx), Cint, (Ptr{Cvoid}, Ptr{UInt8}), ctx.h, handles), ctx.h)

end # module
