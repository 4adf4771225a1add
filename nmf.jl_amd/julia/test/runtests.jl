# runtests.jl -- the reference's hot-path tests through NMFX.DeviceMatrix, for a maintainer who has Julia, NMF.jl and an MI355X:
#
#     NMFX_LIB=/path/to/libnmfx.so julia --project=<env with NMF, StatsBase> nmf.jl_amd/julia/test/runtests.jl
#
# Ports of /root/reference test/multupd.jl:3-22, test/alspgrad.jl:3-25, test/interf.jl:6-43 and the laurberg6x3 problem of
# test/testproblems.jl:6-13: the SAME assertions, with `X` wrapped once -- `NMF.nnmf` / `NMF.solve!` themselves are NMF.jl's unmodified
# functions, the GPU is reached through dispatch on the wrapper (NMFX.jl header).  The last block compares every algorithm with NMF.jl's
# own CPU path on the same start factors (objective within 1e-5 relative for the multiplicative updates, the tolerances of
# DESIGN.md section 6 for the conditioned ones).
# Never run in the build image (no Julia there); tests/test_julia_shim.py checks what can be checked without Julia.
using NMF
using Test
using Random
using LinearAlgebra

include(joinpath(@__DIR__, "..", "NMFX.jl"))
using .NMFX

# test/testproblems.jl:6-13
function laurberg6x3(α)
    H = [α 1 1 α 0 0
         1 α 0 0 α 1
         0 0 α 1 1 α]
    W = H'
    X = W * H
    return X, Matrix(W), H
end

@testset "NMFX drop-in" begin

@testset "multupd (test/multupd.jl:3-22)" begin
    for T in (Float64, Float32)
        for alg in (:mse, :div)
            for lambda_w in (0.0, 1e-4)
                for lambda_h in (0.0, 1e-4)
                    X, Wg, Hg = laurberg6x3(T(0.3))
                    Xd = NMFX.DeviceMatrix(Matrix{T}(X))
                    W = Matrix{T}(Wg .+ rand(T, size(Wg)...) * T(0.1))
                    H = Matrix{T}(Hg)
                    NMF.solve!(NMF.MultUpdate{T}(obj=alg, maxiter=5000, tol=1e-9, lambda_w=lambda_w, lambda_h=lambda_h), Xd, W, H)
                    @test all(W .>= zero(T))
                    @test all(H .>= zero(T))
                    @test !any(isnan.(W))
                    @test !any(isnan.(H))
                    @test X ≈ W * H atol=1e-2
                    NMFX.release!(Xd)
                end
            end
        end
    end
end

@testset "alspgrad (test/alspgrad.jl:3-25)" begin
    for T in (Float64, Float32)
        X, Wg, Hg = laurberg6x3(T(0.3))
        Xd = NMFX.DeviceMatrix(Matrix{T}(X))
        Wg = Matrix{T}(Wg)
        Hg = Matrix{T}(Hg)

        H = rand(T, size(Hg)...)
        NMF.alspgrad_updateh!(Xd, Wg, H; maxiter=1000, tolg=eps(T))
        @test all(H .>= zero(T))
        @test H ≈ Hg atol=eps(T)^(1/4)

        W = rand(T, size(Wg)...)
        NMF.alspgrad_updatew!(Xd, W, Hg; maxiter=1000, tolg=eps(T))
        @test all(W .>= zero(T))
        @test W ≈ Wg atol=eps(T)^(1/4)

        NMF.solve!(NMF.ALSPGrad{T}(), Xd, W, H)
        NMFX.release!(Xd)
    end
end

@testset "interf (test/interf.jl:6-43)" begin
    p = 5
    n = 8
    k = 3
    for T in (Float64, Float32)
        Wg = max.(rand(T, p, k) .- T(0.3), zero(T))
        Hg = max.(rand(T, k, n) .- T(0.3), zero(T))
        X = Wg * Hg
        Xd = NMFX.DeviceMatrix(X)

        for alg in (:multmse, :multdiv, :projals, :alspgrad, :cd, :greedycd)
            for init in (:random, :nndsvd, :nndsvda, :nndsvdar, :spa)
                ret = NMF.nnmf(Xd, k, alg=alg, init=init)
                @test ret isa NMF.Result{T}
                @test size(ret.W) == (p, k) && size(ret.H) == (k, n)
            end
        end

        # external initialization
        F = svd(X)
        for alg in (:multmse, :multdiv, :projals, :alspgrad, :cd, :greedycd)
            ret = NMF.nnmf(Xd, k, alg=alg, init=:nndsvd, initdata=F)
        end

        # replicates test: X is uploaded once, every replicate re-uses the resident copy
        rep = NMF.nnmf(Xd, k, replicates=10, maxiter=10, alg=:multmse)
        ret = NMF.nnmf(Xd, k, W0=rep.W, H0=rep.H, init=:custom)
        @test length(Xd.ctxs) == 1

        # update_H test (test/interf.jl:30-37)
        W = max.(rand(T, p, k) .- T(0.3), zero(T))
        H = max.(rand(T, k, n) .- T(0.3), zero(T))
        for alg in (:multmse, :multdiv, :projals, :alspgrad, :cd, :greedycd)
            ret = NMF.nnmf(Xd, k, alg=alg, init=:custom, W0=copy(W), H0=copy(H), update_H=false)
            @test all(H .== ret.H)
            @test any(W .!= ret.W)
        end

        # printing test
        redirect_stdout(devnull) do
            ret = NMF.nnmf(Xd, k, alg=:cd, init=:nndsvd, verbose=true)
        end

        # errors keep their types (src/interf.jl:15, 18; src/common.jl:12)
        @test_throws ArgumentError NMF.nnmf(NMFX.DeviceMatrix(-X), k)
        @test_throws ArgumentError NMF.nnmf(Xd, min(p, n) + 1)
        @test_throws DimensionMismatch NMF.solve!(NMF.MultUpdate{T}(), Xd, rand(T, p, k), rand(T, k + 1, n))
        NMFX.release!(Xd)
    end
end

@testset "against NMF.jl's own CPU path, same start" begin
    Random.seed!(20240910)
    for T in (Float64, Float32)
        p, n, k = 200, 500, 5                       # BASELINE.json configs[0]
        X = rand(T, p, n)
        Xd = NMFX.DeviceMatrix(X)
        W0, H0 = NMF.randinit(X, k; normalize=true)
        algs = (NMF.MultUpdate{T}(obj=:mse, maxiter=50, tol=floatmin(T)), NMF.MultUpdate{T}(obj=:div, maxiter=50, tol=floatmin(T)),
                NMF.ProjectedALS{T}(maxiter=20, tol=floatmin(T)), NMF.ALSPGrad{T}(maxiter=10, tol=floatmin(T)),
                NMF.CoordinateDescent{T}(maxiter=20, tol=floatmin(T)), NMF.GreedyCD{T}(maxiter=20, tol=floatmin(T)))
        for alg in algs
            rc = NMF.solve!(alg, X, copy(W0), copy(H0))
            rg = NMF.solve!(alg, Xd, copy(W0), copy(H0))
            @test rg.niters == rc.niters
            tol = alg isa NMF.MultUpdate ? 1e-5 : (T == Float64 ? 1e-8 : 2e-3)
            @test isapprox(rg.objvalue, rc.objvalue; rtol=tol)
        end
        NMFX.release!(Xd)
    end
end

end
