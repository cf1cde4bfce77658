# CompressedSensingAMD.jl -- Julia host for libcsmp.so (MI355X / gfx950).
#
# Keeps CompressedSensing.jl's matching-pursuit API surface -- mp / omp / gomp / sp with the same
# positional and keyword forms, the MP / OMP / GOMP update functors -- and forwards the work through
# `ccall` to the C ABI declared in include/csmp.h.  Results are `SparseVector{Float64,Int}` with
# sorted 1-based `nzind`, exactly what the reference returns (src/matchingpursuit.jl:76).
#
# NOTE: the build container has no `julia`, so this file has never been executed; it is the
# reference-side binding a maintainer adds (see INTEGRATION.md).  The same ABI is exercised by the
# Python/ctypes mirror (compressedsensing.jl_amd/_lib.py, api.py), which the test suite drives.
module CompressedSensingAMD

using LinearAlgebra
using SparseArrays
using Random

const libcsmp = get(ENV, "LIBCSMP", joinpath(@__DIR__, "..", "csrc", "libcsmp.so"))

const CSMP_F32, CSMP_F64 = Cint(0), Cint(1)
const CSMP_HOST, CSMP_DEVICE, CSMP_HOST_STREAMED = Cint(0), Cint(1), Cint(2)
const ALGO_MP, ALGO_OMP, ALGO_GOMP, ALGO_FR, ALGO_SP, ALGO_OMPR = Cint(0), Cint(1), Cint(2), Cint(3), Cint(4), Cint(5)

dtype_code(::Type{Float32}) = CSMP_F32
dtype_code(::Type{Float64}) = CSMP_F64

# ---------------------------------------------------------------------------------- context
"""
    Dictionary(A; device = 0, streamed = false)
    Dictionary(path::AbstractString; device = 0, streamed = false)

The measurement matrix resident in HBM (uploaded once).  Stands in for the `A` field of the
reference's MP / OMP / GOMP / SP structs.  `streamed = true` leaves it in host memory, mapped into the
device: every sweep reads it over the host link -- for a dictionary larger than HBM (the array is kept
alive by the object and must not be changed).  `path`: a dictionary file (`write_dictionary_file`).
"""
mutable struct Dictionary{T<:Union{Float32,Float64}}
    ctx::Ptr{Cvoid}
    n::Int   # rows  (reference: n, m = size(A), src/matchingpursuit.jl:20)
    m::Int   # atoms
    keep::Any  # a streamed array, held for the context's lifetime
    function Dictionary{T}(n::Integer, m::Integer, device::Integer) where {T<:Union{Float32,Float64}}
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:csmp_create, libcsmp), Cint, (Ref{Ptr{Cvoid}}, Cint), ref, device)
        rc == 0 || throw(unsafe_string(ccall((:csmp_last_error, libcsmp), Cstring, (Ptr{Cvoid},), C_NULL)))
        D = new{T}(ref[], n, m, nothing)
        finalizer(d -> ccall((:csmp_destroy, libcsmp), Cint, (Ptr{Cvoid},), d.ctx), D)
        return D
    end
end
function Dictionary(A::StridedMatrix{T}; device::Integer = 0, streamed::Bool = false) where {T<:Union{Float32,Float64}}
    D = Dictionary{T}(size(A, 1), size(A, 2), device)
    GC.@preserve A check(D, ccall((:csmp_set_dictionary, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint),
        D.ctx, pointer(A), size(A, 1), size(A, 2), stride(A, 2), dtype_code(T), streamed ? CSMP_HOST_STREAMED : CSMP_HOST))
    streamed && (D.keep = A)
    return D
end
function Dictionary(path::AbstractString; device::Integer = 0, streamed::Bool = false)
    n, m, dt = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0)
    rc = ccall((:csmp_dictionary_file_info, libcsmp), Cint, (Cstring, Ref{Int64}, Ref{Int64}, Ref{Cint}), path, n, m, dt)
    rc == 0 || throw("$path is not a dictionary file")
    D = Dictionary{dt[] == CSMP_F32 ? Float32 : Float64}(n[], m[], device)
    check(D, ccall((:csmp_set_dictionary_file, libcsmp), Cint, (Ptr{Cvoid}, Cstring, Cint), D.ctx, path, streamed ? CSMP_HOST_STREAMED : CSMP_DEVICE))
    return D
end

"""
    write_dictionary_file(path, A)

`A` as a dictionary file (include/csmp.h: a 64-byte header and the columns, padded to 16 bytes); host only.
"""
function write_dictionary_file(path::AbstractString, A::StridedMatrix{T}) where {T<:Union{Float32,Float64}}
    rc = GC.@preserve A ccall((:csmp_dictionary_file_write, libcsmp), Cint, (Cstring, Ptr{Cvoid}, Int64, Int64, Int64, Cint),
        path, pointer(A), size(A, 1), size(A, 2), stride(A, 2), dtype_code(T))
    rc == 0 || throw("cannot write $path")
    return path
end
Base.size(D::Dictionary) = (D.n, D.m)
Base.size(D::Dictionary, i::Int) = size(D)[i]
Base.eltype(::Dictionary{T}) where {T} = T

# the reference throws bare strings (src/matchingpursuit.jl:74; src/twostage.jl:76): so do we
function check(D::Dictionary, rc::Integer)
    rc == 0 && return
    throw(unsafe_string(ccall((:csmp_last_error, libcsmp), Cstring, (Ptr{Cvoid},), D.ctx)))
end

# ---------------------------------------------------------------------------------- options (include/csmp.h, CSMP_OPT_*)
# The reference passes its behavioural choices as arguments and so does this module; the choices that exist only on the GPU
# side are per-Dictionary options.  The library reads no environment variable.
const OPTIONS = Dict(:batch_cert => 1, :batch_gram => 2, :batch_window => 3, :pipeline => 4, :solves_in_flight => 9, :screened_sweep => 10,
                     :batch_screen => 11)
set_option!(D::Dictionary, key::Symbol, value::Integer) =
    check(D, ccall((:csmp_set_option, libcsmp), Cint, (Ptr{Cvoid}, Cint, Int64), D.ctx, OPTIONS[key], value))
function get_option(D::Dictionary, key::Symbol)
    v = Ref{Int64}(0)
    check(D, ccall((:csmp_get_option, libcsmp), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}), D.ctx, OPTIONS[key], v))
    v[]
end

const MatOrDict{T} = Union{StridedMatrix{T}, Dictionary{T}}
dict(A::Dictionary) = A
dict(A::StridedMatrix) = Dictionary(A)

function to_sparse(m::Int, idx::Vector{Int64}, val::Vector{Float64}, nnz::Integer)
    # 0-based sorted indices from the ABI -> SparseVector{Float64,Int} (src/matchingpursuit.jl:76)
    SparseVector(m, idx[1:nnz] .+ 1, val[1:nnz])
end

bvec(b::AbstractVector{Float32}) = (convert(Vector{Float32}, b), CSMP_F32)
bvec(b::AbstractVector) = (convert(Vector{Float64}, b), CSMP_F64)

# ---------------------------------------------------------------------------------- omp
# src/matchingpursuit.jl:73-91
function omp(A::MatOrDict{T}, b::AbstractVector, ε::Real, k::Int = size(A, 1)) where {T}
    ε ≥ 0 || throw("ε = $ε has to be non-negative")
    D = dict(A)
    bb, bt = bvec(b)
    idx, val, nnz = zeros(Int64, max(k, 1)), zeros(Float64, max(k, 1)), Ref{Int64}(0)
    GC.@preserve bb idx val check(D, ccall((:csmp_omp, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ptr{Int64}),
        D.ctx, bb, bt, k, ε, idx, val, nnz, C_NULL))
    to_sparse(size(D, 2), idx, val, nnz[])
end
omp(A::MatOrDict{T}, b::AbstractVector, k::Int) where {T} = omp(A, b, eps(T), k)
omp(A::MatOrDict{T}, b::AbstractVector; max_residual = eps(T), sparsity = min(size(A)...)) where {T} =
    omp(A, b, max_residual, sparsity)

# ---------------------------------------------------------------------------------- gomp
# src/matchingpursuit.jl:126-148
function gomp(A::MatOrDict{T}, b::AbstractVector, l::Int, ε::Real, k::Int = size(A, 1)) where {T}
    ε ≥ 0 || throw("ε = $ε has to be non-negative")
    D = dict(A)
    bb, bt = bvec(b)
    cap = max(k + l, 1)
    idx, val, nnz = zeros(Int64, cap), zeros(Float64, cap), Ref{Int64}(0)
    GC.@preserve bb idx val check(D, ccall((:csmp_gomp, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Int64, Cdouble, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ptr{Int64}),
        D.ctx, bb, bt, l, k, ε, idx, val, nnz, C_NULL))
    to_sparse(size(D, 2), idx, val, nnz[])
end
gomp(A::MatOrDict{T}, b::AbstractVector, l::Int, k::Int) where {T} = gomp(A, b, l, eps(T), k)
gomp(A::MatOrDict{T}, b::AbstractVector, l::Int; max_residual = eps(T), sparsity = size(A, 2)) where {T} =
    gomp(A, b, l, max_residual, sparsity)

# ---------------------------------------------------------------------------------- mp
# src/matchingpursuit.jl:34-40 (x is a warm start and is updated in place)
function mp(A::MatOrDict{T}, b::AbstractVector, k::Int, x::SparseVector = spzeros(size(A, 2))) where {T}
    D = dict(A)
    bb, bt = bvec(b)
    idx0, val0 = convert(Vector{Int64}, x.nzind .- 1), convert(Vector{Float64}, x.nzval)
    cap = max(k + length(idx0), 1)
    idx, val, nnz = zeros(Int64, cap), zeros(Float64, cap), Ref{Int64}(0)
    GC.@preserve bb idx0 val0 idx val check(D, ccall((:csmp_mp, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Int64}, Ptr{Cdouble}, Int64, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}),
        D.ctx, bb, bt, k, idx0, val0, length(idx0), idx, val, nnz))
    y = to_sparse(size(D, 2), idx, val, nnz[])
    resize!(x.nzind, nnz[]); resize!(x.nzval, nnz[])
    copyto!(x.nzind, y.nzind); copyto!(x.nzval, y.nzval)
    return x
end

# ---------------------------------------------------------------------------------- sp
# src/twostage.jl:87-101
function sp(A::MatOrDict{T}, b::AbstractVector, k::Int, δ::Real = 1e-12; maxiter = 16k) where {T}
    2k > length(b) && error("2k = $(2k) > $(length(b)) = length(b) is invalid for Subspace Pursuit")
    D = dict(A)
    bb, bt = bvec(b)
    idx, val, nnz, iters = zeros(Int64, 2k), zeros(Float64, 2k), Ref{Int64}(0), Ref{Int64}(0)
    GC.@preserve bb idx val check(D, ccall((:csmp_sp, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Int64, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ref{Int64}),
        D.ctx, bb, bt, k, δ, maxiter, idx, val, nnz, iters))
    to_sparse(size(D, 2), idx, val, nnz[])
end

# ---------------------------------------------------------------------------------- ompr
# src/twostage.jl:184-202 (x starting empty: the support is filled by oblivious_acquisition!)
function ompr(A::MatOrDict{T}, b::AbstractVector, k::Int, δ::Real; maxiter = size(A, 1)) where {T}
    D = dict(A)
    bb, bt = bvec(b)
    idx, val, nnz, iters = zeros(Int64, k), zeros(Float64, k), Ref{Int64}(0), Ref{Int64}(0)
    GC.@preserve bb idx val check(D, ccall((:csmp_ompr, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Int64, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ref{Int64}),
        D.ctx, bb, bt, k, δ, maxiter, idx, val, nnz, iters))
    to_sparse(size(D, 2), idx, val, nnz[])
end

# ---------------------------------------------------------------------------------- fr = ols = oomp = ormp
# src/forward.jl:34-54 (x starting empty)
function fr(A::MatOrDict{T}, b::AbstractVector, max_ε::Real, min_δ::Real, k::Int = size(A, 1)) where {T}
    size(A, 1) == length(b) || throw(DimensionMismatch("size(A, 1) = $(size(A, 1)) ≠ $(length(b)) = length(b)"))
    D = dict(A)
    bb, bt = bvec(b)
    k = min(k, size(A, 1))
    idx, val, nnz = zeros(Int64, max(k, 1)), zeros(Float64, max(k, 1)), Ref{Int64}(0)
    GC.@preserve bb idx val check(D, ccall((:csmp_fr, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Cdouble, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ptr{Int64}),
        D.ctx, bb, bt, k, max_ε, min_δ, idx, val, nnz, C_NULL))
    to_sparse(size(D, 2), idx, val, nnz[])
end
fr(A::MatOrDict, b::AbstractVector; max_residual::Real = 0., min_decrease::Real = 0., sparsity::Int = size(A, 2)) =
    fr(A, b, max_residual, min_decrease, sparsity)
const ols = fr
const oomp = fr
const ormp = fr

# ---------------------------------------------------------------------------------- srr
# src/twostage.jl:3-33 (x starting empty; initialization 1 = oblivious, 2 = forward regression, 3 = random: the k atoms are
# drawn here without replacement (randperm; the reference: sample(1:n, k, replace = false), src/matchingpursuit.jl:196) or taken
# from `init` (1-based), and handed to csmp_srr_from)
function srr(A::MatOrDict{T}, b::AbstractVector, k::Int, δ::Real = 1e-12; maxiter = 4k,
             initialization::Int = 1, l::Int = 1, init = nothing) where {T}
    D = dict(A)
    bb, bt = bvec(b)
    idx, val, nnz, iters = zeros(Int64, k + l + 1), zeros(Float64, k + l + 1), Ref{Int64}(0), Ref{Int64}(0)
    if initialization == 3
        ind = init === nothing ? randperm(size(D, 2))[1:k] : collect(init)
        ind0 = Int64.(ind) .- 1
        GC.@preserve bb idx val ind0 check(D, ccall((:csmp_srr_from, libcsmp), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Int64, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ref{Int64}),
            D.ctx, bb, bt, k, δ, maxiter, ind0, l, idx, val, nnz, iters))
        return to_sparse(size(D, 2), idx, val, nnz[])
    end
    GC.@preserve bb idx val check(D, ccall((:csmp_srr, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cdouble, Int64, Cint, Int64, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ref{Int64}),
        D.ctx, bb, bt, k, δ, maxiter, initialization, l, idx, val, nnz, iters))
    to_sparse(size(D, 2), idx, val, nnz[])
end

# ---------------------------------------------------------------------------------- rmp, foba
# src/stepwise.jl:5-56 (x starting empty).  kmax bounds the support the forward stage may build (<= 4095).
# (one literal ccall per entry point: ccall needs its symbol and its argument types as constants)
function stepwise_out(A::MatOrDict, kmax::Int)
    cap = min(size(A, 1), size(A, 2), 4095, kmax > 0 ? kmax : typemax(Int))
    cap, zeros(Int64, cap + 1), zeros(Float64, cap + 1), Ref{Int64}(0)
end
function rmp(A::MatOrDict, b::AbstractVector, δ::Real, maxiter::Int = 1; kmax::Int = 0)
    D = dict(A)
    bb, bt = bvec(b)
    cap, idx, val, nnz = stepwise_out(A, kmax)
    GC.@preserve bb idx val check(D, ccall((:csmp_rmp_delta, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Int64, Int64, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}),
        D.ctx, bb, bt, Float64(δ), Int64(maxiter), cap, idx, val, nnz))
    to_sparse(size(D, 2), idx, val, nnz[])
end
function rmp(A::MatOrDict, b::AbstractVector, k::Int; kmax::Int = 0)
    D = dict(A)
    bb, bt = bvec(b)
    cap, idx, val, nnz = stepwise_out(A, kmax)
    GC.@preserve bb idx val check(D, ccall((:csmp_rmp_k, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Int64, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}),
        D.ctx, bb, bt, Int64(k), cap, idx, val, nnz))
    to_sparse(size(D, 2), idx, val, nnz[])
end
# isfast is accepted for signature parity (src/stepwise.jl:47): Val(false) computes the same backward scores the slow way
function foba(A::MatOrDict, b::AbstractVector, δ::Real; kmax::Int = 0, isfast::Val = Val(true))
    D = dict(A)
    bb, bt = bvec(b)
    cap, idx, val, nnz = stepwise_out(A, kmax)
    GC.@preserve bb idx val check(D, ccall((:csmp_foba, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Int64, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}),
        D.ctx, bb, bt, Float64(δ), cap, idx, val, nnz))
    to_sparse(size(D, 2), idx, val, nnz[])
end

# ---------------------------------------------------------------------------------- br = fbr, lace
# src/backward.jl:27-41,148-162,226-242
function backward_call(A::MatOrDict, b::AbstractVector, max_ε::Real, max_δ::Real, k::Int, lace::Bool)
    n, m = size(A)
    n ≥ m || throw("A needs to be overdetermined but is of size ($n, $m)")
    D = dict(A)
    bb, bt = bvec(b)
    idx, val, nnz = zeros(Int64, m + 1), zeros(Float64, m + 1), Ref{Int64}(0)
    GC.@preserve bb idx val check(D, ccall((:csmp_br, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Int64, Cint, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}),
        D.ctx, bb, bt, max_ε, max_δ, k, lace, idx, val, nnz))
    to_sparse(m, idx, val, nnz[])
end
br(A::MatOrDict, b::AbstractVector, max_ε::Real, max_δ::Real, k::Int) = backward_call(A, b, max_ε, max_δ, k, false)
br(A::MatOrDict, b::AbstractVector; max_residual::Real = Inf, max_increase::Real = Inf, sparsity::Int = 0) =
    br(A, b, max_residual, max_increase, sparsity)
const fbr = br
lace(A::MatOrDict, b::AbstractVector, ε::Real, δ::Real, k::Int) = backward_call(A, b, ε, δ, k, true)
lace(A::MatOrDict, b::AbstractVector; max_residual::Real = Inf, max_increase::Real = Inf, sparsity::Int = 0) =
    lace(A, b, max_residual, max_increase, sparsity)

# ---------------------------------------------------------------------------------- functors
# abstract type Update; (U::Update)(x) = update!(U, x)   (src/CompressedSensing.jl:22-23)
abstract type Update{T} end
(U::Update)(x) = update!(U, x)

# Every functor works on a context of its OWN that borrows the Dictionary's device memory (csmp_clone): like the
# reference's P objects, two functors on one A -- or a functor and a driver call -- share A and nothing else.
mutable struct DevicePursuit{T} <: Update{T}
    D::Dictionary{T}     # keeps the borrowed dictionary alive
    ctx::Ptr{Cvoid}      # the clone
    algo::Cint
    l::Int
    kcap::Int
    b::Any               # the signal (the reference's P.b): a two-stage functor restarts from a foreign x with it
end
function begin_solver(A::MatOrDict{T}, b, algo, kcap, l = 1) where {T}
    D = dict(A)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(D, ccall((:csmp_clone, libcsmp), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), D.ctx, ref))
    P = DevicePursuit{T}(D, ref[], algo, l, kcap, b)
    finalizer(p -> ccall((:csmp_destroy, libcsmp), Cint, (Ptr{Cvoid},), p.ctx), P)
    bb, bt = bvec(b)
    GC.@preserve bb pcheck(P, ccall((:csmp_solver_begin, libcsmp), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Int64, Ptr{Int64}, Ptr{Cdouble}, Int64),
        P.ctx, algo, bb, bt, kcap, C_NULL, C_NULL, 0))
    P
end
pcheck(P::DevicePursuit, rc::Integer) =
    rc == 0 || throw(unsafe_string(ccall((:csmp_last_error, libcsmp), Cstring, (Ptr{Cvoid},), P.ctx)))
MP(A, b; steps::Integer = 4096) = begin_solver(A, b, ALGO_MP, steps)                       # :19-24 (steps: length of the device's step log)
OMP(A, b, k::Integer = size(A, 1)) = begin_solver(A, b, ALGO_OMP, min(k, size(A, 1)))      # :54-60
GOMP(A, b, l::Int, k::Integer = size(A, 1)) = begin_solver(A, b, ALGO_GOMP, min(k, size(A, 1)), l)  # :108-114

# update!(P, x): one greedy step on the device, then x <- the device's current solution
function update!(P::DevicePursuit, x::SparseVector = spzeros(size(P.D, 2)), l::Int = P.l)
    (P.algo == ALGO_SP || P.algo == ALGO_OMPR) && return update_twostage!(P, x)
    pcheck(P, ccall((:csmp_solver_step, libcsmp), Cint, (Ptr{Cvoid}, Int64), P.ctx, l))
    idx, val, nnz = zeros(Int64, P.kcap), zeros(Float64, P.kcap), Ref{Int64}(0)
    GC.@preserve idx val pcheck(P, ccall((:csmp_solver_state, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ptr{Cdouble}, Ptr{Int64}, Ptr{Cint}),
        P.ctx, idx, val, nnz, C_NULL, C_NULL, C_NULL))
    y = to_sparse(size(P.D, 2), idx, val, nnz[])
    resize!(x.nzind, nnz[]); resize!(x.nzval, nnz[])
    copyto!(x.nzind, y.nzind); copyto!(x.nzval, y.nzval)
    return x
end

# ---- the two-stage functors: SP (src/twostage.jl:42-83) and OMPR (:110-180).  The device solver owns the x it produced last; the x
# handed to update! is checked against it: SP restarts from a foreign x (the reference recomputes the residual from whatever x it
# is given, :68), OMPR refuses one (its factorisation belongs to the x it built).
function SP(A::MatOrDict, b::AbstractVector, k::Integer)
    2k > length(b) && error("2k = $(2k) > $(length(b)) = length(b) is invalid for Subspace Pursuit")  # :55
    begin_solver(A, b, ALGO_SP, Int(k))
end
const SubspacePursuit = SP
OMPR(A::MatOrDict, b::AbstractVector, k::Int) = begin_solver(A, b, ALGO_OMPR, k)  # :124-132
mutable struct TwoStageX  # the x a two-stage functor returned last (per functor; keyed by the clone's handle)
    nzind::Vector{Int}
    nzval::Vector{Float64}
end
const LAST_X = Dict{Ptr{Cvoid},TwoStageX}()
function fetch_x!(P::DevicePursuit, x::SparseVector)
    cap = 2 * P.kcap
    idx, val, nnz = zeros(Int64, cap), zeros(Float64, cap), Ref{Int64}(0)
    GC.@preserve idx val pcheck(P, ccall((:csmp_solver_state, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ptr{Cdouble}, Ptr{Int64}, Ptr{Cint}),
        P.ctx, idx, val, nnz, C_NULL, C_NULL, C_NULL))
    y = to_sparse(size(P.D, 2), idx, val, nnz[])
    resize!(x.nzind, nnz[]); resize!(x.nzval, nnz[])
    copyto!(x.nzind, y.nzind); copyto!(x.nzval, y.nzval)
    LAST_X[P.ctx] = TwoStageX(copy(x.nzind), copy(x.nzval))
    return x
end
is_mine(P::DevicePursuit, x::SparseVector) = (l = get(LAST_X, P.ctx, TwoStageX(Int[], Float64[])); x.nzind == l.nzind && x.nzval == l.nzval)
# a foreign x restarts the SP solver from it (csmp_solver_begin with the x's entries, 0-based)
function load_x!(P::DevicePursuit, x::SparseVector)
    is_mine(P, x) && return
    bb, bt = bvec(P.b)
    i0, v0 = x.nzind .- 1, Vector{Float64}(x.nzval)
    GC.@preserve bb i0 v0 pcheck(P, ccall((:csmp_solver_begin, libcsmp), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Int64, Ptr{Int64}, Ptr{Cdouble}, Int64),
        P.ctx, P.algo, bb, bt, P.kcap, i0, v0, length(i0)))
    LAST_X[P.ctx] = TwoStageX(copy(x.nzind), copy(x.nzval))
end
# sp_acquisition!(P, x, k = P.k): :67-72
function sp_acquisition!(P::DevicePursuit, x::SparseVector, k::Int = P.kcap)
    P.algo == ALGO_SP || throw("sp_acquisition!: P is not an SP")
    load_x!(P, x)
    pcheck(P, ccall((:csmp_solver_acquire, libcsmp), Cint, (Ptr{Cvoid}, Int64), P.ctx, k))
    fetch_x!(P, x)
end
# oblivious_acquisition!(P, x, k): src/matchingpursuit.jl:207-216, for the functors that keep an updatable QR (OMPR, OMP, GOMP)
function oblivious_acquisition!(P::DevicePursuit, x::SparseVector, k::Int)
    P.algo in (ALGO_OMPR, ALGO_OMP, ALGO_GOMP) || throw("oblivious_acquisition!: P keeps no updatable QR")
    pcheck(P, ccall((:csmp_solver_acquire, libcsmp), Cint, (Ptr{Cvoid}, Int64), P.ctx, k))
    fetch_x!(P, x)
end
# update!(P::SP, x) (:75-83) and update!(P::OMPR, x) (:134-180, eta = 1): nnz(x) == k or the reference's String is thrown (:76, :135)
function update_twostage!(P::DevicePursuit, x::SparseVector)
    nnz(x) == P.kcap || throw("nnz(x) = $(nnz(x)) ≠ $(P.kcap) = k")
    if P.algo == ALGO_SP
        load_x!(P, x)
    else
        is_mine(P, x) || throw("update!(P::OMPR, x): x is not the vector this OMPR object's QR was built for")
    end
    pcheck(P, ccall((:csmp_solver_step, libcsmp), Cint, (Ptr{Cvoid}, Int64), P.ctx, 1))
    fetch_x!(P, x)
end

# ---------------------------------------------------------------------------------- many signals
# [omp(A, B[:, s], eps, k) for s in axes(B, 2)] on one GPU.  method = :exact: single-signal sweeps, three signals
# pipelined (csmp_omp_batch); :mfma: one bf16 screening GEMM per step + Float64 rescoring (csmp_omp_batch_mfma), with
# certificate = :rigorous (the library's default) | :statistical and gram = true | false (CSMP_OPT_BATCH_CERT /
# CSMP_OPT_BATCH_GRAM).  A keyword that is not given leaves the Dictionary's option (set_option!) alone; one that is given holds
# for this call only.
# Returns (idx k x nsig 0-based, -1 padded; val; nnz) as the C ABI does -- the layout csmp_pack_results packs.
function omp_batch_raw(A::MatOrDict{T}, B::StridedMatrix, ε::Real, k::Int; method::Symbol = :exact,
                       certificate::Union{Symbol,Nothing} = nothing, gram::Union{Bool,Nothing} = nothing) where {T}
    D = dict(A)
    saved = Pair{Symbol,Int64}[]
    if method === :mfma && certificate !== nothing
        push!(saved, :batch_cert => get_option(D, :batch_cert))
        set_option!(D, :batch_cert, certificate === :rigorous ? 1 : 0)
    end
    if method === :mfma && gram !== nothing
        push!(saved, :batch_gram => get_option(D, :batch_gram))
        set_option!(D, :batch_gram, gram ? 1 : 0)
    end
    try
        return omp_batch_call(D, B, ε, k, method)
    finally
        for (key, v) in saved
            set_option!(D, key, v)
        end
    end
end
function omp_batch_call(D::Dictionary, B::StridedMatrix, ε::Real, k::Int, method::Symbol)
    BB = eltype(B) <: Union{Float32,Float64} ? B : convert(Matrix{Float64}, B)
    nsig = size(BB, 2)
    idx, val, nnz = fill(Int64(-1), k, nsig), zeros(Float64, k, nsig), zeros(Int64, nsig)
    if method === :mfma
        GC.@preserve BB idx val nnz check(D, ccall((:csmp_omp_batch_mfma, libcsmp), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Int64, Cint, Int64, Cdouble, Ptr{Int64}, Ptr{Cdouble}, Ptr{Int64}, Cint),
            D.ctx, BB, dtype_code(eltype(BB)), stride(BB, 2), nsig, CSMP_HOST, k, ε, idx, val, nnz, CSMP_HOST))
    else
        GC.@preserve BB idx val nnz check(D, ccall((:csmp_omp_batch, libcsmp), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Int64, Cint, Int64, Cdouble, Ptr{Int64}, Ptr{Cdouble}, Ptr{Int64}, Cint),
            D.ctx, BB, dtype_code(eltype(BB)), stride(BB, 2), nsig, CSMP_HOST, k, ε, idx, val, nnz, CSMP_HOST))
    end
    idx, val, nnz
end
omp_batch(A::MatOrDict, B::StridedMatrix, ε::Real, k::Int; kw...) = begin
    idx, val, nnz = omp_batch_raw(A, B, ε, k; kw...)
    [to_sparse(size(A, 2), idx[:, s], val[:, s], nnz[s]) for s in 1:size(B, 2)]
end

# [gomp(A, B[:, s], l, eps, k) for s in axes(B, 2)]: two solves in flight on two streams (csmp_gomp_batch)
function gomp_batch(A::MatOrDict{T}, B::StridedMatrix, l::Int, ε::Real, k::Int) where {T}
    D = dict(A)
    BB = eltype(B) <: Union{Float32,Float64} ? B : convert(Matrix{Float64}, B)
    nsig = size(BB, 2)
    idx, val, nnz = fill(Int64(-1), k, nsig), zeros(Float64, k, nsig), zeros(Int64, nsig)
    GC.@preserve BB idx val nnz check(D, ccall((:csmp_gomp_batch, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Int64, Cint, Int64, Int64, Cdouble, Ptr{Int64}, Ptr{Cdouble}, Ptr{Int64}, Cint),
        D.ctx, BB, dtype_code(eltype(BB)), stride(BB, 2), nsig, CSMP_HOST, l, k, ε, idx, val, nnz, CSMP_HOST))
    [to_sparse(size(A, 2), idx[:, s], val[:, s], nnz[s]) for s in 1:nsig]
end

# [sp(A, B[:, s], k, δ; maxiter) for s in axes(B, 2)]: up to four solves in flight (csmp_sp_batch; CSMP_OPT_SOLVES_IN_FLIGHT)
function sp_batch(A::MatOrDict{T}, B::StridedMatrix, k::Int, δ::Real = 1e-12; maxiter = 16k) where {T}
    2k > size(B, 1) && error("2k = $(2k) > $(size(B, 1)) = length(b) is invalid for Subspace Pursuit")
    D = dict(A)
    BB = eltype(B) <: Union{Float32,Float64} ? B : convert(Matrix{Float64}, B)
    nsig = size(BB, 2)
    idx, val, nnz, its = fill(Int64(-1), k, nsig), zeros(Float64, k, nsig), zeros(Int64, nsig), zeros(Int64, nsig)
    GC.@preserve BB idx val nnz its check(D, ccall((:csmp_sp_batch, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Int64, Int64, Cdouble, Int64, Ptr{Int64}, Ptr{Cdouble}, Ptr{Int64}, Ptr{Int64}),
        D.ctx, BB, dtype_code(eltype(BB)), stride(BB, 2), nsig, k, δ, maxiter, idx, val, nnz, its))
    [to_sparse(size(A, 2), idx[:, s], val[:, s], nnz[s]) for s in 1:nsig]
end

# Signals sharded over ranks (SURVEY section 8e), ONE PROCESS PER GPU, the collective INSIDE the library (csmp_omp_sharded: one
# ncclAllGather over xGMI, device memory to device memory) -- no collective package on the Julia side:
#     id = rank == 0 ? comm_id() : nothing          # 128 bytes; hand them to every rank by whatever started the ranks
#     comm_init!(D, id, rank, world)                # (a file, a socket, Distributed.remotecall_fetch, ...): collective
#     xs = omp_sharded(D, B[:, lo+1:hi], nsig, ε, k; method = :mfma)   # every rank: its block in, ALL nsig results out
const CSMP_COMM_ID_BYTES = 128
function comm_id()
    id = zeros(UInt8, CSMP_COMM_ID_BYTES)
    rc = ccall((:csmp_comm_id, libcsmp), Cint, (Ptr{Cvoid},), id)
    rc == 0 || throw(unsafe_string(ccall((:csmp_last_error, libcsmp), Cstring, (Ptr{Cvoid},), C_NULL)))
    id
end
comm_init!(D::Dictionary, id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(D, ccall((:csmp_comm_init, libcsmp), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), D.ctx, id, rank, world))
comm_free!(D::Dictionary) = check(D, ccall((:csmp_comm_free, libcsmp), Cint, (Ptr{Cvoid},), D.ctx))
# this rank's contiguous block of nsig signals, 1-based inclusive (csmp_shard_range)
function shard_range(nsig::Integer, rank::Integer, world::Integer)
    lo, hi = Ref{Int64}(0), Ref{Int64}(0)
    ccall((:csmp_shard_range, libcsmp), Cint, (Int64, Cint, Cint, Ref{Int64}, Ref{Int64}), nsig, rank, world, lo, hi)
    (lo[] + 1):hi[]
end
function omp_sharded(D::Dictionary, Bblock::StridedMatrix, nsig::Integer, ε::Real, k::Int; method::Symbol = :exact)
    BB = eltype(Bblock) <: Union{Float32,Float64} ? Bblock : convert(Matrix{Float64}, Bblock)
    idx, val, nnz = fill(Int64(-1), k, nsig), zeros(Float64, k, nsig), zeros(Int64, nsig)
    GC.@preserve BB idx val nnz check(D, ccall((:csmp_omp_sharded, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Int64, Cint, Int64, Cdouble, Cint, Ptr{Int64}, Ptr{Cdouble}, Ptr{Int64}, Cint),
        D.ctx, BB, dtype_code(eltype(BB)), max(stride(BB, 2), size(D, 1)), nsig, CSMP_HOST, k, ε, method === :mfma ? 1 : 0, idx, val, nnz, CSMP_HOST))
    [to_sparse(size(D, 2), idx[:, s], val[:, s], nnz[s]) for s in 1:nsig]
end

# The same sharding under a collective the HOST brings (any single-signal driver, or a site that already runs MPI):
# `allgather(v::Vector{Float64}) -> Vector{Float64}`; every rank passes a block of the same length (blocks are padded to the
# longest one: ceil(nsig / world) signals).
function omp_sharded(A::MatOrDict, B::StridedMatrix, ε::Real, k::Int, rank::Int, world::Int, allgather;
                     method::Symbol = :exact, kw...)
    nsig = size(B, 2)
    lo, hi = Ref{Int64}(0), Ref{Int64}(0)
    ccall((:csmp_shard_range, libcsmp), Cint, (Int64, Cint, Cint, Ref{Int64}, Ref{Int64}), nsig, rank, world, lo, hi)
    n = hi[] - lo[]
    idx, val, nnz = omp_batch_raw(A, B[:, lo[]+1:hi[]], ε, k; method = method, kw...)
    maxn = cld(nsig, world)
    w = 2k + 1
    packed = zeros(Float64, w * maxn)                     # row-major rows of 2k+1: [idx | val | nnz] per signal
    ccall((:csmp_pack_results, libcsmp), Cint, (Ptr{Int64}, Ptr{Cdouble}, Ptr{Int64}, Int64, Int64, Ptr{Cdouble}),
          idx, val, nnz, k, n, packed)                    # (idx/val are k x n column-major = n rows of k, as the ABI wants)
    all = allgather(packed)
    out = Vector{SparseVector{Float64,Int}}(undef, nsig)
    for r in 0:world-1
        ccall((:csmp_shard_range, libcsmp), Cint, (Int64, Cint, Cint, Ref{Int64}, Ref{Int64}), nsig, r, world, lo, hi)
        m = hi[] - lo[]
        ri, rv, rn = zeros(Int64, k, m), zeros(Float64, k, m), zeros(Int64, m)
        block = all[r*w*maxn+1:r*w*maxn+w*m]
        ccall((:csmp_unpack_results, libcsmp), Cint, (Ptr{Cdouble}, Int64, Int64, Ptr{Int64}, Ptr{Cdouble}, Ptr{Int64}),
              block, k, m, ri, rv, rn)
        for s in 1:m
            out[lo[]+s] = to_sparse(size(A, 2), ri[:, s], rv[:, s], rn[s])
        end
    end
    out
end

# argmaxinner!(P) / argmaxinner!(P, k): src/matchingpursuit.jl:181-193
function argmaxinner(D::Dictionary, r::AbstractVector, k::Int = 1)
    rr = convert(Vector{Float64}, r)
    ti, tv = zeros(Int64, k), zeros(Float64, k)
    GC.@preserve rr ti tv check(D, ccall((:csmp_sweep, libcsmp), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Int64}, Ptr{Cdouble}),
        D.ctx, rr, C_NULL, k, ti, tv))
    k == 1 ? ti[1] + 1 : ti .+ 1
end

end # module
