# make_golden_reference.jl -- pins the oracle against the REAL CompressedSensing.jl.
#
# The build image has no Julia, so tests/golden/golden_small.npz was produced by the C oracle (make_golden.py) and
# the project's parity claim is capped at "partial" until someone runs this script.  It loads the real package,
# replays the 44 committed golden INPUTS (A, b, parameters) through the reference's own drivers and writes the
# reference-produced outputs in the same npz layout:
#
#     julia --project=/path/to/CompressedSensing.jl tests/golden/make_golden_reference.jl \
#           tests/golden/golden_small.npz tests/golden/golden_reference.npz
#     python tests/golden/compare_golden.py tests/golden/golden_small.npz tests/golden/golden_reference.npz
#
# needs: the CompressedSensing.jl project (with UpdatableQRFactorizations.jl v1.0.0, Manifest.toml:446-450) and NPZ.jl.
# Layout per case `name`: name.A, name.b, name.algo (string), name.params, name.idx (0-BASED sorted), name.val,
# name.order (selection order for omp, gomp and fr -- replayed through the reference's step functions; empty otherwise).  Parameter meaning per algorithm is
# the one make_golden.py documents; indices are written 0-based like the C ABI returns them.
using NPZ, SparseArrays, LinearAlgebra
using CompressedSensing
const CS = CompressedSensing

src, dst = ARGS[1], ARGS[2]
z = npzread(src)
names = String.(z["names"])
out = Dict{String,Any}("names" => z["names"])

# Selection orders, recorded by replaying the reference's OWN step functions beside its drivers (the drivers return only x).
# omp: update!(P::OMP, x) adds at most one atom per call (src/matchingpursuit.jl:62-70): the index that appears.
function omp_order(A, b, k)
    P = CS.OMP(A, b, k)
    x = spzeros(size(A, 2))
    seen = Int[]
    for _ in 1:k
        before = copy(x.nzind)
        CS.update!(P, x)
        new = setdiff(x.nzind, before)
        isempty(new) && break
        append!(seen, new)
    end
    seen
end
# gomp: the driver's own loop (src/matchingpursuit.jl:126-139) around update!(P::GOMP, x, l) (:116-123).  A step inserts the
# atoms of argmaxinner!(P, l) in partialsortperm's order (descending |<a, r>|, ties by ascending index), skipping the ones the
# support already holds (src/util.jl:118-134): residual! + argmaxinner! are run once more in front of every update! to READ that
# list (they only overwrite P.r and P.Ar, which update! recomputes first thing), then the real update! does the step.
function gomp_order(A, b, l, ε, k)
    P = CS.GOMP(A, b, l)
    x = spzeros(size(A, 2))
    seen = Int[]
    function step!(ll)
        nnz(x) < size(A, 1) || return
        CS.residual!(P, x)
        top = copy(CS.argmaxinner!(P, ll))
        before = copy(x.nzind)
        CS.update!(P, x, ll)
        append!(seen, [j for j in top if !(j in before)])
        @assert sort(setdiff(x.nzind, before)) == sort([j for j in top if !(j in before)])
    end
    for _ in 1:(k ÷ l)
        step!(l)
        norm(CS.residual!(P, x)) ≥ ε || break            # :132
    end
    rem = mod(k, l)
    rem > 0 && step!(rem)                                 # :134-137 (runs even after an eps-break)
    seen
end
# fr: the driver's own loop (src/forward.jl:44-50) around forward_step! (:56-72): one atom per successful step
function fr_order(A, b, max_ε, min_δ, k)
    P = CS.FR(A, b)
    x = spzeros(size(A, 2))
    seen = Int[]
    for _ in 1:k
        before = copy(x.nzind)
        CS.forward_step!(P, x, max_ε, min_δ) || break
        append!(seen, setdiff(x.nzind, before))
    end
    seen
end

for name in names
    A = z[name * ".A"]; b = z[name * ".b"]; p = z[name * ".params"]
    algo = String(z[name * ".algo"])
    order = Int[]
    x = if algo == "omp"
        order = omp_order(A, b, Int(p[1]))
        CS.omp(A, b, p[2], Int(p[1]))                    # src/matchingpursuit.jl:73-82
    elseif algo == "mp"
        CS.mp(A, b, Int(p[1]))                           # :34-40
    elseif algo == "gomp"
        order = gomp_order(A, b, Int(p[1]), p[3], Int(p[2]))
        CS.gomp(A, b, Int(p[1]), p[3], Int(p[2]))        # :126-139   params = [l, k, eps]
    elseif algo == "sp"
        CS.sp(A, b, Int(p[1]), p[2])                     # src/twostage.jl:87-101   params = [k, delta, iterations]
    elseif algo == "fr"
        order = fr_order(A, b, p[2], p[3], Int(p[1]))
        CS.fr(A, b, p[2], p[3], Int(p[1]))               # src/forward.jl:44-50     params = [k, max_eps, min_delta]
    elseif algo == "srr"
        CS.srr(A, b, Int(p[1]), p[2]; initialization = Int(p[3]), l = Int(p[4]))   # src/twostage.jl:3-33
    elseif algo == "rmp_k"
        CS.rmp(A, b, Int(p[1]))                          # src/stepwise.jl:32-43
    elseif algo == "rmp_delta"
        CS.rmp(A, b, p[1], Int(p[2]))                    # :5-26
    elseif algo == "foba"
        CS.foba(A, b, p[1])                              # :47-56
    elseif algo == "ompr"
        CS.ompr(A, b, Int(p[1]), p[2])                   # src/twostage.jl:184-202  params = [k, delta, iterations]
    elseif algo == "sp_steps"                            # the functor, call by call: params = [k, steps]
        P = CS.SP(A, b, Int(p[1])); xs = spzeros(size(A, 2))
        CS.sp_acquisition!(P, xs, P.k)                   # src/twostage.jl:67-72
        for _ in 1:Int(p[2]); CS.update!(P, xs); end     # :75-83
        xs
    elseif algo == "ompr_steps"
        P = CS.OMPR(A, b, Int(p[1])); xs = spzeros(size(A, 2))
        CS.oblivious_acquisition!(P, xs, P.k)            # src/matchingpursuit.jl:207-216
        for _ in 1:Int(p[2]); CS.update!(P, xs); end     # src/twostage.jl:134-180
        xs
    elseif algo == "br"
        CS.br(A, b, p[1], p[2], Int(p[3]))               # src/backward.jl:27-35
    elseif algo == "lace"
        CS.lace(A, b, p[1], p[2], Int(p[3]))             # :233-270
    else
        error("unknown algo $algo")
    end
    out[name * ".A"] = A; out[name * ".b"] = b; out[name * ".params"] = p
    out[name * ".algo"] = z[name * ".algo"]
    out[name * ".idx"] = Int64.(x.nzind .- 1)
    out[name * ".val"] = Float64.(x.nzval)
    out[name * ".order"] = Int64.(order .- 1)
end
npzwrite(dst, out)
println("wrote $(length(names)) reference-produced cases to $dst")
