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
# name.order (selection order where the driver defines one; empty otherwise).  Parameter meaning per algorithm is
# the one make_golden.py documents; indices are written 0-based like the C ABI returns them.
using NPZ, SparseArrays, LinearAlgebra
using CompressedSensing
const CS = CompressedSensing

src, dst = ARGS[1], ARGS[2]
z = npzread(src)
names = String.(z["names"])
out = Dict{String,Any}("names" => z["names"])

# selection order of omp / gomp / fr: replay the functor step by step and record which indices appear
function order_of(P, k; l = 1)
    x = spzeros(size(P.A, 2))
    seen = Int[]
    for _ in 1:k
        before = copy(x.nzind)
        l == 1 ? CS.update!(P, x) : CS.update!(P, x, l)
        new = setdiff(x.nzind, before)
        isempty(new) && break
        append!(seen, new)          # (gomp: the l new atoms of a step, ascending -- the order the C ABI reports)
    end
    seen
end

for name in names
    A = z[name * ".A"]; b = z[name * ".b"]; p = z[name * ".params"]
    algo = String(z[name * ".algo"])
    order = Int[]
    x = if algo == "omp"
        order = order_of(CS.OMP(A, b, Int(p[1])), Int(p[1]))
        CS.omp(A, b, p[2], Int(p[1]))                    # src/matchingpursuit.jl:73-82
    elseif algo == "mp"
        CS.mp(A, b, Int(p[1]))                           # :34-40
    elseif algo == "gomp"
        CS.gomp(A, b, Int(p[1]), p[3], Int(p[2]))        # :126-139   params = [l, k, eps]
    elseif algo == "sp"
        CS.sp(A, b, Int(p[1]), p[2])                     # src/twostage.jl:87-101   params = [k, delta, iterations]
    elseif algo == "fr"
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
