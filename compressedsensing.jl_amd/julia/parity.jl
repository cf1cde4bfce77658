# parity.jl -- diff of the MI355X path (CompressedSensingAMD.jl over libcsmp.so) against the REAL CompressedSensing.jl
# on the committed golden inputs and on fresh planted problems at the reference's own test shapes.
#
# Never executed in the build image (no Julia there): this is the script a maintainer with Julia and an MI355X
# runs to pin parity end to end.
#
#     julia --project=/path/to/CompressedSensing.jl compressedsensing.jl_amd/julia/parity.jl [tests/golden/golden_small.npz]
#
# Pass criterion (BASELINE.json north_star): identical support sets, coefficients within 1e-6 relative.
using NPZ, SparseArrays, LinearAlgebra, Random
import CompressedSensing
const REF = CompressedSensing
include(joinpath(@__DIR__, "CompressedSensingAMD.jl"))
const AMD = CompressedSensingAMD

same(x, y) = x.nzind == y.nzind && isapprox(x.nzval, y.nzval; rtol = 1e-6, atol = 1e-6 * maximum(abs, y.nzval; init = 0.0))
bad = 0
function report(name, ok)
    global bad
    println(ok ? "ok       " : "MISMATCH ", name)
    bad += !ok
end

# 1. the committed golden inputs, both implementations side by side
golden = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "..", "tests", "golden", "golden_small.npz")
z = npzread(golden)
for name in String.(z["names"])
    A = z[name * ".A"]; b = z[name * ".b"]; p = z[name * ".params"]; algo = String(z[name * ".algo"])
    D = AMD.Dictionary(A)
    pair = if algo == "omp"
        (REF.omp(A, b, p[2], Int(p[1])), AMD.omp(D, b, p[2], Int(p[1])))
    elseif algo == "mp"
        (REF.mp(A, b, Int(p[1])), AMD.mp(D, b, Int(p[1])))
    elseif algo == "gomp"
        (REF.gomp(A, b, Int(p[1]), p[3], Int(p[2])), AMD.gomp(D, b, Int(p[1]), p[3], Int(p[2])))
    elseif algo == "sp"
        (REF.sp(A, b, Int(p[1]), p[2]), AMD.sp(D, b, Int(p[1]), p[2]))
    elseif algo == "fr"
        (REF.fr(A, b, p[2], p[3], Int(p[1])), AMD.fr(D, b, p[2], p[3], Int(p[1])))
    elseif algo == "srr"
        (REF.srr(A, b, Int(p[1]), p[2]; initialization = Int(p[3]), l = Int(p[4])),
         AMD.srr(D, b, Int(p[1]), p[2]; initialization = Int(p[3]), l = Int(p[4])))
    elseif algo == "rmp_k"
        (REF.rmp(A, b, Int(p[1])), AMD.rmp(D, b, Int(p[1])))
    elseif algo == "rmp_delta"
        (REF.rmp(A, b, p[1], Int(p[2])), AMD.rmp(D, b, p[1], Int(p[2])))
    elseif algo == "foba"
        (REF.foba(A, b, p[1]), AMD.foba(D, b, p[1]))
    elseif algo == "br"
        (REF.br(A, b, p[1], p[2], Int(p[3])), AMD.br(D, b, p[1], p[2], Int(p[3])))
    else
        (REF.lace(A, b, p[1], p[2], Int(p[3])), AMD.lace(D, b, p[1], p[2], Int(p[3])))
    end
    report(name, all(isfinite, pair[1].nzval) ? same(pair[2], pair[1]) : pair[2].nzind == pair[1].nzind)
end

# 2. fresh planted problems at the reference's test shapes (test/matchingpursuit.jl:10-45) and at BASELINE configs[0]
Random.seed!(20240607)
for (n, m, k, T) in [(32, 48, 3, Float64), (32, 48, 3, Float32), (256, 1024, 32, Float64)], trial in 1:5
    A, x0, b = REF.sparse_data(n = n, m = m, k = k)
    A = convert(Matrix{T}, A)
    y = REF.perturb(convert(Vector{Float64}, b), 5e-3)
    D = AMD.Dictionary(A)
    report("omp $(n)x$(m) k=$k $T #$trial", same(AMD.omp(D, y, k), REF.omp(A, y, k)))
    report("gomp $(n)x$(m) k=$k l=2 $T #$trial", same(AMD.gomp(D, y, 2, k), REF.gomp(A, y, 2, k)))
    2k <= n && report("sp $(n)x$(m) k=$k $T #$trial", same(AMD.sp(D, y, k), REF.sp(A, y, k)))
end
println(bad, " mismatching case(s)")
exit(bad == 0 ? 0 : 1)
