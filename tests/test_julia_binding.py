"""The Julia wrapper (compressedsensing.jl_amd/julia/CompressedSensingAMD.jl) is the binding the reference side would use and
it cannot execute here (no Julia in the image).  This CPU test checks every `ccall` in it STATICALLY against the prototype in
include/csmp.h: the symbol is a literal that the header declares, the return type, the arity and every argument's C type agree
(Int64 <-> int64_t, Cint <-> int, Cdouble <-> double, Ptr/Ref{T} <-> T*, Cstring <-> const char*), and as many values are
passed as types are declared.  VERDICT round 3, "Next round" item 6."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "compressedsensing.jl_amd", "julia", "CompressedSensingAMD.jl")
HDR = os.path.join(ROOT, "include", "csmp.h")


def c_prototypes():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(csmp_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        protos[name] = (c_class(ret, is_param=False), [c_class(p) for p in plist])
    return protos


def c_class(decl, is_param=True):
    """'const int64_t *idx' -> 'ptr:i64'; 'int b_dtype' -> 'i32'; 'csmp_ctx **out' -> 'ptr:ptr'."""
    d = decl.replace("const", " ").strip()
    stars = d.count("*")
    d = d.replace("*", " ")
    toks = d.split()
    if is_param and len(toks) > 1 and toks[-1] not in ("int", "double", "float", "char", "void", "int64_t", "csmp_ctx"):
        toks = toks[:-1]  # drop the parameter name
    base = {"int": "i32", "int64_t": "i64", "double": "f64", "float": "f32", "char": "char", "void": "void", "csmp_ctx": "void"}[" ".join(toks)]
    if stars == 0:
        return base
    if stars == 1:
        return "ptr:" + base
    return "ptr:ptr"


JL_CLASS = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Clonglong": "i64", "Cdouble": "f64", "Float64": "f64", "Cfloat": "f32",
            "Cstring": "ptr:char", "Cvoid": "void"}


def jl_class(t):
    t = t.strip()
    m = re.fullmatch(r"(?:Ptr|Ref)\{(.+)\}", t)
    if m:
        inner = jl_class(m.group(1))
        return "ptr:ptr" if inner.startswith("ptr") else "ptr:" + inner
    return JL_CLASS[t]


def split_top(s):
    """split on top-level commas"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def balanced(src, i):
    """src[i] == '(' -> index just past its matching ')'"""
    depth = 0
    for j in range(i, len(src)):
        if src[j] in "([{":
            depth += 1
        elif src[j] in ")]}":
            depth -= 1
            if depth == 0:
                return j + 1
    raise AssertionError("unbalanced")


def julia_ccalls():
    src = open(JL).read()
    src = re.sub(r"#[^\n]*", "", src)  # (no '#' inside strings on ccall lines in this file)
    calls = []
    for m in re.finditer(r"\bccall\s*\(", src):
        end = balanced(src, m.end() - 1)
        parts = split_top(src[m.end():end - 1])
        line = src.count("\n", 0, m.start()) + 1
        calls.append((line, parts))
    return calls


def compatible(jl, c):
    if jl == c:
        return True
    # a typed Julia pointer may be passed where C takes void*, and Ptr{Cvoid} where C takes any object pointer (opaque handles, buffers
    # of either element type); char buffers may be Ptr{UInt8}
    if jl.startswith("ptr") and c.startswith("ptr") and ("void" in (jl[4:], c[4:])):
        return True
    return False


def test_every_ccall_matches_its_prototype():
    protos = c_prototypes()
    assert len(protos) >= 40
    calls = julia_ccalls()
    assert len(calls) >= 30
    seen = set()
    for line, parts in calls:
        head = parts[0]
        m = re.fullmatch(r"\(\s*:(csmp_[a-z0-9_]+)\s*,\s*libcsmp\s*\)", head)
        assert m, f"line {line}: ccall's function must be a LITERAL (:symbol, libcsmp) tuple, got {head!r}"
        name = m.group(1)
        assert name in protos, f"line {line}: {name} is not declared in include/csmp.h"
        seen.add(name)
        ret_c, params_c = protos[name]
        assert compatible(jl_class(parts[1]), ret_c), f"line {line}: {name} returns {ret_c}, the ccall says {parts[1]}"
        tt = parts[2]
        assert tt.startswith("(") and tt.endswith(")"), f"line {line}: {name}: argument types must be a literal tuple"
        assert "..." not in tt, f"line {line}: {name}: splatted argument types are not valid in a ccall"
        types = split_top(tt[1:-1])
        assert len(types) == len(params_c), f"line {line}: {name} takes {len(params_c)} arguments, the ccall declares {len(types)}"
        for pos, (tj, tc) in enumerate(zip(types, params_c)):
            assert compatible(jl_class(tj), tc), f"line {line}: {name} argument {pos}: C {tc} vs Julia {tj}"
        nvals = len(parts) - 3
        assert nvals == len(types), f"line {line}: {name}: {len(types)} types declared, {nvals} values passed"
        assert not any("..." in v for v in parts[3:]), f"line {line}: {name}: splatted values hide the arity"
    # the wrapper binds the drivers of the hot path and their batch / sharding forms
    for need in ("csmp_create", "csmp_destroy", "csmp_set_dictionary", "csmp_mp", "csmp_omp", "csmp_gomp", "csmp_sp", "csmp_omp_batch",
                 "csmp_omp_batch_mfma", "csmp_gomp_batch", "csmp_sp_batch", "csmp_solver_begin", "csmp_solver_step", "csmp_solver_state",
                 "csmp_set_option", "csmp_get_option", "csmp_last_error", "csmp_sweep", "csmp_pack_results", "csmp_unpack_results"):
        assert need in seen, need


def test_option_table_matches_the_header():
    src = open(JL).read()
    hdr = open(HDR).read()
    keys = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define CSMP_OPT_([A-Z_]+)\s+(\d+)", hdr)}
    m = re.search(r"const OPTIONS = Dict\((.*?)\)\n", src, flags=re.S)
    jl = {k: int(v) for k, v in re.findall(r":([a-z_0-9]+)\s*=>\s*(\d+)", m.group(1))}
    assert jl == keys
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (CSMP_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", hdr)}
    for name, val in re.findall(r"const (CSMP_[A-Z0-9_]+)\s*=\s*(?:Cint\()?(-?\d+)\)?", src):
        assert consts.get(name) == int(val), name


def test_replay_script_covers_every_golden_case():
    """tests/golden/make_golden_reference.jl replays the committed golden INPUTS through the real package wherever Julia exists (never
    here).  Statically: every `algo` string of golden_small.npz has a branch in the script, and the three drivers whose golden
    vectors carry a selection order (omp, gomp, fr) record one through the reference's own step functions -- so that
    compare_golden.py checks the step order of update!(::OMP) (src/matchingpursuit.jl:62-70), update!(::GOMP) (:116-123) and
    forward_step! (src/forward.jl:56-72), not only the final supports."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"), allow_pickle=False)
    names = [str(n) for n in z["names"]]
    algos = {str(z[n + ".algo"]) for n in names}
    src = open(os.path.join(ROOT, "tests", "golden", "make_golden_reference.jl")).read()
    branches = set(re.findall(r'algo == "([a-z_]+)"', src))
    assert algos <= branches, sorted(algos - branches)
    with_order = {str(z[n + ".algo"]) for n in names if len(z[n + ".order"])}
    assert with_order == {"omp", "gomp", "fr"}, with_order
    for algo, fn in (("omp", "omp_order"), ("gomp", "gomp_order"), ("fr", "fr_order")):
        m = re.search(r'algo == "%s"\s*\n\s*order = %s\(' % (algo, fn), src)
        assert m, f"the {algo} branch does not record its selection order with {fn}"
        assert re.search(r"function %s\(" % fn, src), fn
    # the replay goes through the reference's step functions, not a restatement of them
    for call in ("CS.update!(P, x)", "CS.update!(P, x, ll)", "CS.forward_step!(P, x, max_ε, min_δ)", "CS.argmaxinner!(P, ll)"):
        assert call in src, call
    cmp_src = open(os.path.join(ROOT, "tests", "golden", "compare_golden.py")).read()
    assert 'np.array_equal(oa, ob)' in cmp_src  # orders are compared wherever both files carry one
    assert len(names) == 44
