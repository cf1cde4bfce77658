"""CPU tests of the boundary: libcsmp.so loads, exports every symbol include/csmp.h declares, and
refuses to run (loudly, no CPU fallback) when no GPU is visible."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="csmp.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(csmp_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(cs):
    assert os.path.exists(cs.LIB_PATH), "build libcsmp.so first: python -c 'import __graft_entry__ as g; g.build()'"
    L = ctypes.CDLL(cs.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms + header_symbols("csmp_internal.h"):
        assert hasattr(L, s), f"{s} declared in include/*.h but not exported"
    L.csmp_version.restype = ctypes.c_int
    assert L.csmp_version() >= 100


def test_binding_table_matches_header(cs):
    from csmp_pkg import load
    lib = load()._lib
    assert sorted(lib.SIGNATURES) == header_symbols()
    assert sorted(lib.INTERNAL_SIGNATURES) == header_symbols("csmp_internal.h")
    # the measurement hooks are not part of the boundary: the public header and the Julia wrapper name none of them
    jl = open(os.path.join(ROOT, "compressedsensing.jl_amd", "julia", "CompressedSensingAMD.jl")).read()
    for name in lib.INTERNAL_SIGNATURES:
        assert name not in header_symbols() and (":" + name) not in jl, name


def test_product_does_not_link_the_oracle(cs):
    import subprocess
    out = subprocess.run(["ldd", cs.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    csrc = os.path.join(ROOT, "compressedsensing.jl_amd", "csrc")
    for d, _, files in os.walk(csrc):
        for f in files:
            if f.endswith((".hip", ".hpp")):
                src = open(os.path.join(d, f)).read()  # (comments may NAME the oracle; nothing may include, load or call it)
                assert not re.search(r'#include[^\n]*oracle|csmp_oracle|cso_[a-z]+\s*\(', src), f
                if f != "rccl.hpp":  # the ONE place that binds anything at run time: RCCL, by name (checked below)
                    assert not re.search(r'dlopen|dlsym', src), f
    rccl = open(os.path.join(csrc, "host", "rccl.hpp")).read()
    assert re.findall(r'for \(const char\* name : \{([^}]*)\}\)', rccl) == ['"librccl.so.1", "librccl.so"']
    assert set(re.findall(r'sym\("([A-Za-z]+)"\)', rccl)) == {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGetErrorString", "ncclAllGather"}
    assert "rccl" not in out  # not linked either: bound lazily by csmp_comm_id / csmp_comm_init only


def test_no_cpu_fallback_without_gpu(cs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cs.CsmpError) as e:
        cs.Context(0)
    assert "no CPU fallback" in str(e.value)
    A, x, b = cs.sparse_data(16, 24, 2, rng=0)
    with pytest.raises(cs.CsmpError):
        cs.omp(A, b, 2)


def test_product_reads_no_environment_variable():
    """behavioural choices are arguments or csmp_set_option keys; the library has no
    getenv, no experiments build (CSMP_EXPERIMENTS) and no tuning knobs left in the product sources."""
    csrc = os.path.join(ROOT, "compressedsensing.jl_amd", "csrc")
    hits = []
    for d, _, files in os.walk(csrc):
        for name in sorted(files):
            if name.endswith((".hip", ".hpp")) or name == "Makefile":
                for ln, line in enumerate(open(os.path.join(d, name)), 1):
                    if re.search(r"\bgetenv\s*\(|CSMP_EXPERIMENTS|tune_env", line):
                        hits.append((name, ln, line.strip()))
    assert hits == [], hits


def test_option_keys_match_the_header(cs):
    from csmp_pkg import load
    lib = load()._lib
    src = open(os.path.join(ROOT, "include", "csmp.h")).read()
    keys = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define CSMP_OPT_([A-Z_]+)\s+(\d+)", src)}
    assert keys == lib.OPTIONS and len(keys) == 7


def build_c_example(cs, out):
    """tests/c_abi_example.c -> an executable linked against the in-tree libcsmp.so (strict C99: the header is C, not C++)"""
    import subprocess
    libdir = os.path.dirname(cs.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-o", out,
           os.path.join(ROOT, "tests", "c_abi_example.c"), "-L", libdir, "-lcsmp", "-lm", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_header_is_plain_c_and_links(cs, tmp_path):
    """include/csmp.h compiles as strict C99 and a C program that uses the drivers, the batch form and the dictionary files links
    against libcsmp.so with nothing else (no C++ runtime named on the command line, no torch): the boundary is a C ABI."""
    exe = build_c_example(cs, str(tmp_path / "c_abi_example"))
    assert os.path.exists(exe)
