"""Mechanical guard for the Julia plug-in (gaussianmarkovrandomfields.jl_amd/julia/GMRFX.jl), which this image cannot parse
or run (no Julia): every `ccall((:gmrfx_*, LIB), ret, (types...), args...)` in the shim is checked against the prototype of the
same name in include/gmrfx.h -- name exists, arity, the C type of every argument and of the return value -- and the `Opts`
struct against `gmrfx_opts` field by field (order, type, total size).

The pattern being guarded is how the reference attaches foreign solvers: `src/workspace/cliquetrees_backend.jl:132-150` and
`ext/GaussianMarkovRandomFieldsPardiso.jl:10-80`; a wrong argument tuple there is a silent stack-corruption bug at run time."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd", "julia", "GMRFX.jl")
HDR = os.path.join(ROOT, "include", "gmrfx.h")
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))

# Julia ccall type -> canonical C type (const and parameter names stripped, pointers as "T*")
JL2C = {
    "Int32": "int32_t", "Int64": "int64_t", "Float64": "double", "Cvoid": "void", "Cstring": "char*",
    "Ptr{Cvoid}": "HANDLE*", "Ptr{Float64}": "double*", "Ptr{Int64}": "int64_t*", "Ptr{Int32}": "int32_t*",
    "Ref{Int64}": "int64_t*", "Ref{Int32}": "int32_t*", "Ref{Float64}": "double*", "Ref{Opts}": "gmrfx_opts*",
    "Ref{Ptr{Cvoid}}": "HANDLE**", "Ptr{Ptr{Cvoid}}": "HANDLE**",
}


def _split_top(s):
    """split at commas that are not inside () {} []"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(text, start):
    """text[start] == '(' -> index just past its matching ')'"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced parentheses in GMRFX.jl")


def julia_ccalls():
    src = open(JL).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(gmrfx_\w+),\s*LIB\)\s*,", src):
        end = _balanced(src, m.start() + len("ccall"))
        inner = src[m.end():end - 1]
        parts = _split_top(inner)
        ret, types = parts[0], parts[1]
        assert types.startswith("(") and types.endswith(")"), (m.group(1), types)
        tlist = _split_top(types[1:-1])
        nargs = len(parts) - 2
        calls.append((m.group(1), ret, tlist, nargs, src.count("\n", 0, m.start()) + 1))
    return calls


def header_prototypes():
    h = open(HDR).read()
    h = re.sub(r"/\*.*?\*/", " ", h, flags=re.S)
    h = re.sub(r"//[^\n]*", " ", h)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(gmrfx_\w+)\s*\(([^;{}]*?)\)\s*;", h):
        ret = " ".join(m.group(1).split())
        if "typedef" in ret or "struct" in ret and "(" in ret:
            continue
        args = m.group(3).strip()
        alist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[m.group(2)] = (canon(ret, is_ret=True), [canon(a) for a in alist])
    return protos


def canon(decl, is_ret=False):
    """'const int64_t *colptr' -> 'int64_t*'; 'gmrfx_handle **out' -> 'HANDLE**'"""
    d = decl.replace("const", " ").replace("struct", " ")
    stars = d.count("*")
    d = d.replace("*", " ")
    toks = d.split()
    base = toks[0] if (is_ret or len(toks) == 1) else " ".join(toks[:-1])       # drop the parameter name
    base = {"gmrfx_handle": "HANDLE", "long long": "int64_t", "int": "int32_t", "char": "char"}.get(base, base)
    return base + "*" * stars


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 20, "the shim binds at least the seam-B protocol"
    for name, ret, tlist, nargs, line in calls:
        assert name in protos, f"GMRFX.jl:{line}: {name} is not declared in include/gmrfx.h"
        cret, cargs = protos[name]
        assert JL2C[ret] == cret, f"GMRFX.jl:{line}: {name} returns {cret}, the shim says {ret}"
        assert len(tlist) == len(cargs), f"GMRFX.jl:{line}: {name} takes {len(cargs)} arguments, the type tuple has {len(tlist)}"
        assert nargs == len(tlist), f"GMRFX.jl:{line}: {name}: {nargs} values passed for {len(tlist)} types"
        for k, (jt, ct) in enumerate(zip(tlist, cargs)):
            assert jt in JL2C, f"GMRFX.jl:{line}: {name} argument {k}: unknown Julia type {jt}"
            want = JL2C[jt]
            # a `const gmrfx_handle *` and a `gmrfx_handle *` are the same pointer; void* scratch / device pointers too
            ok = want == ct or (want == "HANDLE*" and ct in ("void*",))
            assert ok, f"GMRFX.jl:{line}: {name} argument {k}: header has {ct}, the shim passes {jt}"


def test_seam_b_protocol_is_bound():
    """the calls behind the WorkspaceBackend methods (backend.jl:8-30) all appear in the shim"""
    names = {c[0] for c in julia_ccalls()}
    for need in ("gmrfx_create", "gmrfx_destroy", "gmrfx_clone", "gmrfx_refactorize", "gmrfx_solve", "gmrfx_logdet",
                 "gmrfx_selinv_diag", "gmrfx_selinv_nnz", "gmrfx_selinv_csc", "gmrfx_selinv_extract", "gmrfx_selinv_dot",
                 "gmrfx_backward_solve", "gmrfx_get_perm", "gmrfx_last_error", "gmrfx_last_create_error"):
        assert need in names, need


def test_opts_struct_matches_gmrfx_opts():
    from gmrfx import _lib
    src = open(JL).read()
    body = src[src.index("struct Opts"):]
    body = body[:body.index("\nend")]
    jl_fields = re.findall(r"^\s*(\w+)::([\w{}]+)", body, flags=re.M)
    h = open(HDR).read()
    hb = h[h.index("typedef struct gmrfx_opts"):h.index("} gmrfx_opts;")]
    hb = re.sub(r"/\*.*?\*/", " ", hb, flags=re.S)
    c_fields = []
    for m in re.finditer(r"(const\s+)?(int32_t|int64_t|double)\s*(\*?)\s*([\w\s,]+);", hb):
        for nm in m.group(4).split(","):
            c_fields.append((nm.strip(), m.group(2) + ("*" if m.group(3) else "")))
    jmap = {"Int32": "int32_t", "Int64": "int64_t", "Float64": "double", "Ptr{Float64}": "double*"}
    assert [(n, jmap[t]) for n, t in jl_fields] == c_fields
    # size / offsets as the C compiler lays them out (ctypes mirror, itself checked against the library in test_cabi.py)
    size = {"int32_t": 4, "int64_t": 8, "double": 8, "double*": 8}
    off = 0
    for n, t in c_fields:
        a = size[t]
        off = (off + a - 1) // a * a
        assert getattr(_lib.GmrfxOpts, n).offset == off, n
        off += a
    assert (off + 7) // 8 * 8 == ctypes.sizeof(_lib.GmrfxOpts)
