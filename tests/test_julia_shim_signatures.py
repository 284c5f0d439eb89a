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
JL_EXT = os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd", "julia", "ext", "GMRFXAMDGPUExt.jl")      # ROCArray methods (*_dev entry points)
HDR = os.path.join(ROOT, "include", "gmrfx.h")
HDR_RCCL = os.path.join(ROOT, "include", "gmrfx_rccl.h")       # the native RCCL driver (libgmrfx_rccl.so), bound through LIB_RCCL
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))

# Julia ccall type -> canonical C type (const and parameter names stripped, pointers as "T*")
JL2C = {
    "Int32": "int32_t", "Int64": "int64_t", "Float64": "double", "Cvoid": "void", "Cstring": "char*",
    "Ptr{Cvoid}": "HANDLE*", "Ptr{Float64}": "double*", "Ptr{Int64}": "int64_t*", "Ptr{Int32}": "int32_t*",
    "Ref{Int64}": "int64_t*", "Ref{Int32}": "int32_t*", "Ref{Float64}": "double*", "Ref{Opts}": "gmrfx_opts*",
    "Ref{Ptr{Cvoid}}": "HANDLE**", "Ptr{Ptr{Cvoid}}": "HANDLE**", "Ptr{UInt8}": "uint8_t*",
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


def julia_ccalls(path=None):
    if path is None:
        return julia_ccalls(JL) + julia_ccalls(JL_EXT)
    src = open(path).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(gmrfx_\w+),\s*LIB(?:_RCCL)?\)\s*,", src):
        if ("LIB_RCCL" in m.group(0)) != m.group(1).startswith("gmrfx_rccl_"):
            raise AssertionError(f"{m.group(1)} is bound through the wrong library constant")
        end = _balanced(src, m.start() + len("ccall"))
        inner = src[m.end():end - 1]
        parts = _split_top(inner)
        ret, types = parts[0], parts[1]
        assert types.startswith("(") and types.endswith(")"), (m.group(1), types)
        tlist = _split_top(types[1:-1])
        nargs = len(parts) - 2
        calls.append((m.group(1), ret, tlist, nargs, f"{os.path.basename(path)}:{src.count(chr(10), 0, m.start()) + 1}"))
    return calls


def header_prototypes():
    h = open(HDR).read() + "\n" + open(HDR_RCCL).read()
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
    base = {"gmrfx_handle": "HANDLE", "gmrfx_rccl": "HANDLE", "long long": "int64_t", "int": "int32_t", "char": "char"}.get(base, base)
    return base + "*" * stars


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 20, "the shim binds at least the seam-B protocol"
    for name, ret, tlist, nargs, line in calls:
        assert name in protos, f"{line}: {name} is not declared in include/gmrfx.h"
        cret, cargs = protos[name]
        assert JL2C[ret] == cret, f"{line}: {name} returns {cret}, the shim says {ret}"
        assert len(tlist) == len(cargs), f"{line}: {name} takes {len(cargs)} arguments, the type tuple has {len(tlist)}"
        assert nargs == len(tlist), f"{line}: {name}: {nargs} values passed for {len(tlist)} types"
        for k, (jt, ct) in enumerate(zip(tlist, cargs)):
            assert jt in JL2C, f"{line}: {name} argument {k}: unknown Julia type {jt}"
            want = JL2C[jt]
            # a `const gmrfx_handle *` and a `gmrfx_handle *` are the same pointer; void* scratch / device pointers too
            ok = want == ct or (want == "HANDLE*" and ct in ("void*",)) or (want == "double*" and ct == "double*")
            assert ok, f"{line}: {name} argument {k}: header has {ct}, the shim passes {jt}"


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


def test_device_entry_points_are_bound_for_rocarrays():
    """the *_dev calls (operands resident in HBM: the call bench.py times) are bound in the AMDGPU.jl extension, with device
    pointers passed as Ptr{Float64}"""
    names = {c[0] for c in julia_ccalls(JL_EXT)}
    for need in ("gmrfx_refactorize_dev", "gmrfx_refactorize_solve_dev", "gmrfx_solve_dev", "gmrfx_backward_solve_dev",
                 "gmrfx_refactorize_logpdf_dev"):
        assert need in names, need
    src = open(JL_EXT).read()
    assert "module GMRFXAMDGPUExt" in src and "using AMDGPU" in src
    # every function the extension adds methods to exists in the parent module
    parent = open(JL).read()
    for fn in ("refactorize_solve!", "backend_solve!", "backend_backward_solve!", "logpdf_terms"):
        assert re.search(r"function\s+" + re.escape(fn), parent), fn


def test_workspace_solve_is_a_real_method_on_this_backend():
    """the reference's workspace_solve (gmrf_workspace.jl:207-215) is overloaded for workspaces on MI355XBackend -- as code, not as a
    comment -- for vectors and matrices separately (one AbstractVecOrMat method would be ambiguous with the reference's two), and a
    stale factorisation goes down as the pipelined host call gmrfx_refactorize_solve"""
    code = "\n".join(l for l in open(JL).read().splitlines() if not l.lstrip().startswith("#"))
    assert re.search(r"G\.workspace_solve\(ws::GMRFWorkspace\{<:Any,\s*MI355XBackend\},\s*b::AbstractVector\)", code)
    assert re.search(r"G\.workspace_solve\(ws::GMRFWorkspace\{<:Any,\s*MI355XBackend\},\s*B::AbstractMatrix\)", code)
    body = code[code.index("function _workspace_solve"):]
    body = body[:body.index("\nend")]
    assert "refactorize_solve!(ws.backend, Symmetric(ws.Q), B)" in body and "ws.numeric_valid = true" in body
    assert "ws.selinv_valid = false" in body and "ws.logdet_valid = false" in body
    assert any(c[0] == "gmrfx_refactorize_solve" for c in julia_ccalls(JL))


def test_batched_rand_is_a_real_method_on_both_seams():
    """rand(d, k) reaches the backend as ONE multi-column backward sweep (round 5): Distributions' matrix method `_rand!` is
    overloaded -- as code -- for WorkspaceGMRFs on MI355XBackend (reference: one vector at a time, workspace_gmrf.jl:275-286) and for
    GMRFs whose cache algorithm is MI355XCholesky (gmrf.jl:271-281); both go through backend_backward_solve(b, Z::Matrix), which binds
    gmrfx_backward_solve with the matrix's column stride; a failed factor in the pipelined solve throws PosDefException."""
    code = "\n".join(l for l in open(JL).read().splitlines() if not l.lstrip().startswith("#"))
    m = re.search(r"function Distributions\._rand!\(rng::AbstractRNG,\s*d::G\.WorkspaceGMRF\{<:Any,\s*MI355XBackend\},\s*X::AbstractMatrix\{<:Real\}\)(.*?)\nend", code, re.S)
    assert m, "no batched _rand! for WorkspaceGMRF on MI355XBackend"
    body = m.group(1)
    assert "randn!(rng" in body and "backend_backward_solve(d.workspace.backend, Z)" in body and "ci.L_c \\" in body and "d.mean" in body
    # (round 6: restricted by dispatch to GMRFs whose cache carries MI355XCholesky -- no method on a bare G.GMRF any more)
    m = re.search(r"function Distributions\._rand!\(rng::AbstractRNG,\s*d::MI355XGMRF,\s*X::AbstractMatrix\{<:Real\}\)(.*?)\nend\n", code, re.S)
    assert m and "G.backward_solve(d.linsolve_cache, Z)" in m.group(1) and "randn!(rng" in m.group(1) and "d.mean" in m.group(1)
    assert re.search(r"G\._backward_solve_impl\(cache,\s*Z::Matrix\{Float64\},\s*::MI355XCholesky\)", code)
    assert re.search(r"function backend_backward_solve\(b::MI355XBackend,\s*Zm::Matrix\{Float64\}\)", code)
    assert len(re.findall(r"info\[\] > 0 && throw\(PosDefException", code)) >= 1
    ext = "\n".join(l for l in open(JL_EXT).read().splitlines() if not l.lstrip().startswith("#"))
    assert "info[] > 0 && throw(PosDefException" in ext


# ---- round 6: no type piracy, and every foreign name the shim extends exists ------------------------------------------------
OWNED = ("MI355XBackend", "MI355XCholesky", "MI355XCacheval", "MI355XGMRF", "MI355XLinearCache", "Handle", "Opts", "ShardedMI355X")
FOREIGN = ("G", "Distributions", "LinearSolve", "SciMLBase", "Base", "LinearAlgebra", "SparseArrays")


def foreign_method_definitions(path):
    """every `function M.f(sig...)` / `M.f(sig...) = ...` with M a foreign module, as (module, name, signature text, line)"""
    code = "\n".join("" if l.lstrip().startswith("#") else l for l in open(path).read().splitlines())
    out = []
    for m in re.finditer(r"^(?:function\s+)?((?:%s))\.([\w!]+)\(" % "|".join(FOREIGN), code, flags=re.M):
        start = m.end() - 1
        end = _balanced(code, start)
        rest = code[end:end + 40].lstrip()
        is_def = m.group(0).startswith("function") or rest.startswith("=") and not rest.startswith("==") or rest.startswith("where")
        if not is_def:
            continue
        out.append((m.group(1), m.group(2), code[start + 1:end - 1], f"{os.path.basename(path)}:{code.count(chr(10), 0, m.start()) + 1}"))
    return out


def test_no_type_piracy_every_foreign_method_carries_an_owned_type():
    """A method added to a function of ANOTHER package must dispatch on a type this plug-in owns, otherwise loading the plug-in
    changes behaviour for callers that never asked for it (round-5 findings: `Distributions._rand!(rng, d::G.GMRF, X)` replaced the
    matrix sampler of every GMRF; `G.sparse_approximate_cholesky!(::Matrix{Float64}, ::SparseMatrixCSC{Float64, Int})` replaced the
    reference's own method). Aliases count through their definition: MI355XGMRF must bottom out in MI355XCholesky."""
    defs = foreign_method_definitions(JL) + foreign_method_definitions(JL_EXT)
    assert len(defs) >= 15
    for mod, name, sig, line in defs:
        assert any(re.search(r"\b%s\b" % t, sig) for t in OWNED + ("ROCArray", "ROCVector", "ROCMatrix")) or \
            (os.path.basename(JL_EXT) in line and "MI355X" in sig), f"{line}: {mod}.{name}({sig}) dispatches on foreign types only (type piracy)"
    src = open(JL).read()
    m = re.search(r"const MI355XLinearCache = (.*)", src)
    assert m and "MI355XCholesky" in m.group(1) and "LinearSolve.LinearCache{" in m.group(1)
    # the algorithm is the FIFTH parameter of LinearCache{TA, Tb, Tu, Tp, Talg, ...}: four wildcards in front of it
    assert m.group(1).split("LinearSolve.LinearCache{")[1].split("MI355XCholesky")[0].count("<:Any") == 4
    m = re.search(r"const MI355XGMRF = (.*)", src)
    # GMRF{T, VMean, VInfo, PrecisionMap, QSqrt, Cache, RBMCStrat} (reference src/gmrf.jl:144-156): the cache is the SIXTH
    assert m and "G.GMRF{" in m.group(1) and m.group(1).split("G.GMRF{")[1].split("<:MI355XLinearCache")[0].count("<:Any") == 5
    assert "body.parameters[5] === fieldtype(body, :alg)" in src, "the load-time guard for LinearCache's parameter order is gone"
    code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    assert not re.search(r"_rand!\(rng::AbstractRNG,\s*d::G\.GMRF\s*,", code), "the unrestricted GMRF sampler is back"


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference tree is not on this box (GPU boxes)")
def test_every_reference_name_the_shim_extends_or_calls_exists_in_the_reference():
    """grep-level: every `G.name` the shim touches (methods it adds, functions it calls, types it dispatches on) is defined in
    /root/reference/src or ext -- a renamed hook would otherwise only surface when somebody loads the plug-in under Julia"""
    ref = ""
    for base in ("src", "ext"):
        for dp, _, fs in os.walk(os.path.join(REF, base)):
            for f in fs:
                if f.endswith(".jl"):
                    ref += open(os.path.join(dp, f), errors="replace").read() + "\n"
    names = set()
    for path in (JL, JL_EXT):
        code = "\n".join("" if l.lstrip().startswith("#") else l for l in open(path).read().splitlines())
        names |= set(re.findall(r"\bG\.([A-Za-z_][\w!]*)", code))
    assert len(names) >= 15
    for nm in sorted(names):
        e = re.escape(nm)
        defined = re.search(r"(?:^|\n)\s*(?:function|struct|mutable struct|abstract type|const|macro)\s+(?:\w+\.)?%s(?![\w!])" % e, ref) or \
            re.search(r"(?:^|\n)\s*(?:\w+\.)?%s\([^\n]*\)\s*(?:where[^\n=]*)?=" % e, ref) or \
            re.search(r"(?:^|\n)\s*(?:@kwdef\s+)?(?:mutable\s+)?struct\s+%s(?![\w!])" % e, ref)
        assert defined, f"G.{nm} is used by the shim but not defined anywhere under {REF}/src or ext"
    # the GMRF struct still has its cache as sixth type parameter
    m = re.search(r"struct GMRF\{(.*?)\}\s*<:", ref, re.S)
    params = [p.strip().split("<:")[0].strip() for p in _split_top(m.group(1).replace("\n", " "))]
    assert params[5] == "Cache" and len(params) == 7, params


def test_native_rccl_driver_is_bound_in_the_shim():
    """INTEGRATION.md section 6 as code: the sharded protocol is reachable from the Julia host through libgmrfx_rccl.so (LIB_RCCL) --
    communicator id, driver creation with the shard options passed to gmrfx_create, refactorise / solve / sample with ROCArray
    operands (extension), log-determinant, selected-inverse diagonal, the row mask of a row-sharded B."""
    names = {c[0] for c in julia_ccalls()}
    for need in ("gmrfx_rccl_unique_id", "gmrfx_rccl_create", "gmrfx_rccl_destroy", "gmrfx_rccl_last_error", "gmrfx_rccl_refactorize",
                 "gmrfx_rccl_solve", "gmrfx_rccl_backward_solve", "gmrfx_rccl_logdet", "gmrfx_rccl_selinv_diag", "gmrfx_rccl_needed_rows"):
        assert need in names, need
    src = open(JL).read()
    assert "shard_rank = shard_rank, shard_world = shard_world, shard_min_top = shard_min_top" in src
    assert re.search(r"mutable struct ShardedMI355X", src) and "const LIB_RCCL" in src
