# GMRFX.jl -- thin ccall layer over libgmrfx.so (include/gmrfx.h), and the two plug-ins that attach
# it to GaussianMarkovRandomFields.jl:
#   * seam B: `MI355XBackend <: WorkspaceBackend`            (src/workspace/backend.jl:8-30)
#   * seam A: `MI355XCholesky` LinearSolve algorithm + hooks  (ext/GaussianMarkovRandomFieldsPardiso.jl:10-80)
# NOTE: this image has no Julia; the same call sequences are exercised by the Python ctypes binding
# (gmrfx/_lib.py, gmrfx/backend.py) and the test mirrors of the host code above the seams
# (tests/mirror/workspace.py = seam B state machine, tests/mirror/linsolve.py = seam A: `MI355XCacheval`,
# `solve!`, the hooks below, `deepcopy(cache)`). Keep them in sync.
module GMRFX

using LinearAlgebra, SparseArrays, Random
import Distributions
import GaussianMarkovRandomFields as G
import GaussianMarkovRandomFields: WorkspaceBackend, refactorize!, backend_solve, compute_logdet,
    compute_selinv!, get_selinv, get_selinv_diag, backend_backward_solve, selinv_dot, selinv_extract_at,
    GMRFWorkspace

const LIB = get(ENV, "GMRFX_LIB", joinpath(@__DIR__, "..", "libgmrfx.so"))

Base.@kwdef struct Opts            # mirrors gmrfx_opts
    struct_size::Int32 = 0
    uplo::Int32 = 0
    ordering::Int32 = 0
    device::Int32 = -1
    symbolic_only::Int32 = 0
    check_posdef::Int32 = 0
    nd_leaf::Int32 = 0
    relax_cols::Int32 = 0
    relax_zeros::Float64 = 0.0
    coord_dim::Int32 = 0
    reserved0::Int32 = 0
    coords::Ptr{Float64} = C_NULL
    shard_rank::Int32 = 0          # one factorisation sharded over several GPUs (include/gmrfx.h); 0 / 1 = unsharded
    shard_world::Int32 = 1
    shard_min_top::Int32 = 0       # > 0: at least this many top fronts; with shard_world == 1 a sharded handle of one rank (testing)
    reserved1::Int32 = 0
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(p)
        h = new(p)
        finalizer(x -> (x.ptr == C_NULL || ccall((:gmrfx_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.ptr); x.ptr = C_NULL), h)
        return h
    end
end

function check(code::Int32, h::Union{Handle, Nothing} = nothing)
    code == 0 && return nothing
    msg = h === nothing ? unsafe_string(ccall((:gmrfx_last_create_error, LIB), Cstring, ())) :
        unsafe_string(ccall((:gmrfx_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr))
    code == 1 && throw(ArgumentError(msg))
    code == 5 && throw(PosDefException(1))
    error("gmrfx error $code: $msg")
end

# `ordering` (src/workspace/backend.jl:73-153): nothing -> libgmrfx's own nested dissection; :natural; a permutation
# vector; anything else CliqueTrees accepts (an EliminationAlgorithm such as CliqueTrees.MMD(), an (order, index)
# tuple) or a `PinDenseColumns` wrapper is resolved ONCE by the reference's own `ordering_permutation(A, ordering)`
# (backend.jl:86-133) and reaches the library as an explicit permutation. Nothing is dropped silently: whatever
# that resolver rejects throws there.
resolve_ordering(Q, ordering::Nothing) = nothing
resolve_ordering(Q, ordering::Symbol) = ordering === :natural ? nothing : throw(ArgumentError("unknown ordering $ordering"))
resolve_ordering(Q, ordering::AbstractVector{<:Integer}) = Vector{Int}(ordering)
resolve_ordering(Q, ordering) = Vector{Int}(G.ordering_permutation(Q, ordering))

function create(Q::SparseMatrixCSC{Float64, Int}; ordering = nothing, coords = nothing, device = -1,
        check_posdef = false, shard_rank = 0, shard_world = 1, shard_min_top = 0)
    n = size(Q, 1)
    perm = resolve_ordering(Q, ordering)
    perm === nothing || isperm(perm) && length(perm) == n || throw(ArgumentError("ordering is not a permutation of 1:$n"))
    C = coords === nothing ? nothing : Matrix{Float64}(transpose(coords))     # dim x n == n x dim row-major
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Q perm C begin
        o = Opts(struct_size = sizeof(Opts), device = device, check_posdef = check_posdef,
            shard_rank = shard_rank, shard_world = shard_world, shard_min_top = shard_min_top,
            ordering = ordering === :natural ? 1 : 0,
            coord_dim = C === nothing ? 0 : size(C, 1), coords = C === nothing ? C_NULL : pointer(C))
        check(ccall((:gmrfx_create, LIB), Int32,
            (Int64, Ptr{Int64}, Ptr{Int64}, Int32, Ptr{Int64}, Ref{Opts}, Ref{Ptr{Cvoid}}),
            n, SparseArrays.getcolptr(Q), rowvals(Q), 1, perm === nothing ? C_NULL : pointer(perm), Ref(o), out))
    end
    return Handle(out[])
end

# ---------------------------------------------------------------------------------- seam B
mutable struct MI355XBackend <: WorkspaceBackend
    h::Handle
    n::Int
    selinv_cache::Union{Nothing, SparseMatrixCSC{Float64, Int}}
    selinv_diag_cache::Union{Nothing, Vector{Float64}}
end

function MI355XBackend(Q::Symmetric{Float64, <:SparseMatrixCSC{Float64}}; ordering = nothing, coords = nothing, device = -1)
    A = parent(Q)                                   # both triangles stored; uplo=:U defines Q
    b = MI355XBackend(create(A; ordering, coords, device), size(A, 1), nothing, nothing)
    refactorize!(b, Q)
    return b
end

function refactorize!(b::MI355XBackend, Q::Symmetric)
    nz = nonzeros(parent(Q))
    info = Ref{Int64}(0)
    GC.@preserve nz check(ccall((:gmrfx_refactorize, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int64}), b.h.ptr, nz, info), b.h)
    b.selinv_cache = nothing
    b.selinv_diag_cache = nothing
    return nothing                                  # never throws on indefiniteness (backend.jl:184)
end

# workspace_solve(ws, B) on a workspace whose values were just updated (gmrf_workspace.jl:170-178, 207-215: ensure_numeric! then
# backend_solve) as ONE pipelined call: the forward sweep follows the factorisation up the tree on a side stream. The host
# transfers are SERIAL -- B goes up in front of the factorisation (staged through page-locked memory by host threads when
# pageable), X comes down in slices behind the backward sweep (csrc/device.cpp, host_upload / host_download: a transfer beside
# the factorisation slows its launch chain by more than it hides). Same bits as refactorize! + backend_solve.
# The reference's workspace_solve ends in `b.factor \ rhs`, which throws PosDefException on a failed factor (backend.jl:178-193):
# so does this call -- a non-positive pivot never comes back as a silent NaN solution.
function refactorize_solve!(b::MI355XBackend, Q::Symmetric, rhs::AbstractVecOrMat)
    nz = nonzeros(parent(Q))
    B = Matrix{Float64}(reshape(rhs, b.n, :)); X = similar(B)
    info = Ref{Int64}(0)
    GC.@preserve nz B X check(ccall((:gmrfx_refactorize_solve, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ref{Int64}),
        b.h.ptr, nz, B, b.n, size(B, 2), X, b.n, info), b.h)
    b.selinv_cache = nothing
    b.selinv_diag_cache = nothing
    info[] > 0 && throw(PosDefException(Int(info[])))
    return rhs isa AbstractVector ? vec(X) : X
end

# The reference's workspace_solve (gmrf_workspace.jl:207-215) for workspaces on this backend: a STALE factorisation and the solve
# behind it go down as the one pipelined call above (this is the call bench.py times, through the host entry point); a current
# factorisation is a plain backend_solve. Two methods, as in the reference (one AbstractVecOrMat method would be ambiguous with its).
function _workspace_solve(ws::GMRFWorkspace, B::AbstractVecOrMat)
    ws.numeric_valid && return backend_solve(ws.backend, B isa AbstractVector ? B : Matrix{Float64}(B))
    X = refactorize_solve!(ws.backend, Symmetric(ws.Q), B)          # = ensure_numeric!(ws) + backend_solve(ws.backend, B); throws on a failed factor
    ws.numeric_valid = true
    ws.selinv_valid = false
    ws.logdet_valid = false
    return X
end
G.workspace_solve(ws::GMRFWorkspace{<:Any, MI355XBackend}, b::AbstractVector) = _workspace_solve(ws, b)
G.workspace_solve(ws::GMRFWorkspace{<:Any, MI355XBackend}, B::AbstractMatrix) = _workspace_solve(ws, B)

# Device-resident operands (the *_dev entry points of include/gmrfx.h) are bound for AMDGPU.jl's ROCArray in the package
# extension ext/GMRFXAMDGPUExt.jl (weak dependency: loaded when AMDGPU is): refactorize!(b, d_nz), refactorize_solve!(X, b, d_nz, B),
# backend_solve!(X, b, B), backend_backward_solve!(X, b, Z), logpdf_terms(b, d_nz, X; mean).
function backend_solve! end
function backend_backward_solve! end
function logpdf_terms end

function backend_solve(b::MI355XBackend, rhs::AbstractVector)
    B = Vector{Float64}(rhs); X = similar(B)
    GC.@preserve B X check(ccall((:gmrfx_solve, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
        b.h.ptr, B, b.n, 1, X, b.n), b.h)
    return X
end
function backend_solve(b::MI355XBackend, RHS::Matrix{Float64})
    X = similar(RHS)
    GC.@preserve RHS X check(ccall((:gmrfx_solve, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
        b.h.ptr, RHS, stride(RHS, 2), size(RHS, 2), X, stride(X, 2)), b.h)
    return X
end

function compute_logdet(b::MI355XBackend)
    out = Ref{Float64}(0)
    check(ccall((:gmrfx_logdet, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}), b.h.ptr, out), b.h)
    return out[]
end

compute_selinv!(b::MI355XBackend) = nothing         # lazy, like CHOLMODBackend (backend.jl:215-221)

function get_selinv(b::MI355XBackend)
    if b.selinv_cache === nothing
        nnzr = Ref{Int64}(0)
        check(ccall((:gmrfx_selinv_nnz, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}), b.h.ptr, nnzr), b.h)
        colptr = Vector{Int}(undef, b.n + 1); rv = Vector{Int}(undef, nnzr[]); nz = Vector{Float64}(undef, nnzr[])
        GC.@preserve colptr rv nz check(ccall((:gmrfx_selinv_csc, LIB), Int32,
            (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), b.h.ptr, 1, colptr, rv, nz), b.h)
        b.selinv_cache = SparseMatrixCSC(b.n, b.n, colptr, rv, nz)
    end
    return b.selinv_cache
end

function get_selinv_diag(b::MI355XBackend)
    if b.selinv_diag_cache === nothing
        d = Vector{Float64}(undef, b.n)
        GC.@preserve d check(ccall((:gmrfx_selinv_diag, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), b.h.ptr, d), b.h)
        b.selinv_diag_cache = d
    end
    return b.selinv_diag_cache
end

function selinv_extract_at(b::MI355XBackend, B::SparseMatrixCSC)
    out = Vector{Float64}(undef, nnz(B))
    cp = Vector{Int}(SparseArrays.getcolptr(B)); rv = Vector{Int}(rowvals(B))
    GC.@preserve cp rv out check(ccall((:gmrfx_selinv_extract, LIB), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Int32, Ptr{Float64}), b.h.ptr, b.n, cp, rv, 1, out), b.h)
    return SparseMatrixCSC(size(B)..., cp, rv, out)
end
# values of Sigma are gathered on the GPU; the contraction stays in Julia so B may carry Duals
selinv_dot(b::MI355XBackend, B::SparseMatrixCSC) = dot(nonzeros(selinv_extract_at(b, B)), nonzeros(B))

# Float64 B: contracted on the device. Anything else (ForwardDiff.Dual values): gather Sigma on B's pattern and
# contract in Julia, as the generic fallback does.
function selinv_dot(b::MI355XBackend, B::SparseMatrixCSC{Float64, Int})
    out = Ref{Float64}(0.0)
    cp, rv, nz = B.colptr, B.rowval, B.nzval
    GC.@preserve cp rv nz check(ccall((:gmrfx_selinv_dot, LIB), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32, Ref{Float64}),
        b.h.ptr, size(B, 2), cp, rv, nz, 1, out), b.h)
    return out[]
end
selinv_dot(b::MI355XBackend, B::AbstractMatrix) = dot(selinv_extract_at(b, sparse(B)), B)

# diag(A * Sigma * A') for a sparse design matrix (rows of A = columns of At): src/linear_predictor_marginals.jl:125-165
function row_diag_AΣAt(b::MI355XBackend, A::SparseMatrixCSC{Float64, Int})
    At = sparse(transpose(A))
    out = Vector{Float64}(undef, size(A, 1))
    cp, rv, nz = At.colptr, At.rowval, At.nzval
    GC.@preserve cp rv nz out check(ccall((:gmrfx_selinv_row_diag, LIB), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32, Ptr{Float64}),
        b.h.ptr, size(A, 1), cp, rv, nz, 1, out), b.h)
    return out
end

function backend_backward_solve(b::MI355XBackend, x::AbstractVector)
    Z = Vector{Float64}(x); X = similar(Z)          # copies views (backend.jl:282-283)
    GC.@preserve Z X check(ccall((:gmrfx_backward_solve, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
        b.h.ptr, Z, b.n, 1, X, b.n), b.h)
    return X
end
# batched sampling: k samples in one sweep instead of the reference's k single-RHS solves (src/gmrf.jl:271-281)
function backend_backward_solve(b::MI355XBackend, Zm::Matrix{Float64})
    X = similar(Zm)
    GC.@preserve Zm X check(ccall((:gmrfx_backward_solve, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
        b.h.ptr, Zm, stride(Zm, 2), size(Zm, 2), X, stride(X, 2)), b.h)
    return X
end

# ---- rand(d, k): k samples in ONE backward sweep ------------------------------------------------------------------------------
# The reference's `_rand!` takes one vector (src/gmrf.jl:271-281, src/workspace/workspace_gmrf.jl:275-286) and Distributions' matrix
# method `_rand!(rng, d, X::AbstractMatrix)` loops it over the columns: rand(d, 256) is 256 single-RHS sweeps with 256 host round
# trips. On this backend the matrix method is ONE `backend_backward_solve(b, Z::Matrix)`: randn! fills Z column by column in the
# order the column loop would draw it, so the same rng gives the same samples (to 1e-12: a wide pass and a one-column pass take kernels
# that add the same terms in different orders: tests/mirror/workspace_gmrf.py, tests/test_seam_a_and_constraints.py); the mean and the constraint correction
# (x -= A~' (L_c \ (A x - e)), workspace_gmrf.jl:280-284) are applied to all columns at once.
function Distributions._rand!(rng::AbstractRNG, d::G.WorkspaceGMRF{<:Any, MI355XBackend}, X::AbstractMatrix{<:Real})
    G.ensure_loaded!(d)
    Z = randn!(rng, Matrix{Float64}(undef, size(X, 1), size(X, 2)))
    G.ensure_numeric!(d.workspace)
    Y = backend_backward_solve(d.workspace.backend, Z)
    Y .+= d.mean
    if d.constraints !== nothing
        ci = d.constraints
        Y .-= ci.A_tilde_T * (ci.L_c \ (ci.matrix * Y .- ci.vector))
    end
    X .= Y
    return X
end

ordering_permutation(b::MI355XBackend) = (p = Vector{Int}(undef, b.n);
    ccall((:gmrfx_get_perm, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}), b.h.ptr, 1, p); p)

# selector, like GMRFWorkspace(Q, CliqueTreesBackend; alg) (src/workspace/cliquetrees_backend.jl:132-150)
function G.GMRFWorkspace(Q::SparseMatrixCSC{Float64}, ::Type{MI355XBackend}; kwargs...)
    n = size(Q, 1)
    n == size(Q, 2) || throw(ArgumentError("Q must be square"))
    backend = MI355XBackend(Symmetric(Q); kwargs...)
    return GMRFWorkspace{Float64, MI355XBackend}(copy(Q), backend, zeros(n), zeros(n), true, false, false, 0.0, 1, 0)
end

# Newton loop with Q resident on the device (include/gmrfx.h "Newton loop"): the workspace-side
# `_update_hessian!` (src/workspace/gaussian_approximation.jl:103-129) becomes one upload of the prior values
# and of the index map (`_diag_indices(ws.Q)` or `_sparse_hessian_map(ws.Q, H)`, 1-based positions into nzval)
# followed, per iterate, by the Hessian's values only.
function set_prior!(b::MI355XBackend, prior_nzval::Vector{Float64}, hess_map::Vector{Int})
    GC.@preserve prior_nzval hess_map check(ccall((:gmrfx_set_prior, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Int64, Int32), b.h.ptr, prior_nzval, hess_map, length(hess_map), 1), b.h)
    return nothing
end
function refactorize_update!(b::MI355XBackend, hvals::Vector{Float64})
    info = Ref{Int64}(0)
    GC.@preserve hvals check(ccall((:gmrfx_refactorize_update, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int64}),
        b.h.ptr, hvals, info), b.h)
    b.selinv_cache = nothing; b.selinv_diag_cache = nothing
    return nothing
end

# one Newton iterate (gaussian_approximation.jl:103-129) in one pipelined call: Hessian values in, solve for the new mean out
function refactorize_update_solve!(b::MI355XBackend, hvals::Vector{Float64}, rhs::AbstractVecOrMat)
    B = Matrix{Float64}(reshape(rhs, b.n, :)); X = similar(B)
    info = Ref{Int64}(0)
    GC.@preserve hvals B X check(ccall((:gmrfx_refactorize_update_solve, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ref{Int64}),
        b.h.ptr, hvals, B, b.n, size(B, 2), X, b.n, info), b.h)
    b.selinv_cache = nothing; b.selinv_diag_cache = nothing
    return rhs isa AbstractVector ? vec(X) : X
end

# dot(r, Q * r), r = x - mean, on the device with the values of the last refactorize! (the quadratic form of
# logpdf(::WorkspaceGMRF, z), src/workspace/workspace_gmrf.jl:288-292, and sqmahal, src/gmrf.jl:94-97).
# X: n vector or n x k matrix (one value per column).
function sqmahal(b::MI355XBackend, X::StridedVecOrMat{Float64}; mean::Union{Nothing, Vector{Float64}} = nothing)
    nvec = size(X, 2)
    out = Vector{Float64}(undef, nvec)
    mu = mean === nothing ? Ptr{Float64}(C_NULL) : pointer(mean)
    GC.@preserve X mean out check(ccall((:gmrfx_quadform, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
        b.h.ptr, C_NULL, X, X isa AbstractVector ? length(X) : stride(X, 2), nvec, mu, out), b.h)
    return X isa AbstractVector ? out[1] : out
end

# KL (Vecchia) sparse approximate Cholesky on the GPU (src/kl_cholesky/kl_cholesky.jl:32-55): same contract as the
# reference method, for a dense Float64 covariance. One task per column: its row indices in descending order.
# Selected by a trailing `MI355XBackend` argument, like `GMRFWorkspace(Q, MI355XBackend)`: a method on (Matrix{Float64},
# SparseMatrixCSC{Float64, Int}) alone would REPLACE the reference's own for those types (piracy) and take its AD-transparent
# generic path away from every caller that merely loads the plug-in.
function G.sparse_approximate_cholesky!(Θ::Matrix{Float64}, L::SparseMatrixCSC{Float64, Int}, ::Type{MI355XBackend}; device::Integer = -1)
    n = size(L, 2)
    rows = similar(L.rowval)
    for k in 1:n
        r = nzrange(L, k)
        rows[r] .= @view L.rowval[reverse(r)]
    end
    cols = collect(1:n); colptr_t = collect(1:(n + 1)); info = Ref{Int64}(0)
    GC.@preserve Θ L rows cols colptr_t begin
        code = ccall((:gmrfx_kl_cholesky, LIB), Int32,
            (Int64, Ptr{Float64}, Int64, Int32, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
             Int32, Float64, Int32, Ptr{Float64}, Ref{Int64}),
            n, Θ, stride(Θ, 2), 0, L.colptr, n, L.colptr, rows, colptr_t, cols, 1, 1.0e-6, device, L.nzval, info)
    end
    code == 5 && throw(PosDefException(Int(info[])))
    check(code)
    return
end

function Base.deepcopy_internal(b::MI355XBackend, stackdict::IdDict)   # deepcopy(cache) in Newton loops
    haskey(stackdict, b) && return stackdict[b]::MI355XBackend
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:gmrfx_clone, LIB), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), b.h.ptr, out))
    c = MI355XBackend(Handle(out[]), b.n, nothing, nothing)
    stackdict[b] = c
    return c
end

# ---------------------------------------------------------------------------------- one factorisation over several GPUs
# The native driver of the sharded protocol (libgmrfx_rccl.so, include/gmrfx_rccl.h: RCCL point-to-point / broadcast / all-reduce
# between the phases of libgmrfx.so), one Julia process (Distributed.jl worker / MPI rank) per GPU:
#     id = rank == 0 ? rccl_unique_id() : <the 128 bytes from rank 0, by whatever the host uses: Distributed, MPI, a file>
#     sf = ShardedMI355X(Q, world, rank, id; coords)          # same Q on every rank
#     refactorize!(sf, d_nz); solve!(d_X, sf, d_B); logdet(sf) # ROCArray operands: julia/ext/GMRFXAMDGPUExt.jl
# The reference has nothing to replace here (CHOLMOD has one address space: src/workspace/backend.jl:165-209); this is what the
# exchange of Schur-complement blocks stands in for.
const LIB_RCCL = get(ENV, "GMRFX_RCCL_LIB", joinpath(@__DIR__, "..", "libgmrfx_rccl.so"))

function rccl_unique_id()
    id = Vector{UInt8}(undef, 128)
    code = ccall((:gmrfx_rccl_unique_id, LIB_RCCL), Int32, (Ptr{Cvoid},), id)
    code == 0 || error("gmrfx_rccl_unique_id failed ($code)")
    return id
end

mutable struct ShardedMI355X
    h::Handle
    drv::Ptr{Cvoid}
    n::Int
    world::Int
    rank::Int
end

function ShardedMI355X(Q::SparseMatrixCSC{Float64, Int}, world::Integer, rank::Integer, id::Vector{UInt8};
        ordering = nothing, coords = nothing, device = -1, shard_min_top = 0)
    length(id) == 128 || throw(ArgumentError("the communicator id has 128 bytes"))
    h = create(Q; ordering, coords, device, shard_rank = rank, shard_world = world, shard_min_top)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    code = GC.@preserve id ccall((:gmrfx_rccl_create, LIB_RCCL), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
        h.ptr, world, rank, id, C_NULL, out)
    code == 0 || error("gmrfx_rccl_create failed ($code)")
    sf = ShardedMI355X(h, out[], size(Q, 1), world, rank)
    finalizer(x -> (x.drv == C_NULL || ccall((:gmrfx_rccl_destroy, LIB_RCCL), Cvoid, (Ptr{Cvoid},), x.drv); x.drv = C_NULL), sf)
    return sf
end

function check_rccl(code::Int32, sf::ShardedMI355X)
    code == 0 && return nothing
    error("gmrfx_rccl ($code): " * unsafe_string(ccall((:gmrfx_rccl_last_error, LIB_RCCL), Cstring, (Ptr{Cvoid},), sf.drv)))
end

# log det Q over all ranks + the pivot report (PosDefException like refactorize! under check_posdef)
function LinearAlgebra.logdet(sf::ShardedMI355X)
    ld = Ref{Float64}(0.0); info = Ref{Int64}(0)
    check_rccl(ccall((:gmrfx_rccl_logdet, LIB_RCCL), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Int64}), sf.drv, ld, info), sf)
    info[] > 0 && throw(PosDefException(Int(info[])))
    return ld[]
end

# diag(Q^-1) on every rank (the sharded Takahashi recursion + an all-reduce)
function selinv_diag(sf::ShardedMI355X)
    out = Vector{Float64}(undef, sf.n)
    check_rccl(ccall((:gmrfx_rccl_selinv_diag, LIB_RCCL), Int32, (Ptr{Cvoid}, Ptr{Float64}), sf.drv, out), sf)
    return out
end

# the rows of B this rank reads in solve! (the ranks' masks partition 1:n: B may be row-sharded)
function needed_rows(sf::ShardedMI355X)
    mask = Vector{UInt8}(undef, sf.n)
    check_rccl(ccall((:gmrfx_rccl_needed_rows, LIB_RCCL), Int32, (Ptr{Cvoid}, Ptr{UInt8}), sf.drv, mask), sf)
    return mask .!= 0
end

function solve! end       # ROCArray methods: julia/ext/GMRFXAMDGPUExt.jl

# ---------------------------------------------------------------------------------- seam A
# LinearSolve algorithm; GMRF-side hooks exactly as the Pardiso extension (ext/GaussianMarkovRandomFieldsPardiso.jl:10-80).
import LinearSolve, SciMLBase
struct MI355XCholesky <: LinearSolve.AbstractFactorization
    ordering::Any
end
MI355XCholesky() = MI355XCholesky(nothing)

# `LinearCache`'s `cacheval` field is parameterised on the type `init_cacheval` returns, so it must be a concrete
# MUTABLE holder from the start (returning `nothing` and assigning a backend later is a `convert` error). The
# backend handle is created by the first `solve!`; `colptr` / `rowval` remember the pattern it was analysed for.
mutable struct MI355XCacheval
    be::Union{Nothing, MI355XBackend}
    colptr::Vector{Int}
    rowval::Vector{Int}
end
LinearSolve.init_cacheval(::MI355XCholesky, A, b, u, Pl, Pr, maxiters, abstol, reltol, verbose, assumptions) =
    MI355XCacheval(nothing, Int[], Int[])
# deepcopy(cache) (gaussian_approximation.jl:103-109): the default field-wise deepcopy of the holder reaches
# `deepcopy_internal(::MI355XBackend)` above, i.e. gmrfx_clone -- the factor survives the fork.

function SciMLBase.solve!(cache::LinearSolve.LinearCache, alg::MI355XCholesky; kwargs...)
    A = cache.A isa Symmetric ? parent(cache.A) : cache.A
    cv = LinearSolve.@get_cacheval(cache, :MI355XCholesky)::MI355XCacheval
    if cache.isfresh
        cp, rv = SparseArrays.getcolptr(A), rowvals(A)
        if cv.be === nothing || cv.colptr != cp || cv.rowval != rv        # first solve!, or cache.A got a new pattern
            h = create(A; ordering = alg.ordering, check_posdef = true)     # seam A throws PosDefException
            cv.be = MI355XBackend(h, size(A, 1), nothing, nothing)
            cv.colptr, cv.rowval = copy(cp), copy(rv)
        end
        refactorize!(cv.be, Symmetric(A))
        cache.isfresh = false
    end
    cache.u .= backend_solve(cv.be::MI355XBackend, cache.b)
    return SciMLBase.build_linear_solution(alg, cache.u, nothing, cache)
end
_be(cache) = (LinearSolve.@get_cacheval(cache, :MI355XCholesky)::MI355XCacheval).be::MI355XBackend
G.supports_selinv(::MI355XCholesky) = Val{true}()
G.supports_backward_solve(::MI355XCholesky) = Val{true}()
G._selinv_diag_impl(cache, ::MI355XCholesky) = get_selinv_diag(_be(cache))
G._selinv_impl(cache, ::MI355XCholesky) = Symmetric(get_selinv(_be(cache)))
G._backward_solve_impl(cache, x, ::MI355XCholesky) = backend_backward_solve(_be(cache), x)
G._backward_solve_impl(cache, Z::Matrix{Float64}, ::MI355XCholesky) = backend_backward_solve(_be(cache), Z)
# seam-A twin of the batched sampler above (src/gmrf.jl:271-281 draws the columns one by one). Restricted BY DISPATCH to GMRFs whose
# LinearSolve cache carries this plug-in's algorithm: `GMRF` has its cache type as its sixth parameter (src/gmrf.jl:144-156) and
# `LinearSolve.LinearCache{TA, Tb, Tu, Tp, Talg, ...}` its algorithm type as its fifth (LinearSolve 2 and 3, the compat range of
# the reference's Project.toml; checked when the module loads). Every other GMRF keeps Distributions' own matrix method -- the
# plug-in owns a type in the signature (`MI355XCholesky`), so this is no piracy on Distributions / GaussianMarkovRandomFields.
let body = Base.unwrap_unionall(LinearSolve.LinearCache)
    body.parameters[5] === fieldtype(body, :alg) ||
        error("GMRFX: LinearSolve.LinearCache no longer has its algorithm as fifth type parameter; adjust MI355XLinearCache")
end
const MI355XLinearCache = LinearSolve.LinearCache{<:Any, <:Any, <:Any, <:Any, MI355XCholesky}
const MI355XGMRF = G.GMRF{<:Any, <:Any, <:Any, <:Any, <:Any, <:MI355XLinearCache}
function Distributions._rand!(rng::AbstractRNG, d::MI355XGMRF, X::AbstractMatrix{<:Real})
    Z = randn!(rng, Matrix{Float64}(undef, size(X, 1), size(X, 2)))
    X .= G.backward_solve(d.linsolve_cache, Z) .+ d.mean
    return X
end
G._logdet_cov_impl(cache, ::MI355XCholesky) = -compute_logdet(_be(cache))       # note the sign (logdet.jl:30)
G.prepare_for_linsolve(A::SparseMatrixCSC, ::MI355XCholesky) = Symmetric(A)
G.configure_algorithm(alg::MI355XCholesky) = alg
# Val{true}() / Val{false}() like every method of this predicate (linsolve_utils.jl:49-56): `_resolve_linsolve`
# dispatches on the Val; dense or SymTridiagonal storage falls back to LinearSolve's default
G.algorithm_applicable(::MI355XCholesky, ::Union{SparseMatrixCSC, Symmetric{<:Any, <:SparseMatrixCSC}}) = Val{true}()
G.algorithm_applicable(::MI355XCholesky, ::AbstractMatrix) = Val{false}()

export MI355XBackend, MI355XCholesky, MI355XCacheval, ordering_permutation, ShardedMI355X, rccl_unique_id
end # module
