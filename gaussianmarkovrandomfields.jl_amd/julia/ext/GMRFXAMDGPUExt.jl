# GMRFXAMDGPUExt.jl -- package extension of GMRFX (weak dependency AMDGPU.jl): the device-resident entry points of
# include/gmrfx.h (`*_dev`: operands already in HBM, no PCIe traffic inside the call) for ROCArray arguments.
#
# Project.toml of the plug-in package:
#     [weakdeps]
#     AMDGPU = "21141c5a-9bdb-4563-92ae-f87d6854732e"
#     [extensions]
#     GMRFXAMDGPUExt = "AMDGPU"
#
# Layout: a ROCMatrix{Float64} of size n x k is column-major with leading dimension stride(A, 2) -- exactly the `ldb` / `ldx`
# convention of the C ABI. The library runs on its own streams and returns when the result is complete; AMDGPU.jl's
# task-local stream is synchronised before the call so that the operands are.
# Reference calls replaced: src/workspace/backend.jl:165-189 (refactorize!), :191-209 (backend_solve), :281-284
# (backend_backward_solve), src/workspace/workspace_gmrf.jl:288-292 (logpdf).
# NOTE: like GMRFX.jl this file cannot be run in this image (no Julia); tests/test_julia_shim_signatures.py checks every
# ccall tuple below against include/gmrfx.h.
module GMRFXAMDGPUExt

using AMDGPU
using LinearAlgebra
import GMRFX
import GMRFX: MI355XBackend, LIB, check, refactorize_solve!, backend_solve!, backend_backward_solve!, logpdf_terms
import GMRFX: ShardedMI355X, LIB_RCCL, check_rccl, solve!
import GaussianMarkovRandomFields: refactorize!

devptr(A::ROCArray{Float64}) = reinterpret(Ptr{Float64}, pointer(A))

function _invalidate!(b::MI355XBackend)
    b.selinv_cache = nothing
    b.selinv_diag_cache = nothing
    return nothing
end

# numeric refactorisation from values resident in HBM (pattern order of the Q the backend was built from)
function refactorize!(b::MI355XBackend, d_nz::ROCVector{Float64})
    AMDGPU.synchronize()
    info = Ref{Int64}(0)
    GC.@preserve d_nz check(ccall((:gmrfx_refactorize_dev, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int64}),
        b.h.ptr, devptr(d_nz), info), b.h)
    _invalidate!(b)
    return info[]                                   # 0, or the elimination step of the first non-positive pivot
end

# workspace_solve on a stale factorisation, everything in HBM: the call bench.py times (gmrfx_refactorize_solve_dev)
function refactorize_solve!(X::ROCMatrix{Float64}, b::MI355XBackend, d_nz::ROCVector{Float64}, B::ROCMatrix{Float64})
    size(B, 1) == b.n && size(X) == size(B) || throw(DimensionMismatch("B / X must be n x k"))
    AMDGPU.synchronize()
    info = Ref{Int64}(0)
    GC.@preserve d_nz B X check(ccall((:gmrfx_refactorize_solve_dev, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ref{Int64}),
        b.h.ptr, devptr(d_nz), devptr(B), stride(B, 2), size(B, 2), devptr(X), stride(X, 2), info), b.h)
    _invalidate!(b)
    info[] > 0 && throw(PosDefException(Int(info[])))      # (as `b.factor \ rhs` does on a failed CHOLMOD factor: backend.jl:178-193)
    return X
end

function backend_solve!(X::ROCMatrix{Float64}, b::MI355XBackend, B::ROCMatrix{Float64})
    size(B, 1) == b.n && size(X) == size(B) || throw(DimensionMismatch("B / X must be n x k"))
    AMDGPU.synchronize()
    GC.@preserve B X check(ccall((:gmrfx_solve_dev, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
        b.h.ptr, devptr(B), stride(B, 2), size(B, 2), devptr(X), stride(X, 2)), b.h)
    return X
end

# k samples x = P' L^-T z in one sweep (src/gmrf.jl:271-281 draws them one by one)
function backend_backward_solve!(X::ROCMatrix{Float64}, b::MI355XBackend, Z::ROCMatrix{Float64})
    size(Z, 1) == b.n && size(X) == size(Z) || throw(DimensionMismatch("Z / X must be n x k"))
    AMDGPU.synchronize()
    GC.@preserve Z X check(ccall((:gmrfx_backward_solve_dev, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
        b.h.ptr, devptr(Z), stride(Z, 2), size(Z, 2), devptr(X), stride(X, 2)), b.h)
    return X
end

# one evaluation of the hyper-parameter loop: new values -> factorisation, (x_k - mean)' Q (x_k - mean) for every column of X and
# log det Q in one call; logpdf_k = -q_k / 2 + logdet / 2 - n log(2 pi) / 2 (workspace_gmrf.jl:288-292)
function logpdf_terms(b::MI355XBackend, d_nz::ROCVector{Float64}, X::ROCMatrix{Float64}; mean::Union{Nothing, ROCVector{Float64}} = nothing)
    size(X, 1) == b.n || throw(DimensionMismatch("X must be n x k"))
    AMDGPU.synchronize()
    quad = Vector{Float64}(undef, size(X, 2))
    ld = Ref{Float64}(0.0)
    info = Ref{Int64}(0)
    mu = mean === nothing ? Ptr{Float64}(C_NULL) : devptr(mean)
    GC.@preserve d_nz X mean quad check(ccall((:gmrfx_refactorize_logpdf_dev, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int64}),
        b.h.ptr, devptr(d_nz), devptr(X), stride(X, 2), size(X, 2), mu, quad, ld, info), b.h)
    _invalidate!(b)
    return quad, ld[]
end


# ---- one factorisation over several GPUs: the native RCCL driver (libgmrfx_rccl.so) with operands resident in HBM ----
function refactorize!(sf::ShardedMI355X, d_nz::ROCVector{Float64})
    AMDGPU.synchronize()
    GC.@preserve d_nz check_rccl(ccall((:gmrfx_rccl_refactorize, LIB_RCCL), Int32, (Ptr{Cvoid}, Ptr{Float64}), sf.drv, devptr(d_nz)), sf)
    return sf
end

# Q X = B over all ranks; gather = true: all of X on rank 0, false: every rank keeps the rows it owns (+ every top front's)
function solve!(X::ROCMatrix{Float64}, sf::ShardedMI355X, B::ROCMatrix{Float64}; gather::Bool = true)
    size(B, 1) == sf.n && size(X) == size(B) || throw(DimensionMismatch("B / X must be n x k"))
    AMDGPU.synchronize()
    GC.@preserve B X check_rccl(ccall((:gmrfx_rccl_solve, LIB_RCCL), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int32),
        sf.drv, devptr(B), stride(B, 2), size(B, 2), devptr(X), stride(X, 2), gather ? 1 : 0), sf)
    return X
end

# X = P' L^-T Z over all ranks (samples: Z in elimination order, as CHOLMOD's F.UP \ z takes it)
function backend_backward_solve!(X::ROCMatrix{Float64}, sf::ShardedMI355X, Z::ROCMatrix{Float64}; gather::Bool = true)
    size(Z, 1) == sf.n && size(X) == size(Z) || throw(DimensionMismatch("Z / X must be n x k"))
    AMDGPU.synchronize()
    GC.@preserve Z X check_rccl(ccall((:gmrfx_rccl_backward_solve, LIB_RCCL), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Int32),
        sf.drv, devptr(Z), stride(Z, 2), size(Z, 2), devptr(X), stride(X, 2), gather ? 1 : 0), sf)
    return X
end

end # module
