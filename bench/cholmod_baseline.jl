# cholmod_baseline.jl -- times the reference's own CPU hot path (the Julia-stdlib CHOLMOD calls of
# src/workspace/backend.jl:148-149, 184, 192, 208, 212, 283) on the SAME Q and the SAME permutation as the GPU
# bench. Needs only the Julia standard library. bench.py probes `julia` at run time and runs
#     julia -t auto bench/cholmod_baseline.jl <dir>
# where <dir> holds raw little-endian files written by bench.py: meta.bin (Int64: n, nnz, nrhs), colptr.bin,
# rowval.bin (Int64, 1-based), nzval.bin (Float64), perm.bin (Int64, 1-based), B.bin (Float64, n x nrhs col-major).
# Prints one JSON object. (No Julia in the authoring image: written from the API docs, never executed there.)
#
# PARITY VECTORS (the one route to an oracle pinned by the reference's own arithmetic, SURVEY 8c): with a second argument
#     julia -t auto bench/cholmod_baseline.jl <dir> <outdir>
# the script also WRITES what the reference's calls return on this Q / permutation -- raw little-endian files in <outdir>:
#   scalars.bin  Float64 x 4: logdet(F) (backend.jl:212), nnz(F) (as Float64), n, k = number of solved columns
#   p.bin        Int64 n     the permutation CHOLMOD really used (F.p, 1-based; CHOLMOD may postorder the given one)
#   X.bin        Float64 n x k   F \ B[:, 1:k]  (backend.jl:208), k = min(nrhs, 4)
#   UPz.bin      Float64 n       F.UP \ B[:, 1]  (backend.jl:283)
#   Lcolptr.bin / Lrowval.bin (Int64, 1-based) / Lnzval.bin (Float64): sparse(F.L) -- unique for LL' with a positive
#                diagonal, so the oracle and the HIP path factoring P Q P' with p.bin must reproduce it entry by entry
#   selinv_diag.bin Float64 n   SelectedInversion.selinv_diag(F) (backend.jl:253) -- only if SelectedInversion is installed
# tests/test_cholmod_parity.py compares the CPU oracle and the HIP path with these files (cases: tests/golden/cholmod_inputs/*,
# written by tests/golden/make_cholmod_inputs.py; outputs go to tests/golden/cholmod_outputs/<case>/) and SKIPS, saying so,
# while they are absent. For n > 200 000 only scalars.bin and the first 4096 rows of X / UPz are written (bench-sized runs).
using LinearAlgebra, SparseArrays
dir = ARGS[1]
rd(T, name, k) = (a = Vector{T}(undef, k); read!(joinpath(dir, name), a); a)
n, nz, nrhs = rd(Int64, "meta.bin", 3)
Q = SparseMatrixCSC(n, n, rd(Int64, "colptr.bin", n + 1), rd(Int64, "rowval.bin", nz), rd(Float64, "nzval.bin", nz))
perm = rd(Int64, "perm.bin", n)
B = reshape(rd(Float64, "B.bin", n * nrhs), n, nrhs)
S = Symmetric(Q)
BLAS.set_num_threads(Sys.CPU_THREADS)
t_sym = @elapsed F = cholesky(S; perm = perm)                        # backend.jl:148-149 (analyse + first numeric)
t_fac = minimum(@elapsed(cholesky!(F, S; check = false)) for _ in 1:3)   # backend.jl:184 refactorize!
t_sol = minimum(@elapsed(F \ B) for _ in 1:2)                        # backend.jl:208 blocked multi-RHS solve
X = F \ B
t_ld = @elapsed ld = logdet(F)                                       # backend.jl:212
t_up = @elapsed F.UP \ B[:, 1]                                       # backend.jl:283 (one sample)
t_amd = @elapsed Famd = cholesky(S)                                  # what an unmodified user gets: CHOLMOD's own AMD
t_fac_amd = minimum(@elapsed(cholesky!(Famd, S; check = false)) for _ in 1:2)
res = norm(Q * X - B) / norm(B)
if length(ARGS) >= 2
    out = ARGS[2]
    mkpath(out)
    wr(name, a) = open(io -> write(io, a), joinpath(out, name), "w")
    k = min(nrhs, 4)
    rows = n > 200_000 ? (1:4096) : (1:n)
    wr("scalars.bin", Float64[ld, nnz(F), n, k])
    wr("p.bin", Vector{Int64}(F.p))
    wr("X.bin", Matrix{Float64}(X[rows, 1:k]))
    wr("UPz.bin", Vector{Float64}((F.UP \ B[:, 1])[rows]))
    if n <= 200_000
        Ls = sparse(F.L)
        wr("Lcolptr.bin", Vector{Int64}(SparseArrays.getcolptr(Ls)))
        wr("Lrowval.bin", Vector{Int64}(rowvals(Ls)))
        wr("Lnzval.bin", Vector{Float64}(nonzeros(Ls)))
        try
            @eval using SelectedInversion
            wr("selinv_diag.bin", Vector{Float64}(Base.invokelatest(SelectedInversion.selinv_diag, F)))
        catch err
            @warn "SelectedInversion not available: selinv_diag.bin not written" err
        end
    end
end
println("{\"kind\": \"reference\", \"julia\": \"$(VERSION)\", \"threads\": $(Sys.CPU_THREADS), \"blas_threads\": $(BLAS.get_num_threads()), ",
    "\"n\": $n, \"nnz_L\": $(nnz(F)), \"s_symbolic_plus_first_factor\": $t_sym, \"s_refactorize\": $t_fac, \"s_solve\": $t_sol, ",
    "\"s_logdet\": $t_ld, \"s_backward_solve_1\": $t_up, \"dof_per_s\": $(n / (t_fac + t_sol)), \"logdet\": $ld, \"rel_residual\": $res, ",
    "\"amd_default\": {\"nnz_L\": $(nnz(Famd)), \"s_analyse_plus_factor\": $t_amd, \"s_refactorize\": $t_fac_amd}}")
