# cholmod_baseline.jl -- times the reference's own CPU hot path (the Julia-stdlib CHOLMOD calls of
# src/workspace/backend.jl:148-149, 184, 192, 208, 212, 283) on the SAME Q and the SAME permutation as the GPU
# bench. Needs only the Julia standard library. bench.py probes `julia` at run time and runs
#     julia -t auto bench/cholmod_baseline.jl <dir>
# where <dir> holds raw little-endian files written by bench.py: meta.bin (Int64: n, nnz, nrhs), colptr.bin,
# rowval.bin (Int64, 1-based), nzval.bin (Float64), perm.bin (Int64, 1-based), B.bin (Float64, n x nrhs col-major).
# Prints one JSON object. (No Julia in the authoring image: written from the API docs, never executed there.)
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
println("{\"kind\": \"reference\", \"julia\": \"$(VERSION)\", \"threads\": $(Sys.CPU_THREADS), \"blas_threads\": $(BLAS.get_num_threads()), ",
    "\"n\": $n, \"nnz_L\": $(nnz(F)), \"s_symbolic_plus_first_factor\": $t_sym, \"s_refactorize\": $t_fac, \"s_solve\": $t_sol, ",
    "\"s_logdet\": $t_ld, \"s_backward_solve_1\": $t_up, \"dof_per_s\": $(n / (t_fac + t_sol)), \"logdet\": $ld, \"rel_residual\": $res, ",
    "\"amd_default\": {\"nnz_L\": $(nnz(Famd)), \"s_analyse_plus_factor\": $t_amd, \"s_refactorize\": $t_fac_amd}}")
