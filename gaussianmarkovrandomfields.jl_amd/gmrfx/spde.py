"""Synthetic Matérn-SPDE precision generator (inputs of the hot path, not the hot path).

Reproduces, for P1 simplices on structured meshes, the formulas of the reference's FEM
extension so that the benchmark Q *is* the matrix a reference user would hand to the solver:

* ``ν = smoothness+1`` (even d) / ``smoothness+½`` (odd d); ``κ = √(8ν)/range``; ``α = ν+d/2``
  (ext/GaussianMarkovRandomFieldsFEM/matern_spde.jl:415-422, 340-343)
* element stiffness ``G_e[i,j] = |T| ∇φ_i·∇φ_j``, lumped mass ``C_ii = Σ_{T∋i} |T|/(d+1)``
  (fem_utils.jl:6-8, 42-110; matern_spde.jl:53-78)
* ``K = κ²C + G`` (matern_spde.jl:346); ``ratio = Γ(ν)/(Γ(ν+d/2)(4π)^{d/2}κ^{2ν})``
  (matern_spde.jl:349-353)
* α=1: ``Q = ratio·K``; α=2: ``Q = Kᵀ(ratio·C⁻¹)K``; α≥3: ``Q = Kᵀ(ratio·C⁻¹ Q_{α-2} C⁻¹)K``
  with the inner recursion unscaled (matern_spde.jl:177-231)
* MaternModel scatters ``τ·Q`` into the κ-invariant structural pattern ``P_α`` of
  ``S = I ∪ pattern(G)`` — explicit zeros are stored (matern_spde.jl:248-265,
  matern_model.jl:109-121).

No boundary conditions (empty constraint handler ⇒ natural/Neumann boundary).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp


@dataclass
class Mesh:
    points: np.ndarray   # (n, d) float64
    cells: np.ndarray    # (m, d+1) int64, P1 simplices
    shape: tuple         # nodes per axis

    @property
    def n(self) -> int:
        return self.points.shape[0]

    @property
    def dim(self) -> int:
        return self.points.shape[1]


def grid_mesh_2d(nx: int, ny: int, lo=-1.0, hi=1.0, jitter: float = 0.0, seed: int = 0) -> Mesh:
    """nx × ny *nodes* on [lo,hi]², each quad split into two triangles along the (1,1)
    diagonal (the split `generate_grid(Triangle, …)` uses).  Node id = ix + nx*iy.
    ``jitter`` moves interior nodes by U(-j·h, j·h) per axis (SURVEY §8d cfg 2: j=0.25)."""
    xs = np.linspace(lo, hi, nx)
    ys = np.linspace(lo, hi, ny)
    X, Y = np.meshgrid(xs, ys, indexing="xy")  # Y rows, X cols -> id = ix + nx*iy
    pts = np.stack([X.ravel(), Y.ravel()], axis=1)
    if jitter > 0.0:
        rng = np.random.default_rng(seed)
        hx = (hi - lo) / (nx - 1)
        hy = (hi - lo) / (ny - 1)
        d = rng.uniform(-jitter, jitter, size=pts.shape) * np.array([hx, hy])
        ix = np.arange(nx * ny) % nx
        iy = np.arange(nx * ny) // nx
        interior = (ix > 0) & (ix < nx - 1) & (iy > 0) & (iy < ny - 1)
        pts[interior] += d[interior]
    ix, iy = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="xy")
    v00 = (ix + nx * iy).ravel()
    v10 = v00 + 1
    v01 = v00 + nx
    v11 = v01 + 1
    t1 = np.stack([v00, v10, v11], axis=1)
    t2 = np.stack([v00, v11, v01], axis=1)
    cells = np.concatenate([t1, t2], axis=0).astype(np.int64)
    return Mesh(pts, cells, (nx, ny))


_KUHN = np.array([[0, 1, 3, 7], [0, 1, 5, 7], [0, 2, 3, 7], [0, 2, 6, 7], [0, 4, 5, 7], [0, 4, 6, 7]])


def grid_mesh_3d(nx: int, ny: int, nz: int, lo=-1.0, hi=1.0) -> Mesh:
    """nx × ny × nz nodes, every cube split into 6 Kuhn tetrahedra sharing the main
    diagonal (SURVEY §8d cfg 4). Node id = ix + nx*(iy + ny*iz)."""
    xs, ys, zs = (np.linspace(lo, hi, k) for k in (nx, ny, nz))
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    iz, iy, ix = np.meshgrid(np.arange(nz - 1), np.arange(ny - 1), np.arange(nx - 1), indexing="ij")
    base = (ix + nx * (iy + ny * iz)).ravel()
    # corner c = bx + 2*by + 4*bz
    off = np.array([bx + nx * (by + ny * bz) for bz in (0, 1) for by in (0, 1) for bx in (0, 1)])
    corners = base[:, None] + off[None, :]
    cells = np.concatenate([corners[:, k] for k in _KUHN], axis=0).astype(np.int64)
    return Mesh(pts, cells, (nx, ny, nz))


def assemble_C_G(mesh: Mesh):
    """Lumped mass C (diagonal, as a vector) and stiffness G (CSR) for P1 simplices."""
    P, T = mesh.points, mesh.cells
    d = mesh.dim
    n = mesh.n
    V = P[T]                                    # (m, d+1, d)
    E = V[:, 1:, :] - V[:, :1, :]               # (m, d, d) edge matrix rows = v_k - v_0
    det = np.linalg.det(E)
    vol = np.abs(det) / math.factorial(d)
    Einv = np.linalg.inv(E)                     # columns = gradients of φ_1..φ_d
    grads = np.empty((T.shape[0], d + 1, d))
    grads[:, 1:, :] = np.transpose(Einv, (0, 2, 1))
    grads[:, 0, :] = -grads[:, 1:, :].sum(axis=1)
    Ge = vol[:, None, None] * np.einsum("mid,mjd->mij", grads, grads)
    I = np.repeat(T[:, :, None], d + 1, axis=2).ravel()
    J = np.repeat(T[:, None, :], d + 1, axis=1).ravel()
    G = sp.coo_matrix((Ge.ravel(), (I, J)), shape=(n, n)).tocsr()
    G.sum_duplicates()
    C = np.bincount(T.ravel(), weights=np.repeat(vol / (d + 1), d + 1), minlength=n)
    return C, G


def smoothness_to_nu(smoothness: int, d: int) -> float:
    if smoothness < 0:
        raise ValueError("smoothness must be non-negative")
    return smoothness + 1.0 if d % 2 == 0 else smoothness + 0.5


def range_to_kappa(range_: float, nu: float) -> float:
    return math.sqrt(8.0 * nu) / range_


def _pattern_bool(M: sp.spmatrix) -> sp.csr_matrix:
    P = M.tocsr().copy()
    P.data = np.ones_like(P.data, dtype=np.float64)
    return P


def structural_pattern(G: sp.csr_matrix, alpha: int) -> sp.csc_matrix:
    """P_1 = S, P_2 = SᵀS, P_α = Sᵀ P_{α-2} S with S = I ∪ pattern(G); all-ones values
    (no cancellation possible since entries are positive counts)."""
    n = G.shape[0]
    # pattern(G) = mesh connectivity incl. entries whose value cancels to 0.0 (Ferrite's
    # allocate_matrix pattern, matern_spde.jl:59) -- take the structure, never the values.
    S = _pattern_bool(_pattern_bool(G) + sp.identity(n, format="csr"))
    if alpha == 1:
        P = S
    elif alpha == 2:
        P = (S.T @ S)
    else:
        P = S.T @ structural_pattern(G, alpha - 2).tocsr() @ S
    P = P.tocsc()
    P.data[:] = 1.0
    P.sort_indices()
    return P


def matern_precision(mesh: Mesh, smoothness: int = 0, range_: float = 0.2, tau: float = 1.0,
                     sigma2: float = 1.0):
    """τ·Q on the structural pattern, as CSC with int64 indices, both triangles stored
    (what `precision_matrix(::MaternModel; τ, range)` returns under `Symmetric`)."""
    d = mesh.dim
    nu = smoothness_to_nu(smoothness, d)
    alpha2 = 2 * nu + d
    if abs(alpha2 / 2 - round(alpha2 / 2)) > 1e-12:
        raise ValueError(f"α = ν + d/2 = {alpha2/2} is not an integer (reference throws InexactError)")
    alpha = int(round(alpha2 / 2))
    kappa = range_to_kappa(range_, nu)
    C, G = assemble_C_G(mesh)
    n = mesh.n
    K = (sp.diags(kappa**2 * C) + G).tocsr()
    ratio = math.gamma(nu) / (math.gamma(nu + d / 2) * (4 * math.pi) ** (d / 2) * kappa ** (2 * nu)) / sigma2
    Cinv = sp.diags(1.0 / C)

    def rec(a, scale):
        if a == 1:
            return (scale * K).tocsr()
        if a == 2:
            rhs = Cinv
        else:
            rhs = Cinv @ rec(a - 2, 1.0) @ Cinv
        return (K.T @ (scale * rhs) @ K).tocsr()

    Q = rec(alpha, ratio).tocsc()
    Q.sort_indices()
    # scatter τ·Q into the structural pattern (explicit zeros kept)
    return _scatter_into_pattern(Q, tau, structural_pattern(G, alpha))


def _scatter_into_pattern(Q: sp.csc_matrix, tau: float, P: sp.csc_matrix) -> sp.csc_matrix:
    n = Q.shape[0]
    out = np.zeros(P.nnz, dtype=np.float64)
    # both are column-sorted CSC; pattern(Q) ⊆ pattern(P)
    pcol = np.repeat(np.arange(n, dtype=np.int64), np.diff(P.indptr))
    qcol = np.repeat(np.arange(n, dtype=np.int64), np.diff(Q.indptr))
    pkey = pcol * n + P.indices.astype(np.int64)
    qkey = qcol * n + Q.indices.astype(np.int64)
    pos = np.searchsorted(pkey, qkey)
    if not np.array_equal(pkey[pos], qkey):
        raise AssertionError("pattern(Q) is not contained in the structural pattern")
    out[pos] = tau * Q.data
    R = sp.csc_matrix((out, P.indices.astype(np.int64), P.indptr.astype(np.int64)), shape=(n, n))
    return R


def ar1_precision(T: int, rho: float = 0.9, tau: float = 1.0) -> sp.csc_matrix:
    """AR(1) precision (src/latent_models/ar.jl:135-148): diag [τ,(1+ρ²)τ,…,τ], off −ρτ."""
    main = np.full(T, (1 + rho * rho) * tau)
    main[0] = main[-1] = tau
    off = np.full(T - 1, -rho * tau)
    return sp.diags([off, main, off], [-1, 0, 1], format="csc")


def random_spd_precision(n: int, density: float = 0.3, seed: int = 42) -> sp.csc_matrix:
    """Equivalent of the reference's test fixture `_make_test_precision`
    (test/workspace/test_gmrf_workspace.jl:8-13): Q = S Sᵀ + n I, S = sprand(n,n,density).
    Julia's MersenneTwister stream is not reproducible here; own seed."""
    rng = np.random.default_rng(seed)
    S = sp.random(n, n, density=density, random_state=rng, format="csr")
    Q = (S @ S.T + n * sp.identity(n)).tocsc()
    Q.sort_indices()
    Q.indices = Q.indices.astype(np.int64)
    Q.indptr = Q.indptr.astype(np.int64)
    return Q


def tall_front_precision(nblocks: int = 6, bw: int = 8, sep: int = 150, part: int = 30, seed: int = 7):
    """A test precision whose natural order gives TALL, NARROW fronts: nblocks pairs of dense bw-node blocks (A_i, B_i); A_i is
    coupled to B_i and to `part` nodes of a dense `sep`-node separator, B_i to the whole separator. In the order A_1, B_1, ...,
    A_n, B_n, separator every B_i is a front of bw columns and bw + sep rows (more than 128 rows below its diagonal block: the
    long-chunk path of the sweep tasks). Returns (Q, natural ordering keyword)."""
    rng = np.random.default_rng(seed)
    n = 2 * nblocks * bw + sep
    A = sp.lil_matrix((n, n))
    s0 = 2 * nblocks * bw
    for i in range(nblocks):
        a0, b0 = 2 * i * bw, (2 * i + 1) * bw
        A[a0:a0 + bw, a0:a0 + bw] = rng.uniform(-1, 1, (bw, bw))
        A[b0:b0 + bw, b0:b0 + bw] = rng.uniform(-1, 1, (bw, bw))
        A[b0:b0 + bw, a0:a0 + bw] = rng.uniform(-1, 1, (bw, bw))
        cols = s0 + rng.choice(sep, part, replace=False)
        for cidx in cols:
            A[cidx, a0:a0 + bw] = rng.uniform(-1, 1, bw)
        A[s0:s0 + sep, b0:b0 + bw] = rng.uniform(-1, 1, (sep, bw))
    A[s0:, s0:] = rng.uniform(-1, 1, (sep, sep))
    A = sp.csc_matrix(A)
    M = sp.tril(A) + sp.tril(A, -1).T
    Q = (M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 1.0)).tocsc()
    Q.sort_indices()
    Q.indices = Q.indices.astype(np.int64)
    Q.indptr = Q.indptr.astype(np.int64)
    return Q
