"""Space-time precisions (BASELINE config 5): Q = kron(Q_t, Q_s) (+ a likelihood term) handed WHOLE to the
sparse Cholesky, exactly as the reference does (`SeparableModel.precision_matrix` folds `kron` over the components and
the joint matrix goes to CHOLMOD, src/latent_models/separable.jl:143-172; the implicit-Euler state-space prior of
ext/GaussianMarkovRandomFieldsFEM/linear_ssm.jl:88-100 is the same block-tridiagonal shape).

For a tridiagonal Q_t (AR(1), RW1) the joint precision is block tridiagonal in time. Instead of a sequential block
(Kalman-style) elimination -- T dependent steps with n_s x n_s dense blocks -- the factorisation here is the general
multifrontal one on a NESTED DISSECTION OF THE SPACE-TIME GRAPH: node (t, i) gets the coordinates (x_i, y_i, t * dt), the
geometric dissection cuts time slabs and space alike, and all slabs factor concurrently. The prior alone never needs
this (gmrfx.KroneckerWorkspace answers it from two factor-scale factorisations); a posterior Q_prior + diag(h) does."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def spacetime_precision(Qt, Qs, obs_diag=None) -> sp.csc_matrix:
    """kron(Q_t, Q_s) [+ diag(obs_diag)] as CSC with int64 indices; rightmost factor (space) varies fastest
    (x[t * n_s + i], separable.jl:17, 41)."""
    Q = sp.kron(sp.csc_matrix(Qt), sp.csc_matrix(Qs), format="csc")
    if obs_diag is not None:
        Q = (Q + sp.diags(np.asarray(obs_diag, dtype=np.float64))).tocsc()
    Q.sort_indices()
    Q.indices = Q.indices.astype(np.int64)
    Q.indptr = Q.indptr.astype(np.int64)
    return Q


def spacetime_coords(points, T: int, dt: float | None = None) -> np.ndarray:
    """(T n_s) x (d + 1) coordinates for the geometric nested dissection of the space-time graph. dt defaults to the
    mean spatial node spacing, so that a time step and a mesh edge weigh the same in the dissection."""
    P = np.asarray(points, dtype=np.float64)
    ns, d = P.shape
    if dt is None:
        ext = P.max(axis=0) - P.min(axis=0)
        dt = float(np.prod(ext) / ns) ** (1.0 / d)
    out = np.empty((T * ns, d + 1))
    out[:, :d] = np.tile(P, (T, 1))
    out[:, d] = np.repeat(np.arange(T) * dt, ns)
    return out
