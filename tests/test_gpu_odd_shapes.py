"""Odd-sized medium meshes (fronts of a few hundred rows with odd row / column counts, odd numbers of right-hand sides):
the kernels load operands 16 bytes at a time (row pairs, right-hand-side pairs, index pairs), so the last row, the last
column and the last right-hand side of everything are the interesting ones. Checked against identities that need no
oracle at these sizes: residuals, logdet scaling, the selected inverse's diagonal against unit-vector solves, the sample
identity z'z = x'Qx of the backward solve."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
pytestmark = pytest.mark.gpu


def _meshes():
    from gmrfx import spde
    yield "2d_137x211", spde.grid_mesh_2d(137, 211, jitter=0.2, seed=5), 0.15
    yield "2d_301x97", spde.grid_mesh_2d(301, 97, jitter=0.3, seed=6), 0.3
    yield "3d_23x17x19", spde.grid_mesh_3d(23, 17, 19), 0.5
    yield "3d_31x29x7", spde.grid_mesh_3d(31, 29, 7), 0.4


@pytest.mark.parametrize("case", ["2d_137x211", "2d_301x97", "3d_23x17x19", "3d_31x29x7"])
def test_odd_shapes(case):
    import gmrfx
    from gmrfx import spde
    name, mesh, rng_ = next(t for t in _meshes() if t[0] == case)
    Q = sp.csc_matrix(spde.matern_precision(mesh, 0, rng_))
    n = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, coords=mesh.points)
    assert be.last_info == 0
    rng = np.random.default_rng(11)
    qnorm = abs(Q).sum(axis=0).max()
    for nrhs in (1, 7, 63, 65, 129):
        B = rng.standard_normal((n, nrhs))
        X = be.backend_solve(B)
        # normwise backward error (the jittered meshes are not all well conditioned: |X| can be 20 x |B|, and the
        # oracle's own residual is then 1e-8 as well)
        eta = np.linalg.norm(Q @ X - B) / (qnorm * np.linalg.norm(X) + np.linalg.norm(B))
        assert eta < 1e-13, (case, nrhs, eta)
        Z = B[:, :min(nrhs, 33)]
        Xs = be.backend_backward_solve(Z)                       # x = P' L^-T z: x'Qx = z'z column by column
        assert np.allclose(np.einsum("ij,ij->j", Xs, Q @ Xs), np.einsum("ij,ij->j", Z, Z), rtol=1e-8)
    ld = be.compute_logdet()
    d = be.get_selinv_diag()
    idx = rng.choice(n, size=9, replace=False)
    E = np.zeros((n, idx.size)); E[idx, np.arange(idx.size)] = 1.0
    S = be.backend_solve(E)
    assert np.allclose(d[idx], S[idx, np.arange(idx.size)], rtol=1e-7)
    Zs = be.get_selinv()
    assert abs((Zs.multiply(Q)).sum() - n) < 1e-6 * n            # tr(Q^-1 Q) = n on pattern(Q) subset of pattern(L)
    be.refactorize_values(Q.data * 3.0)
    assert abs(be.compute_logdet() - (ld + n * np.log(3.0))) < 1e-11 * abs(ld)
    be.close()
