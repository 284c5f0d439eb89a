"""GPU parity tests (run with -m gpu on an MI355X): every result of the HIP path, obtained
through the C ABI (ctypes -> libgmrfx.so), is compared with the CPU oracle on the same Q and the
same permutation, and with the dense identities the reference's own tests use
(test/workspace/test_gmrf_workspace.jl:26-224, test_backend_ordering.jl:25-68,
test_precision_logdet.jl:198-204). Tolerances: factor/solve/logdet 1e-10 relative (north_star
asks 1e-8), selinv diag 1e-8, full selinv 1e-6, as in the reference's tests."""
import numpy as np
import pytest
import scipy.sparse as sp

import gmrfx
import orc
from gmrfx import spde
from mirror import GMRFWorkspace, WorkspacePool

pytestmark = pytest.mark.gpu


def _cases():
    yield "rand20", spde.random_spd_precision(20), {}
    yield "rand400", spde.random_spd_precision(400, 0.02), {}
    m = spde.grid_mesh_2d(24, 24, jitter=0.25)
    yield "matern24_graph", spde.matern_precision(m, 0, 0.3), {}
    m = spde.grid_mesh_2d(64, 64, jitter=0.25)
    yield "matern64_coords", spde.matern_precision(m, 0, 0.2), {"coords": m.points}
    m = spde.grid_mesh_2d(65, 65)
    yield "cfg1_alpha3_65x65", spde.matern_precision(m, 1, 0.3), {"coords": m.points}
    m3 = spde.grid_mesh_3d(10, 10, 10)
    yield "matern3d_10", spde.matern_precision(m3, 0, 0.5), {"coords": m3.points}
    yield "natural_chain", spde.matern_precision(spde.grid_mesh_2d(14, 14), 0, 0.3), {"ordering": "natural"}
    yield "dense70", sp.csc_matrix(np.cov(np.random.default_rng(3).standard_normal((70, 300))) + np.eye(70)), {}
    yield "tall_fronts", spde.tall_front_precision(), {"ordering": "natural", "relax_cols": 1, "relax_zeros": 1e-9}
    yield "scalar", sp.csc_matrix(np.array([[2.5]])), {}
    yield "diag", sp.diags(np.arange(1.0, 40.0)).tocsc(), {}


CASES = list(_cases())


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def case(request):
    name, Q, kw = request.param
    Q = sp.csc_matrix(Q)
    ws = GMRFWorkspace(Q, **kw)
    F = orc.OracleFactor(Q, ws.backend.ordering_permutation())
    return name, Q, ws, F


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_factor_values_match_oracle(case):
    _, Q, ws, F = case
    Lg = ws.backend.factor_csc()
    Lo = F.L()
    # every true entry of L matches; extra stored entries (amalgamation zeros) are ~0
    D = (Lg - Lo)
    assert abs(D).max() <= 1e-10 * abs(Lo).max()
    assert ws.backend.last_info == 0


@pytest.mark.parametrize("nrhs", [1, 3, 16, 64, 70])
def test_solve_matches_oracle(case, nrhs):
    _, Q, ws, F = case
    n = Q.shape[0]
    B = np.random.default_rng(nrhs).standard_normal((n, nrhs))
    X = ws.workspace_solve(B[:, 0] if nrhs == 1 else B)
    Xo = F.solve(B[:, 0] if nrhs == 1 else B)
    assert X.shape == Xo.shape
    assert relerr(X, Xo) < 1e-10
    # dense identity of the reference's test: x ~ Q \ b
    if n <= 700:
        assert relerr(X.reshape(n, -1), np.linalg.solve(Q.toarray(), B[:, :X.reshape(n, -1).shape[1]])) < 1e-9


def test_selinv_contractions_on_device_match_oracle(case):
    # diag(A Sigma A') (linear_predictor_marginals.jl:125-165) and tr(Sigma B) (backend.jl:258-267) reduced on the
    # device from the selected-inverse panels: only m values / one scalar come back
    _, Q, ws, F = case
    n = Q.shape[0]
    rng = np.random.default_rng(17)
    # rows with two entries (j, k) that are neighbours in Q: every pair lies in pattern(Q), hence in the factor
    # pattern of BOTH implementations (the GPU's supernodes store a superset of the oracle's pattern(L))
    C = sp.triu(Q, 0).tocoo()
    pick = rng.choice(len(C.row), size=min(len(C.row), 300), replace=False)
    rows = np.repeat(np.arange(len(pick)), 2)
    cols = np.stack([C.row[pick], C.col[pick]], axis=1).ravel()
    A = sp.csr_matrix((rng.standard_normal(2 * len(pick)), (rows, cols)), shape=(len(pick), n))   # j == k rows: summed
    A = sp.vstack([A, sp.csr_matrix((1, n))]).tocsr()          # an empty row too
    v = ws.row_diag_ASigmaAt(A)
    vo = orc.row_diag_ASigmaAt(F, A)
    assert v.shape == (len(pick) + 1,) and v[-1] == 0.0
    assert np.abs(v - vo).max() <= 1e-8 * max(np.abs(vo).max(), 1e-300)
    # arbitrary columns: pairs outside the stored pattern count as 0 -- the same numbers as contracting the
    # extracted sparse Sigma on the host (what the reference does with selinv_extract_at(ws, A'A))
    k = min(3, n)
    m = 60
    rows = np.repeat(np.arange(m), k)
    cols = np.concatenate([rng.choice(n, size=k, replace=False) for _ in range(m)])
    A2 = sp.csr_matrix((rng.standard_normal(m * k), (rows, cols)), shape=(m, n))
    S_loc = ws.selinv_extract_at(sp.csc_matrix(abs(A2).T @ abs(A2))).toarray()
    want = np.einsum("ij,jk,ik->i", A2.toarray(), S_loc, A2.toarray())
    assert np.abs(ws.row_diag_ASigmaAt(A2) - want).max() <= 1e-10 * max(np.abs(want).max(), 1e-300)
    # B = Q itself: tr(Q^-1 Q) = n, the reference's own identity for selinv_dot (test_gmrf_workspace.jl)
    ws.ensure_selinv()
    d = ws.backend.selinv_dot_device(Q)
    assert abs(d - n) <= 1e-8 * n
    assert abs(d - orc.selinv_dot(F, Q)) <= 1e-8 * n
    assert abs(d - ws.selinv_dot(Q)) <= 1e-10 * n


def test_predictor_variances_of_fem_evaluation_matrix():
    # A = P1 evaluation at random points of the mesh (3 barycentric weights per row): the nodes of one triangle
    # are mutual neighbours in Q, so pattern(A'A) lies in pattern(L + L') and diag(A Sigma A') is exact
    mesh = spde.grid_mesh_2d(18, 18, jitter=0.25)
    Q = sp.csc_matrix(spde.matern_precision(mesh, 0, 0.4))
    n = Q.shape[0]
    rng = np.random.default_rng(2)
    cells = mesh.cells[rng.integers(0, len(mesh.cells), size=250)]
    wts = rng.dirichlet(np.ones(3), size=len(cells))
    A = sp.csr_matrix((wts.ravel(), (np.repeat(np.arange(len(cells)), 3), cells.ravel())), shape=(len(cells), n))
    ws = GMRFWorkspace(Q, coords=mesh.points)
    v = ws.row_diag_ASigmaAt(A)
    Sigma = np.linalg.inv(Q.toarray())
    want = np.einsum("ij,jk,ik->i", A.toarray(), Sigma, A.toarray())
    assert np.abs(v - want).max() <= 1e-9 * np.abs(want).max()
    with pytest.raises(ValueError):
        ws.backend.row_diag_ASigmaAt(sp.csr_matrix((2, n + 1)))
    # the pair plan stays on the device: new values of A and a new Q reuse it (one plan, same numbers as one-shot)
    assert len(ws.backend._rd_plans) == 1
    A2 = A.copy(); A2.data *= 1.5
    Q2 = sp.csc_matrix(spde.matern_precision(mesh, 0, 0.7))
    ws.update_precision(Q2)
    v2 = ws.row_diag_ASigmaAt(A2)
    assert len(ws.backend._rd_plans) == 1
    assert (v2 == ws.backend.row_diag_ASigmaAt_once(A2)).all()
    want2 = np.einsum("ij,jk,ik->i", A2.toarray(), np.linalg.inv(Q2.toarray()), A2.toarray())
    assert np.abs(v2 - want2).max() <= 1e-9 * np.abs(want2).max()
    # a clone has its own handle and builds its own plan
    wc = ws.backend.clone()
    assert np.abs(wc.row_diag_ASigmaAt(A2) - v2).max() <= 1e-12 * np.abs(v2).max()


def test_sqmahal_and_logpdf_match_oracle(case):
    # dot(r, Q r) and logpdf of workspace_gmrf.jl:288-292, Q's values and z resident on the device
    _, Q, ws, F = case
    n = Q.shape[0]
    rng = np.random.default_rng(11)
    z = rng.standard_normal(n); mu = rng.standard_normal(n)
    for mean in (None, mu):
        q = ws.sqmahal(z, mean)
        qo = orc.sqmahal(Q, z, mean)
        assert abs(q - qo) <= 1e-12 * max(1.0, abs(qo))
        r = z - (0 if mean is None else mean)
        assert abs(q - r @ (Q @ r)) <= 1e-11 * max(1.0, abs(qo))
        lp, lpo = ws.logpdf(z, mean), orc.logpdf(F, Q, z, mean)
        assert abs(lp - lpo) <= 1e-10 * max(1.0, abs(lpo))
    # a batch of vectors gives the same numbers as one call per vector (fixed summation order)
    Z = rng.standard_normal((n, 5))
    qb = ws.sqmahal(Z, mu)
    assert qb.shape == (5,)
    for k in range(5):
        assert qb[k] == ws.sqmahal(Z[:, k].copy(), mu)


def test_sqmahal_reads_only_the_defining_triangle():
    # Symmetric(Q) semantics (gmrf_workspace.jl:176): with both triangles stored the upper one defines Q
    # (or the lower with uplo="L"); garbage in the other triangle must not matter. One stored triangle is
    # used as it is. Explicit values override the handle's.
    Q = sp.csc_matrix(spde.matern_precision(spde.grid_mesh_2d(20, 20, jitter=0.25), 0, 0.3))
    n = Q.shape[0]
    rng = np.random.default_rng(5)
    z = rng.standard_normal((n, 3))
    want = np.einsum("ik,ik->k", z, Q @ z)
    U, Lo = sp.triu(Q).tocsc(), sp.tril(Q).tocsc()
    G = (U + 7.0 * sp.tril(Q, -1)).tocsc()          # lower triangle is garbage
    be = gmrfx.MI355XBackend(G)
    assert np.abs(be.sqmahal(z) - want).max() <= 1e-11 * np.abs(want).max()
    G2 = (Lo + 7.0 * sp.triu(Q, 1)).tocsc()
    be2 = gmrfx.MI355XBackend(G2, uplo="L")
    assert np.abs(be2.sqmahal(z) - want).max() <= 1e-11 * np.abs(want).max()
    for T in (U, Lo):
        bt = gmrfx.MI355XBackend(T)
        assert np.abs(bt.sqmahal(z) - want).max() <= 1e-11 * np.abs(want).max()
        assert abs(bt.sqmahal(z[:, 0].copy(), nzval=2.0 * T.data) - 2.0 * want[0]) <= 1e-11 * abs(want[0])
    # device-pointer form; a handle refactorised from a caller's device buffer does not hold the values
    import torch
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).cuda()
    d_z = torch.from_numpy(np.ascontiguousarray(z.T)).cuda()     # row k = vector k
    bq = gmrfx.MI355XBackend(Q, factorize=False)
    bq.refactorize_dev(d_nz.data_ptr())
    got = bq.quadform_dev(d_nz.data_ptr(), d_z.data_ptr(), n, 3)
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    with pytest.raises(ValueError):
        bq.quadform_dev(0, d_z.data_ptr(), n, 3)
    with pytest.raises(ValueError):
        bq.sqmahal(np.zeros(n + 1))


def test_logdet(case):
    _, Q, ws, F = case
    assert abs(ws.logdet() - F.logdet()) <= 1e-10 * max(1.0, abs(F.logdet()))
    if Q.shape[0] <= 700:
        assert abs(ws.logdet() - np.linalg.slogdet(Q.toarray())[1]) <= 1e-10 * max(1.0, abs(F.logdet()))
    assert ws.logdet_cov() == -ws.logdet()


@pytest.mark.parametrize("nrhs", [1, 5, 64])
def test_backward_solve(case, nrhs):
    _, Q, ws, F = case
    n = Q.shape[0]
    Z = np.random.default_rng(10 + nrhs).standard_normal((n, nrhs))
    X = ws.backward_solve(Z[:, 0] if nrhs == 1 else Z)
    Xo = F.backward_solve(Z[:, 0] if nrhs == 1 else Z)
    assert relerr(X, Xo) < 1e-10


def test_selinv_diag_and_full(case):
    _, Q, ws, F = case
    n = Q.shape[0]
    d = ws.selinv_diag()
    assert relerr(d, F.selinv_diag()) < 1e-8
    Zg = ws.selinv()
    Zo = F.selinv()
    # pattern contains pattern(Q); both triangles; values equal the true inverse there
    assert abs(Zg - Zg.T).max() == 0.0
    Qp = Q.copy(); Qp.data[:] = 1.0
    Zp = Zg.copy(); Zp.data[:] = 1.0
    assert (Qp - Qp.multiply(Zp)).nnz == 0
    # on the oracle's (true-fill) pattern the values agree
    Zo_p = Zo.copy(); Zo_p.data[:] = 1.0
    diff = Zg.multiply(Zo_p) - Zo
    assert abs(diff).max() <= 1e-6 * abs(Zo).max()
    # selinv_diag(ws) == diag(selinv(ws)) bit for bit (test_gmrf_workspace.jl:222-223)
    assert np.array_equal(Zg.diagonal(), d)
    if n <= 700:
        Qi = np.linalg.inv(Q.toarray())
        coo = Zg.tocoo()
        assert np.allclose(coo.data, Qi[coo.row, coo.col], rtol=1e-6, atol=1e-12 * abs(Qi).max())


def test_selinv_extract_and_dot(case):
    _, Q, ws, F = case
    n = Q.shape[0]
    Se = ws.selinv_extract_at(Q)
    Sf = ws.selinv()
    assert np.array_equal(Se.indptr, Q.indptr) and np.array_equal(Se.indices, Q.indices)
    full = Sf[Q.nonzero()]
    assert np.array_equal(np.asarray(Se[Q.nonzero()]).ravel(), np.asarray(full).ravel())   # bit-identical
    assert abs(ws.selinv_dot(Q) - n) <= 1e-8 * n
    # outside the factor pattern the extract reads 0
    if n >= 20:
        far = sp.csc_matrix(([1.0], ([0], [n - 1])), shape=(n, n))
        Zg = ws.selinv()
        if Zg[0, n - 1] == 0:
            assert ws.selinv_extract_at(far)[0, n - 1] == 0.0


def test_getters_are_cached_and_invalidate(case):
    _, Q, ws, F = case
    a = ws.selinv_diag(); b = ws.selinv_diag()
    assert a is b
    s1 = ws.selinv(); s2 = ws.selinv()
    assert s1 is s2
    Q2 = Q.copy(); Q2.data *= 2.0
    ws.update_precision(Q2)
    try:
        assert not ws.numeric_valid
        b = np.random.default_rng(5).standard_normal(Q.shape[0])
        assert relerr(ws.workspace_solve(b), F.solve(b) / 2.0) < 1e-10
        assert abs(ws.logdet() - (F.logdet() + Q.shape[0] * np.log(2.0))) <= 1e-10 * max(1.0, abs(F.logdet()))
        assert relerr(ws.selinv_diag(), F.selinv_diag() / 2.0) < 1e-8
        assert ws.selinv_diag() is not a
    finally:
        ws.update_precision_values(Q.data)
        ws.ensure_numeric()


def test_pattern_mismatch_and_length_errors():
    Q = spde.random_spd_precision(20)
    ws = GMRFWorkspace(Q)
    Qbad = Q.copy().tolil(); Qbad[0, 19] = 0.0; Qbad[19, 0] = 0.0
    Qbad = sp.csc_matrix(Qbad); Qbad.eliminate_zeros()
    if Qbad.nnz != Q.nnz:
        with pytest.raises(ValueError):
            ws.update_precision(Qbad)
    with pytest.raises(ValueError):
        ws.update_precision_values(np.ones(3))
    with pytest.raises(ValueError):
        ws.workspace_solve(np.ones(19))


def test_indefinite_matrix_behaviour():
    Q = spde.random_spd_precision(30)
    Qn = Q.copy(); Qn.setdiag(-1.0)
    Qn = sp.csc_matrix(Qn)
    # seam B: never throws (cholesky!(...; check=false), backend.jl:184), reports through info
    b = gmrfx.MI355XBackend(Qn)
    assert b.last_info > 0
    # seam A: throws PosDefException
    with pytest.raises(gmrfx.PosDefException):
        gmrfx.MI355XBackend(Qn, check_posdef=True)
    # handle stays usable after a good refactorisation
    b.refactorize(Q)
    assert b.last_info == 0
    x = b.backend_solve(np.ones(30))
    assert relerr(Q @ x, np.ones(30)) < 1e-10


def test_explicit_permutation_and_orderings_agree():
    """Orderings must not change answers (test_backend_ordering.jl:25-68)."""
    m = spde.grid_mesh_2d(12, 12)
    Q = spde.matern_precision(m, 0, 0.4)
    n = Q.shape[0]
    rng = np.random.default_rng(0)
    b = rng.standard_normal(n)
    ref = np.linalg.solve(Q.toarray(), b)
    ld = np.linalg.slogdet(Q.toarray())[1]
    dg = np.diag(np.linalg.inv(Q.toarray()))
    for kw in ({}, {"ordering": rng.permutation(n)}, {"ordering": "natural"}, {"coords": m.points}):
        ws = GMRFWorkspace(Q, **kw)
        assert relerr(ws.workspace_solve(b), ref) < 1e-10
        assert abs(ws.logdet() - ld) < 1e-10 * abs(ld)
        assert relerr(ws.selinv_diag(), dg) < 1e-8


def test_clone_is_independent():
    Q = spde.random_spd_precision(50)
    b = gmrfx.MI355XBackend(Q)
    c = b.clone()
    Q2 = Q.copy(); Q2.data *= 3.0
    b.refactorize(Q2)
    rhs = np.ones(50)
    assert relerr(c.backend_solve(rhs) / 3.0, b.backend_solve(rhs)) < 1e-12
    assert abs(c.compute_logdet() + 50 * np.log(3.0) - b.compute_logdet()) < 1e-9


def test_sampling_moments():
    """cov(backward_solve(z)) = Q^-1 in ORIGINAL ordering (test_gmrf_workspace.jl:85-100)."""
    Q = spde.random_spd_precision(20)
    ws = GMRFWorkspace(Q)
    rng = np.random.default_rng(123)
    Zs = rng.standard_normal((20, 50000))
    S = ws.backward_solve(Zs)
    emp = (S * S).mean(axis=1)
    assert np.allclose(emp, np.diag(np.linalg.inv(Q.toarray())), rtol=0.1)


def test_pool_threads():
    """Independent workspaces used concurrently (test_workspace_pool.jl:88-115)."""
    import threading
    m = spde.grid_mesh_2d(20, 20, jitter=0.2)
    Q = spde.matern_precision(m, 0, 0.3)
    pool = WorkspacePool(Q, size=3, coords=m.points)
    ref = np.linalg.slogdet(Q.toarray())[1]
    out = {}

    def work(i):
        with pool.with_workspace() as ws:
            ws.update_precision_values(Q.data * (i + 1))
            out[i] = ws.logdet() - Q.shape[0] * np.log(i + 1)
            ws.update_precision_values(Q.data)

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in th]; [t.join() for t in th]
    assert all(abs(v - ref) < 1e-9 * abs(ref) for v in out.values()) and len(out) == 6


def test_residual_property_256():
    """Size-independent property at a size the dense check cannot reach: ||QX-B||/||B||."""
    m = spde.grid_mesh_2d(256, 256, jitter=0.25)
    Q = spde.matern_precision(m, 0, 0.2)
    ws = GMRFWorkspace(Q, coords=m.points)
    B = np.random.default_rng(1).standard_normal((Q.shape[0], 64))
    X = ws.workspace_solve(B)
    assert np.linalg.norm(Q @ X - B) / np.linalg.norm(B) < 1e-10
    F = orc.OracleFactor(Q, ws.backend.ordering_permutation())
    assert abs(ws.logdet() - F.logdet()) < 1e-10 * abs(F.logdet())
    assert relerr(ws.selinv_diag(), F.selinv_diag()) < 1e-8
    # tr(Q^-1 Q) = n
    assert abs(ws.selinv_dot(Q) - Q.shape[0]) < 1e-8 * Q.shape[0]


@pytest.mark.parametrize("name", ["rand400", "matern64_coords", "cfg1_alpha3_65x65", "matern3d_10", "natural_chain"])
def test_generic_path_matches_fused_small_front_path(name, monkeypatch):
    """The fused small-front kernels (default) and the generic level-batched kernels
    (GMRFX_SMALL_ROWS=0) must both agree with the oracle."""
    Q, kw = next((sp.csc_matrix(q), k) for n, q, k in CASES if n == name)
    monkeypatch.setenv("GMRFX_SMALL_ROWS", "0")
    ws = GMRFWorkspace(Q, **kw)
    assert ws.backend.stats()["n_small_fronts"] == 0
    F = orc.OracleFactor(Q, ws.backend.ordering_permutation())
    assert abs(ws.backend.factor_csc() - F.L()).max() <= 1e-10 * abs(F.L()).max()
    B = np.random.default_rng(0).standard_normal((Q.shape[0], 64))
    assert relerr(ws.workspace_solve(B), F.solve(B)) < 1e-10
    assert relerr(ws.backward_solve(B), F.backward_solve(B)) < 1e-10
    assert relerr(ws.selinv_diag(), F.selinv_diag()) < 1e-8
    for rows in ("96", "128"):      # the other size-class cut-offs (default: 64)
        monkeypatch.setenv("GMRFX_SMALL_ROWS", rows)
        wsr = GMRFWorkspace(Q, **kw)
        assert relerr(wsr.workspace_solve(B), F.solve(B)) < 1e-10
        assert relerr(wsr.selinv_diag(), F.selinv_diag()) < 1e-8


GOLD = sorted(__import__("glob").glob(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[__import__("os").path.basename(p)[:-4] for p in GOLD])
def test_hip_path_matches_golden_fixtures(path):
    """Committed dense-float64 known answers (tests/golden/make_golden.py), same permutation."""
    g = np.load(path)
    n = int(g["n"])
    Q = sp.csc_matrix((g["nzval"], g["rowval"], g["colptr"]), shape=(n, n))
    ws = GMRFWorkspace(Q, ordering=g["perm"])
    # the user's order is kept up to an etree postorder: same fill, exact integer check
    assert ws.backend.stats()["nnz_l"] == int(g["L_colcount"].sum())
    assert relerr(ws.workspace_solve(g["B"]), g["X"]) < 1e-10
    assert abs(ws.logdet() - float(g["logdet"])) <= 1e-10 * max(1.0, abs(float(g["logdet"])))
    assert relerr(ws.selinv_diag(), g["selinv_diag"]) < 1e-8
    assert np.allclose(ws.selinv_extract_at(Q).data, g["Qinv_on_pattern"], rtol=1e-6, atol=1e-13)
    # F.UP \ z depends on the elimination order actually used: compare through the oracle on it
    F = orc.OracleFactor(Q, ws.backend.ordering_permutation())
    assert relerr(ws.backward_solve(g["Z"]), F.backward_solve(g["Z"])) < 1e-10
    # and its covariance identity is order-independent: x = P'L^-T z  =>  Q = (x-map)^-T (x-map)^-1
    Xb = ws.backward_solve(np.eye(n))
    assert np.allclose(Xb @ Xb.T, np.linalg.inv(Q.toarray()), rtol=1e-7, atol=1e-12)


def test_full_size_properties_cfg2():
    """BASELINE.json config 2 at FULL size (10^6 nodes): no oracle can follow here, so the check is
    through size-independent properties: residual of the 64-RHS solve, bit-reproducibility of the
    factorisation (no atomics anywhere), logdet scaling law, solve linearity, backward-solve
    covariance identity on probe vectors, tr(Q^-1 Q) = n from the selected inverse."""
    m = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
    Q = spde.matern_precision(m, 0, 0.2)
    n = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, coords=m.points)
    assert be.last_info == 0
    rng = np.random.default_rng(1)
    B = rng.standard_normal((n, 64))
    ld1 = be.compute_logdet()
    f1 = be.factor_values()           # (taken before any solve: the dense inverses of the big fronts
    be.refactorize(Q)                 #  are filled in lazily by the first sweep)
    assert np.array_equal(be.factor_values(), f1)            # deterministic, bit for bit
    assert be.compute_logdet() == ld1
    X = be.backend_solve(B)
    assert np.linalg.norm(Q @ X - B) / np.linalg.norm(B) < 1e-9
    X2 = be.backend_solve(B)
    assert np.array_equal(X, X2)                             # sweeps are bit-reproducible too
    be.refactorize_values(Q.data * 2.0)
    assert abs(be.compute_logdet() - (ld1 + n * np.log(2.0))) < 1e-10 * abs(ld1)
    x2 = be.backend_solve(B[:, :2])
    # (2Q)^-1 b = Q^-1 b / 2; the two factors round differently and cond(Q) ~ 1e8 here
    assert np.linalg.norm(2.0 * x2 - X[:, :2]) / np.linalg.norm(X[:, :2]) < 1e-7
    be.refactorize(Q)
    # x = P' L^-T z  =>  Q x = P' L z' ... equivalently z'z = x' Q x for every probe z
    Z = rng.standard_normal((n, 4))
    S = be.backend_backward_solve(Z)
    assert np.allclose(np.einsum("ij,ij->j", S, Q @ S), np.einsum("ij,ij->j", Z, Z), rtol=1e-9)
    d = be.get_selinv_diag()
    assert d.min() > 0
    assert abs(be.selinv_dot(Q) - n) < 1e-7 * n
    # diag(Q^-1) against 3 unit-vector solves
    for k in (0, n // 2 + 17, n - 1):
        e = np.zeros(n); e[k] = 1.0
        assert abs(be.backend_solve(e)[k] - d[k]) < 1e-7 * d[k]


def test_midsize_multiblock_fronts_match_oracle():
    """300 x 300 nodes: the top separators have several hundred columns, so this is the smallest case that
    runs the two-level blocked panel factorisation (K = 256 updates), the latency variants of TRSM / GEMM /
    sweep GEMMs, the dense-inverse stages beyond B = 64 and the multi-batch K loops -- checked VALUE BY VALUE
    against the oracle (the 1M-node test can only check properties)."""
    m = spde.grid_mesh_2d(300, 300, jitter=0.25, seed=5)
    Q = sp.csc_matrix(spde.matern_precision(m, 0, 0.2))
    n = Q.shape[0]
    ws = GMRFWorkspace(Q, coords=m.points)
    st = ws.backend.stats()
    assert st["max_cols"] > 256          # the K = 256 outer update and >= 3 inverse stages are exercised
    F = orc.OracleFactor(Q, ws.backend.ordering_permutation())
    Lg, Lo = ws.backend.factor_csc(), F.L()
    assert abs(Lg - Lo).max() <= 1e-10 * abs(Lo).max()
    assert abs(ws.logdet() - F.logdet()) <= 1e-11 * abs(F.logdet())
    rng = np.random.default_rng(11)
    for nrhs in (1, 70):      # 70 = one full 64-wide chunk + a ragged one
        B = rng.standard_normal((n, nrhs))
        X = ws.workspace_solve(B[:, 0] if nrhs == 1 else B)
        Xo = F.solve(B[:, 0] if nrhs == 1 else B)
        assert relerr(X, Xo) < 1e-10
    Z = rng.standard_normal((n, 64))
    assert relerr(np.column_stack([ws.backward_solve(Z[:, j]) for j in range(2)]), F.backward_solve(Z[:, :2])) < 1e-10
    assert relerr(ws.selinv_diag(), F.selinv_diag()) < 1e-8


def test_disconnected_and_arrow_patterns():
    """Edge patterns: a block-diagonal Q (forest of elimination trees: several roots) and an arrow matrix
    (one dense row/column: a front whose row structure is the whole matrix)."""
    rng = np.random.default_rng(4)
    blocks = [spde.random_spd_precision(k, 0.2) for k in (1, 7, 40, 130)]
    Qb = sp.block_diag(blocks, format="csc")
    n = 300
    A = sp.diags(rng.uniform(2.0, 3.0, n)).tolil()
    A[0, :] = 0.01; A[:, 0] = 0.01; A[0, 0] = 5.0
    for Q in (Qb, sp.csc_matrix(A)):
        Q = sp.csc_matrix(Q)
        ws = GMRFWorkspace(Q)
        F = orc.OracleFactor(Q, ws.backend.ordering_permutation())
        B = rng.standard_normal((Q.shape[0], 5))
        assert relerr(ws.workspace_solve(B), np.linalg.solve(Q.toarray(), B)) < 1e-10
        assert abs(ws.logdet() - np.linalg.slogdet(Q.toarray())[1]) < 1e-9 * max(1.0, abs(F.logdet()))
        assert relerr(ws.selinv_diag(), np.diag(np.linalg.inv(Q.toarray()))) < 1e-8


@pytest.mark.parametrize("name", ["matern64_coords", "cfg1_alpha3_65x65", "matern3d_10", "rand400"])
def test_subtree_tasks_on_and_off_agree(name, monkeypatch):
    """Whole-subtree workgroup tasks (GMRFX_SUBTREE_MAX=24) vs pure level scheduling (the default, = 0):
    same factor bit for bit (same arithmetic per front), same answers."""
    Q, kw = next((sp.csc_matrix(q), k) for n, q, k in CASES if n == name)
    monkeypatch.setenv("GMRFX_SUBTREE_MAX", "24")
    ws_on = GMRFWorkspace(Q, **kw)
    monkeypatch.setenv("GMRFX_SUBTREE_MAX", "0")
    monkeypatch.setenv("GMRFX_SWEEP_TASK_ROWS", "0")      # pure level schedule on both sides (sweep tasks sum in another order)
    ws_off = GMRFWorkspace(Q, **kw)
    assert np.array_equal(ws_on.backend.factor_values(), ws_off.backend.factor_values())
    B = np.random.default_rng(0).standard_normal((Q.shape[0], 64))
    assert np.array_equal(ws_on.workspace_solve(B), ws_off.workspace_solve(B))
    assert np.array_equal(ws_on.selinv_diag(), ws_off.selinv_diag())


def _shard_gpu_worker(rank, world, port, q, case="2d"):
    """One rank of the sharded factorisation (gmrfx/shard.py); all ranks share cuda:0, the process group is
    gloo (RCCL refuses two ranks on one device): the exchange goes through host staging, everything else is
    the real HIP path."""
    try:
        import os, sys
        here = os.path.dirname(os.path.abspath(__file__))
        for p in (os.path.join(os.path.dirname(here), "gaussianmarkovrandomfields.jl_amd"), os.path.join(os.path.dirname(here), "oracle"), here):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import gmrfx as g
        from gmrfx import spde as sp_, shard
        if case == "3d_dist":      # 28^3-node 3-D mesh: the root front (~800 columns) is factored by all ranks together, at world 4 the
            os.environ["GMRFX_DIST_MIN"] = "256"        # separators below it (~400 columns, with contribution blocks) by two ranks each
            m = sp_.grid_mesh_3d(28, 28, 28)
            Q = sp_.matern_precision(m, 0, 0.4)
        elif case == "spacetime":  # BASELINE cfg 5 in small: AR(1) over 24 time steps (x) 2-D Matern + observation term, handed WHOLE to the
            os.environ["GMRFX_DIST_MIN"] = "256"        # factorisation on the space-time dissection (separable.jl:143-172); the top
            from gmrfx import spacetime as st_          # separators (time slabs / space cuts of ~600 columns) are DISTRIBUTED fronts
            ms = sp_.grid_mesh_2d(24, 23, jitter=0.2, seed=4)
            Qs = sp_.matern_precision(ms, 0, 0.4)
            T = 24
            Q = st_.spacetime_precision(sp_.ar1_precision(T, 0.8), Qs, obs_diag=np.random.default_rng(5).uniform(0.5, 2.0, T * Qs.shape[0]))
            class _M: pass
            m = _M(); m.points = st_.spacetime_coords(ms.points, T)
        else:
            m = sp_.grid_mesh_2d(120, 120, jitter=0.25, seed=2)
            Q = sp_.matern_precision(m, 0, 0.2)
        dev = torch.device("cuda", 0)
        d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
        sf = shard.ShardedFactor(Q, dist, device=0, coords=m.points)

        def same_t(a, b):
            # The FACTOR is compared bit for bit in every case. The sweeps pick kernel variants by the size of a launch (split-K
            # depth of the top-level updates: a level of a sharded handle holds fewer fronts), so the solves of the 3-D case
            # agree to rounding; on the 2-D mesh the same variants run and the solves are bit-identical as well.
            if case == "2d":
                return bool(torch.equal(a, b))
            return float((a - b).abs().max()) <= 1e-11 * float(b.abs().max())

        if case == "3d_dist":
            assert len(sf.df["front"]) >= 1 and max(sf.df["cols"]) >= 513 and sf.df["group"][-1] == list(range(world)), sf.df
        if case == "spacetime":
            assert len(sf.df["front"]) >= 1 and max(sf.df["cols"]) > 256, sf.df
        for _ in range(2):                               # twice: the second run reuses every buffer
            assert sf.refactorize_dev(d_nz.data_ptr()) == 0
        # the form bench.py times: no pivot report (host round trip) between the factorisation and what follows; it rides with logdet()
        sf.last_info = -1
        assert sf.refactorize_dev(d_nz.data_ptr(), check=False) == 0 and sf._info_pending
        ld = sf.logdet()
        assert sf.last_info == 0 and not sf._info_pending
        # sharded solve: B on every rank, X on rank 0; must equal the unsharded solve bit for bit
        nrhs = 64 if world == 2 else 7
        Bh = torch.randn((nrhs, Q.shape[0]), generator=torch.Generator().manual_seed(3), dtype=torch.float64)
        d_B = Bh.to(dev); d_X = torch.zeros_like(d_B)
        torch.cuda.synchronize()      # torch's fill kernel runs on torch's stream, the library on its own ones
        sf.solve_dev(d_B.data_ptr(), Q.shape[0], nrhs, d_X.data_ptr(), Q.shape[0])
        owner = sf.be.shard_owner()
        mine = owner == rank
        sy = sf.be.symbolic()
        vals = sf.be.factor_values()
        # compare the panels this rank factored with an unsharded handle, bit for bit
        ref = g.MI355XBackend(Q, coords=m.points, device=0)
        rv = ref.factor_values()
        rsy = ref.symbolic()            # a sharded handle only stores its own panels: its panel offsets are its own
        same = True
        why = []
        check = set(np.nonzero(mine)[0].tolist())
        for i, s_ in enumerate(sf.df["front"]):     # after its last broadcast EVERY member of the group holds a distributed front's panel
            if rank in sf.df["group"][i]:
                check.add(int(s_))
        dist_pos = {int(s_): i for i, s_ in enumerate(sf.df["front"])}
        n_compact = 0
        for s in sorted(check):
            a, ar = int(sy.panel_ptr[s]), int(rsy.panel_ptr[s])
            c, r, ldp = int(sy.super_first[s + 1] - sy.super_first[s]), int(sy.row_ptr[s + 1] - sy.row_ptr[s]), int(sy.panel_ld[s])
            Pb = rv[ar:ar + ldp * c].reshape(c, ldp).T[:r]
            if s in dist_pos and r == c and owner[s] != rank:
                # block-cyclic STORAGE (round 6): a member that is not the owner of a front without trailing rows keeps its own
                # 256-column blocks only (gmrfx_dist_front_block says where): those must equal the unsharded factor's columns
                G_ = sf.df["group"][dist_pos[s]]
                pos = G_.index(rank)
                n_compact += 1
                for b in range((c + 255) // 256):
                    if b % len(G_) != pos:
                        continue
                    off, cnt = sf.be.dist_front_block(s, b)
                    nc = min(256, c - 256 * b)
                    assert cnt == nc * ldp and off == a + (b // len(G_)) * 256 * ldp
                    Pa = vals[off:off + cnt].reshape(nc, ldp).T[:r]
                    Pbb = Pb[:, 256 * b:256 * b + nc]
                    msk = np.arange(r)[:, None] >= (256 * b + np.arange(nc))[None, :]
                    if not np.array_equal(Pa * msk, Pbb * msk):
                        same = False
                        why.append(f"own block {b} of the distributed front {s} differs by {np.abs((Pa - Pbb) * msk).max():.3e}")
                continue
            Pa = vals[a:a + ldp * c].reshape(c, ldp).T[:r]
            if not np.array_equal(np.tril(Pa), np.tril(Pb)):
                same = False
                why.append(f"panel {s} (c={c}, r={r}) differs by {np.abs(np.tril(Pa) - np.tril(Pb)).max():.3e}")
        if rank == 0:
            d_Xr = torch.zeros_like(d_B)
            torch.cuda.synchronize()
            ref.solve_dev(d_B.data_ptr(), Q.shape[0], nrhs, d_Xr.data_ptr(), Q.shape[0])
            torch.cuda.synchronize()
            if not same_t(d_X, d_Xr):
                same = False
                why.append(f"sharded solve differs from the unsharded one by {float((d_X - d_Xr).abs().max()):.3e}")
            X = d_X.cpu().numpy().T
            resid = float(np.linalg.norm(Q @ X - Bh.numpy().T) / np.linalg.norm(Bh.numpy()))
            if not resid < 1e-10:
                same = False
                why.append(f"residual {resid:.3e}")
        # X left DISTRIBUTED (gather = False: no transfer behind the backward sweep): every rank's valid rows equal the unsharded solution,
        # and the ranks' masks cover every row (the top fronts' rows on all of them)
        d_Xd = torch.full_like(d_B, float("nan"))
        torch.cuda.synchronize()
        sf.solve_dev(d_B.data_ptr(), Q.shape[0], nrhs, d_Xd.data_ptr(), Q.shape[0], gather=False)
        d_Xall = torch.zeros_like(d_B)
        torch.cuda.synchronize()
        ref.solve_dev(d_B.data_ptr(), Q.shape[0], nrhs, d_Xall.data_ptr(), Q.shape[0])
        torch.cuda.synchronize()
        vr = sf.valid_rows()
        vmask = torch.from_numpy(vr).to(dev)
        if not same_t(d_Xd[:, vmask], d_Xall[:, vmask]):
            same = False
            why.append(f"distributed solve: the valid rows of rank {rank} differ from the unsharded solution by {float((d_Xd[:, vmask] - d_Xall[:, vmask]).abs().max()):.3e}")
        cover = torch.from_numpy(vr.astype(np.int64))
        dist.all_reduce(cover)
        if int(cover.min()) < 1 or int(vr.sum()) >= Q.shape[0]:
            same = False
            why.append(f"valid_rows: coverage min {int(cover.min())}, rank {rank} claims {int(vr.sum())} of {Q.shape[0]} rows")
        # B ROW-SHARDED (round 6): a rank reads only the rows needed_rows() names -- its subtrees' and its own top fronts'. Everything
        # else of B is set to garbage here (1e30: a real dependence on it would wreck the answer; a clamped / masked read multiplies it
        # by zero); the masks of the ranks partition the rows. Same bits as with the full B.
        need = sf.needed_rows()
        part = torch.from_numpy(need.astype(np.int64))
        dist.all_reduce(part)
        if int(part.min()) != 1 or int(part.max()) != 1:
            same = False
            why.append(f"needed_rows: the ranks' masks do not partition the rows (min {int(part.min())}, max {int(part.max())})")
        d_Bp = d_B.clone()
        d_Bp[:, torch.from_numpy(~need).to(dev)] = 1e30
        d_Xp = torch.full_like(d_B, float("nan"))
        torch.cuda.synchronize()
        sf.solve_dev(d_Bp.data_ptr(), Q.shape[0], nrhs, d_Xp.data_ptr(), Q.shape[0], gather=False)
        torch.cuda.synchronize()
        if not torch.equal(d_Xp[:, vmask], d_Xd[:, vmask]):
            same = False
            why.append(f"row-sharded B: the valid rows of rank {rank} differ from the full-B solve by {float((d_Xp[:, vmask] - d_Xd[:, vmask]).abs().max()):.3e}")
        # backward-only solve (F.UP \\ z, the sampling path) and more than 64 columns (two passes), sharded: same bits as unsharded
        nb = 70 if world == 2 else 5
        Zh = torch.randn((nb, Q.shape[0]), generator=torch.Generator().manual_seed(4), dtype=torch.float64)
        d_Z = Zh.to(dev); d_S = torch.zeros_like(d_Z)
        torch.cuda.synchronize()
        sf.backward_solve_dev(d_Z.data_ptr(), Q.shape[0], nb, d_S.data_ptr(), Q.shape[0])
        d_W = torch.zeros_like(d_Z)
        sf.solve_dev(d_Z.data_ptr(), Q.shape[0], nb, d_W.data_ptr(), Q.shape[0])
        torch.cuda.synchronize()
        if rank == 0:
            d_Sr = torch.zeros_like(d_Z); d_Wr = torch.zeros_like(d_Z)
            torch.cuda.synchronize()
            ref.backward_solve_dev(d_Z.data_ptr(), Q.shape[0], nb, d_Sr.data_ptr(), Q.shape[0])
            ref.solve_dev(d_Z.data_ptr(), Q.shape[0], nb, d_Wr.data_ptr(), Q.shape[0])
            torch.cuda.synchronize()
            if not same_t(d_S, d_Sr):
                same = False
                why.append(f"sharded backward solve ({nb} columns) differs from the unsharded one by {float((d_S - d_Sr).abs().max()):.3e}")
            if not same_t(d_W, d_Wr):
                same = False
                why.append(f"sharded {nb}-column solve differs from the unsharded one by {float((d_W - d_Wr).abs().max()):.3e}")
        # sharded selected inversion: the all-reduced diagonal equals the unsharded one bit for bit
        sf.selinv_compute()
        dsh = sf.selinv_diag()
        dref = ref.get_selinv_diag()
        if not (np.array_equal(dsh, dref) if case == "2d" else np.abs(dsh - dref).max() <= 1e-11 * np.abs(dref).max()):
            same = False
            why.append(f"sharded selinv diagonal differs from the unsharded one by {np.abs(dsh - dref).max():.3e} (rel {np.abs(dsh / dref - 1).max():.3e})")
        info = dict(sf.be.shard_info()); info["why"] = why[:5]; info["n_dist"] = len(sf.df["front"]); info["n_compact"] = n_compact
        info["n_dist_cb"] = int(sum(1 for i in range(len(sf.df["front"])) if sf.df["rows"][i] > sf.df["cols"][i] and len(sf.df["group"][i]) > 1))
        st, rst = sf.be.stats(), ref.stats()
        info["mem"] = {k: (float(st[k]), float(rst[k])) for k in ("bytes_factor", "bytes_cb_arena", "bytes_device_total")}
        q.put((rank, ld, ref.compute_logdet(), bool(same), int(mine.sum()), info))
        dist.barrier()
        sf.close(); ref.close()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_factorisation_rehearsal_on_one_gpu(world):
    """ONE factorisation sharded over `world` processes (subtrees per rank, every top front on one rank
    of its group, Schur-complement contribution blocks / update vectors point-to-point along the owner-crossing tree
    edges, x of the top fronts broadcast by their owners, all-reduced logdet and pivot check) -- rehearsed with all
    ranks on this one GPU. Every rank's panels and the solution must equal the unsharded ones bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_gpu_worker, args=(r, world, 29650 + world, q)) for r in range(world)]
    [p.start() for p in procs]
    got = []
    for _ in range(world):
        item = q.get(timeout=240)
        assert item[0] != "error", f"rank {item[1]} failed:\n{item[2]}"
        got.append(item)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ld, ld_ref, same, nmine, info in got:
        assert same and nmine > 0, f"rank {rank}: {info.get('why')}"
        assert abs(ld - ld_ref) <= 1e-12 * abs(ld_ref)
        assert info["n_top_fronts"] >= 1
    _check_shard_memory(got, world)


def _check_shard_memory(got, world):
    # per-rank memory: a rank stores the panels of its own fronts only -- every panel exactly once over the ranks -- and
    # its arena holds its own contribution blocks plus the exchange region of the cross-edge children
    fac = [info["mem"]["bytes_factor"][0] for *_, info in got]
    ref_fac = got[0][5]["mem"]["bytes_factor"][1]
    assert abs(sum(fac) - ref_fac) <= 0.01 * ref_fac + 128 * 8 * 50000
    assert max(fac) <= (0.70 if world == 2 else 0.45) * ref_fac, (fac, ref_fac)
    for *_, info in got:
        assert info["mem"]["bytes_cb_arena"][0] <= 1.05 * info["mem"]["bytes_cb_arena"][1]
        # (bytes_device_total also holds the right-hand-side panels, n x 64 doubles twice, and the symbolic tables, which
        #  every rank keeps whole: on this 14 400-node mesh they outweigh the factor)


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_top_fronts_rehearsal_on_one_gpu(world):
    """Top fronts factored by their whole GROUP (256-column blocks dealt cyclically, block-column broadcasts inside the group,
    K = 256 updates of the own later blocks, own column blocks of the contribution block, children's blocks by column ranges;
    Symbolic::dist_fronts, Device::dist_front_phase, shard.py _factor_distributed_front) on a 28^3-node 3-D mesh: the root by
    all ranks, at world 4 the separators below it by two ranks each. Every rank's panels -- and a distributed front's whole
    panel on every member of its group -- equal the unsharded factor bit for bit, and so do the log-determinant, and to
    rounding the solves and the selected-inverse diagonal."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_gpu_worker, args=(r, world, 29750 + world, q, "3d_dist")) for r in range(world)]
    [p.start() for p in procs]
    got = []
    for _ in range(world):
        item = q.get(timeout=300)
        assert item[0] != "error", f"rank {item[1]} failed:\n{item[2]}"
        got.append(item)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ld, ld_ref, same, nmine, info in got:
        assert same and nmine > 0 and info["n_dist"] >= 1, f"rank {rank}: {info.get('why')}"
        if world == 4:
            assert info["n_dist_cb"] >= 1, info        # a distributed front WITH a contribution block was exercised
        assert abs(ld - ld_ref) <= 1e-12 * abs(ld_ref)
    # the root (no trailing rows) is stored whole by its owner only: every other rank went through the block-cyclic storage path
    assert sum(1 for *_, info in got if info["n_compact"] >= 1) == world - 1, [info["n_compact"] for *_, info in got]


def test_randomised_pattern_sweep():
    """36 random patterns (sparse random, ragged 2-D / 3-D meshes, forests, bands, dense multi-block fronts; 1..65
    right-hand sides) against dense LAPACK identities: tools/fuzz_gpu.py."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py"), "36", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_newton_update_on_device_matches_host_update():
    """Q_k = Q_prior - H_k formed on the device from the Hessian values only (index map uploaded
    once). Checked against the ORACLE factor of Q_prior - H_k (value by value, logdet, one solve) and, bit for bit,
    against refactorising from host-side updated values (the reference's _update_hessian!,
    src/workspace/gaussian_approximation.jl:103-129), for diagonal and sparse Hessians."""
    m = spde.grid_mesh_2d(48, 48, jitter=0.2, seed=9)
    Q = sp.csc_matrix(spde.matern_precision(m, 0, 0.3))
    n = Q.shape[0]
    rng = np.random.default_rng(12)
    be = gmrfx.MI355XBackend(Q, coords=m.points)
    ref = gmrfx.MI355XBackend(Q, coords=m.points)
    perm = be.ordering_permutation()
    b = rng.standard_normal(n)

    def against_oracle(nz):
        Qk = sp.csc_matrix((nz, Q.indices, Q.indptr), shape=Q.shape)
        F = orc.OracleFactor(Qk, perm)
        Lg, Lo = be.factor_csc(), F.L()
        assert abs(Lg - Lo).max() <= 1e-10 * abs(Lo).max()
        assert abs(be.compute_logdet() - F.logdet()) <= 1e-11 * abs(F.logdet())
        assert relerr(be.backend_solve(b), F.solve(b)) < 1e-10

    # diagonal Hessian (Diagonal branch): positions of the diagonal entries in nzval
    diag_idx = np.array([Q.indptr[j] + np.searchsorted(Q.indices[Q.indptr[j]:Q.indptr[j + 1]], j) for j in range(n)])
    be.set_prior(Q.data, diag_idx)
    for _ in range(3):
        h = -rng.uniform(0.1, 2.0, n)                        # H = -diag(.) : Q - H stays SPD
        be.refactorize_update(h)
        nz = Q.data.copy(); nz[diag_idx] -= h
        ref.refactorize_values(nz)
        # (bit comparison BEFORE any solve: the first sweep fills the dense inverses of the big fronts into the
        #  unused strict upper triangles of the diagonal blocks)
        assert np.array_equal(be.factor_values(), ref.factor_values())
        assert be.compute_logdet() == ref.compute_logdet()
        against_oracle(nz)
    # sparse Hessian on a SUB-pattern of Q (SparseMatrixCSC branch): the diagonal plus every third off-diagonal pair
    # (i, j) / (j, i) -- a symmetric selection -- with symmetric negative-definite values
    coo = Q.tocoo()
    lo, hi = np.minimum(coo.row, coo.col), np.maximum(coo.row, coo.col)
    keep = np.flatnonzero((lo == hi) | ((lo * 7919 + hi) % 3 == 0))
    assert 0 < keep.size < Q.nnz
    H = Q.copy(); H.data = -0.05 * np.abs(H.data)
    H.data[diag_idx] = -(0.2 + np.abs(Q).sum(axis=0).A1 * 0.05)      # diagonally dominant: -H is SPD
    be.set_prior(Q.data, keep)
    be.refactorize_update(H.data[keep])
    nz = Q.data.copy(); nz[keep] -= H.data[keep]
    ref.refactorize_values(nz)
    assert np.array_equal(be.factor_values(), ref.factor_values())
    against_oracle(nz)
    with pytest.raises(ValueError):
        be.refactorize_update(np.zeros(3))
    # the whole iterate as ONE pipelined call (gmrfx_refactorize_update_solve): same factor, same solve, bit for bit
    B3 = rng.standard_normal((n, 3))
    hk = 0.5 * H.data[keep]
    Xf = be.refactorize_update_solve(hk, B3)
    nz = Q.data.copy(); nz[keep] -= hk
    ref.refactorize_values(nz)
    assert np.array_equal(Xf, ref.backend_solve(B3))
    assert np.array_equal(be.factor_values(), ref.factor_values())      # (after the solve on both: see the note above)
    with pytest.raises(ValueError):
        be.refactorize_update_solve(np.zeros(3), B3)


# ---- KL (Vecchia) sparse approximate Cholesky: kl_cholesky.jl:32-55 and :74-113 -------------------

def _kl_problem(n, rho_len, seed):
    rng = np.random.default_rng(seed)
    X = rng.random((n, 2))
    d = np.linalg.norm(X[:, None, :] - X[None, :, :], axis=-1)
    K = (1 + np.sqrt(3) * d / 0.3) * np.exp(-np.sqrt(3) * d / 0.3)          # Matern 3/2, as in the reference's test
    from gmrfx import klchol
    return X, K, klchol.radius_pattern(X, rho_len)


@pytest.mark.parametrize("n,rho_len", [(4, 2.0), (60, 0.25), (300, 0.12), (300, 0.3), (500, 0.5), (400, 2.0)])
def test_kl_cholesky_columns_match_oracle(n, rho_len):
    # local systems of up to 32 / 64 / 128 rows (LDS classes) and beyond (global-scratch class; (400, 2.0) is the
    # complete pattern: 400-row systems and L L' = (K + 1e-6 I)^-1 exactly)
    from gmrfx import klchol
    X, K, P = _kl_problem(n, rho_len, seed=n)
    L = klchol.sparse_approximate_cholesky_inplace(K, P)
    Lo = orc.kl_cholesky_inplace(K, P)
    assert (L.indices == Lo.indices).all() and (L.indptr == Lo.indptr).all()
    assert abs(L - Lo).max() <= 1e-8 * abs(Lo).max()
    assert abs(sp.triu(L, 1)).sum() == 0 and (L.diagonal() > 0).all()
    if rho_len >= 2.0:
        Qk = (L @ L.T).toarray()
        want = np.linalg.inv(K + 1e-6 * np.eye(n))
        assert np.abs(Qk - want).max() <= 1e-6 * np.abs(want).max()


def test_kl_cholesky_reference_known_answers():
    # test/kl_cholesky/test_sparse_cholesky.jl:14-25: diagonal Theta, full lower pattern
    from gmrfx import klchol
    Theta = np.diag([1.0, 2.0, 3.0, 4.0])
    L = klchol.sparse_approximate_cholesky_inplace(Theta, sp.csc_matrix(np.tril(np.ones((4, 4)))))
    assert abs(sp.triu(L, 1)).sum() == 0
    assert np.abs((L @ L.T).toarray() - np.linalg.inv(Theta)).max() < 1e-4
    # a block that is not positive definite is reported, not silently factored (the reference throws PosDefException)
    bad = np.array([[1.0, 2.0], [2.0, 1.0]])
    with pytest.raises(gmrfx.PosDefException):
        klchol.sparse_approximate_cholesky_inplace(bad, sp.csc_matrix(np.tril(np.ones((2, 2)))))
    with pytest.raises(ValueError):
        klchol._run(np.eye(3), [0, 1, 2, 3], [0, 1], [5], [0, 1], [0], 1e-6, -1)


def test_kl_cholesky_supernodal_matches_oracle():
    from gmrfx import klchol
    X, K, P = _kl_problem(240, 0.2, seed=7)
    n = len(X)
    # supernodes in the spirit of form_supernodes (supernodes.jl:58-90): walk the columns, open a supernode at the
    # first unassigned column j, adopt the unassigned rows of its pattern that lie close to it; the supernode's
    # rows = the union of the members' patterns, descending
    assigned = np.zeros(n, bool)
    col_idx, row_idx = [], []
    for j in range(n):
        if assigned[j]:
            continue
        pj = P.indices[P.indptr[j]:P.indptr[j + 1]]
        members = [i for i in pj if not assigned[i] and np.linalg.norm(X[i] - X[j]) < 0.08]
        assigned[members] = True
        rows = set()
        for k in members:
            rows.update(P.indices[P.indptr[k]:P.indptr[k + 1]].tolist())
        rows.update(members)
        col_idx.append(sorted(members)); row_idx.append(sorted(rows, reverse=True))
    assert max(map(len, col_idx)) > 1
    L = klchol.sparse_approximate_cholesky_supernodal(K, col_idx, row_idx)
    Lo = orc.kl_cholesky_supernodal(K, col_idx, row_idx)
    assert (L.indices == Lo.indices).all() and (L.indptr == Lo.indptr).all()
    assert abs(L - Lo).max() <= 1e-8 * abs(Lo).max()
    # and the resulting precision is a usable GMRF: it enters the hot path and its logdet matches the dense one
    Qk = sp.csc_matrix(L @ L.T)
    ws = GMRFWorkspace(Qk)
    assert abs(ws.logdet() - np.linalg.slogdet(Qk.toarray())[1]) <= 1e-9 * abs(ws.logdet())


def test_kronecker_workspace_matches_dense_kron():
    # Q = kron(Q_t, Q_s) (SeparableModel, separable.jl:143-156: rightmost factor fastest), answered from the two
    # factor-scale workspaces; logdet by the factor rule of precision_logdet (separable.jl:122-141)
    Qt = sp.csc_matrix(spde.ar1_precision(6, 0.9))
    Qs = sp.csc_matrix(spde.matern_precision(spde.grid_mesh_2d(7, 7, jitter=0.2), 0, 0.5))
    kw = gmrfx.KroneckerWorkspace(Qt, Qs)
    Q = sp.kron(Qt, Qs).toarray()
    n = Q.shape[0]
    assert kw.dimension() == n
    assert abs(kw.logdet() - np.linalg.slogdet(Q)[1]) <= 1e-10 * abs(kw.logdet())
    rng = np.random.default_rng(0)
    b = rng.standard_normal(n)
    assert relerr(kw.solve(b), np.linalg.solve(Q, b)) < 1e-10
    Sigma = np.linalg.inv(Q)
    assert relerr(kw.selinv_diag(), np.diag(Sigma)) < 1e-9
    A = np.stack([kw.backward_solve(e) for e in np.eye(n)], axis=1)      # the sampling map z -> x
    assert relerr(A @ A.T, Sigma) < 1e-9
    with pytest.raises(ValueError):
        kw.solve(np.zeros(n + 1))


@pytest.mark.parametrize("dense_max", [4096, 0])
def test_kronecker_workspace_device_path_matches_dense_kron(dense_max):
    """KroneckerWorkspace.solve_dev / backward_solve_dev: the flat device vector is the column-major n2 x n1 panel of the
    sweep over factor 2; the small factor is applied as a dense operator (default) or swept too (DENSE_MAX = 0: the
    transposing fallback for two large factors). Against the dense kron, and bit-compatible with the host-panel path."""
    import torch
    Qt = sp.csc_matrix(spde.ar1_precision(9, 0.9))
    Qs = sp.csc_matrix(spde.matern_precision(spde.grid_mesh_2d(11, 10, jitter=0.2, seed=1), 0, 0.5))
    kw = gmrfx.KroneckerWorkspace(Qt, Qs)
    kw.DENSE_MAX = dense_max
    Q = sp.kron(Qt, Qs).toarray()
    n = Q.shape[0]
    rng = np.random.default_rng(3)
    b = rng.standard_normal(n)
    xb = kw.solve_dev(torch.from_numpy(b).cuda()).cpu().numpy()
    assert relerr(xb, np.linalg.solve(Q, b)) < 1e-10
    assert relerr(xb, kw.solve(b)) < 1e-12
    z = rng.standard_normal(n)
    assert relerr(kw.backward_solve_dev(torch.from_numpy(z).cuda()).cpu().numpy(), kw.backward_solve(z)) < 1e-12
    Z = torch.from_numpy(np.eye(n)).cuda()
    A = np.stack([kw.backward_solve_dev(Z[k].contiguous()).cpu().numpy() for k in range(n)], axis=1)
    assert relerr(A @ A.T, np.linalg.inv(Q)) < 1e-9
    with pytest.raises(ValueError):
        kw.solve_dev(torch.zeros(n + 1, dtype=torch.float64).cuda())


@pytest.mark.parametrize("n1,n2", [(9, 110), (64, 64), (70, 1001), (130, 4099), (512, 777)])
def test_dense_apply_and_transpose_kernels(n1, n2):
    """gmrfx_dense_apply_dev (R = D T, row-major; the dense-operator leg of the Kronecker path, separable.jl:122-172) and
    gmrfx_transpose_dev against a float64 torch reference of the same ops: odd and even widths, sizes that are not multiples of
    the 64 x 64 tiles, more column tiles than one XCD round; bit-reproducible."""
    import torch
    be = gmrfx.MI355XBackend(sp.identity(4, format="csc"))          # any numeric handle of the device
    g = torch.Generator(device="cpu").manual_seed(n1 * 131 + n2)
    D = torch.randn((n1, n1), generator=g, dtype=torch.float64).cuda()
    Tb = torch.zeros(n1 * n2 + 2, dtype=torch.float64).cuda()
    T = Tb[:n1 * n2].view(n1, n2); T.copy_(torch.randn((n1, n2), generator=g, dtype=torch.float64))
    R = torch.full((n1, n2), float("nan"), dtype=torch.float64).cuda()
    torch.cuda.synchronize()
    be.dense_apply_dev(D.data_ptr(), n1, T.data_ptr(), n2, R.data_ptr())
    ref = D.cpu().numpy() @ T.cpu().numpy()
    assert np.abs(R.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max() * n1 ** 0.5
    R2 = torch.empty_like(R)
    be.dense_apply_dev(D.data_ptr(), n1, T.data_ptr(), n2, R2.data_ptr())
    assert torch.equal(R, R2)
    Tt = torch.full((n2, n1), float("nan"), dtype=torch.float64).cuda()
    be.transpose_dev(T.data_ptr(), n1, n2, Tt.data_ptr())
    assert torch.equal(Tt, T.t().contiguous())
    with pytest.raises(ValueError):
        be.dense_apply_dev(D.data_ptr(), n1, T.data_ptr(), n2, T.data_ptr())
    be.close()


@pytest.mark.parametrize("cap", ["64", "128", "256"])
def test_blocked_substitution_in_wide_fronts(cap, monkeypatch):
    """Fronts wider than the inverse cap (2048 columns by default; GMRFX_INV_CAP lowers it here) only hold the
    inverses of their cap-column diagonal blocks: the sweeps substitute block by block inside them, the selected
    inversion completes the inverses on demand. Same answers as with full inverses."""
    monkeypatch.setenv("GMRFX_INV_CAP", cap)
    rng = np.random.default_rng(int(cap))
    m3 = spde.grid_mesh_3d(13, 12, 11)
    cases = [sp.csc_matrix(np.cov(rng.standard_normal((330, 900))) + np.eye(330)),            # one dense 330-column front
             sp.csc_matrix(spde.matern_precision(spde.grid_mesh_2d(70, 66, jitter=0.2), 0, 0.3)),
             sp.csc_matrix(spde.matern_precision(m3, 0, 0.5))]
    kws = [{}, {}, {"coords": m3.points}]
    for k, (Q, kw) in enumerate(zip(cases, kws)):
        ws = GMRFWorkspace(Q, **kw)
        if k == 0:
            assert ws.backend.stats()["max_cols"] > int(cap)      # the dense case is blocked at every cap tried here
        F = orc.OracleFactor(Q, ws.backend.ordering_permutation())
        n = Q.shape[0]
        for nrhs in (1, 64, 70):
            B = rng.standard_normal((n, nrhs))
            assert relerr(ws.workspace_solve(B), F.solve(B)) < 1e-10
            assert relerr(ws.backward_solve(B), F.backward_solve(B)) < 1e-10
        assert relerr(ws.selinv_diag(), F.selinv_diag()) < 1e-8
        B = rng.standard_normal((n, 5))                   # a solve after the inverses have been completed
        assert relerr(ws.workspace_solve(B), F.solve(B)) < 1e-10
        ws.update_precision(sp.csc_matrix(1.5 * Q))       # refactorise: back to capped inverses
        assert relerr(ws.workspace_solve(B), F.solve(B) / 1.5) < 1e-10


@pytest.mark.parametrize("name", ["matern64_coords", "cfg1_alpha3_65x65", "matern3d_10", "rand400", "natural_chain"])
def test_sweep_tasks_on_and_off_agree(name, monkeypatch):
    """Sweep tasks (whole bottom subtrees on an LDS-resident local vector, csrc/sweep_chunk.hip; the default) against
    the pure level schedule (GMRFX_SWEEP_TASK_ROWS=0): same solves / backward solves to rounding (the summation order
    of the updates differs), for full and ragged right-hand-side blocks, and both against the oracle."""
    Q, kw = next((sp.csc_matrix(q), k) for n, q, k in CASES if n == name)
    n = Q.shape[0]
    rng = np.random.default_rng(5)
    on = gmrfx.MI355XBackend(Q, **kw)
    assert on.sweep_tasks()[1].size > 0
    monkeypatch.setenv("GMRFX_SWEEP_TASK_ROWS", "0")
    off = gmrfx.MI355XBackend(Q, ordering=on.ordering_permutation())
    assert off.sweep_tasks()[1].size == 0
    monkeypatch.delenv("GMRFX_SWEEP_TASK_ROWS")
    F = orc.OracleFactor(Q, on.ordering_permutation())
    for nrhs in (1, 5, 64, 70):
        B = rng.standard_normal((n, nrhs))
        Xo = F.solve(B)
        assert relerr(on.backend_solve(B), Xo) < 1e-10
        assert relerr(on.backend_solve(B), off.backend_solve(B)) < 1e-11
        Zo = F.backward_solve(B)
        assert relerr(on.backend_backward_solve(B), Zo) < 1e-10
        assert relerr(on.backend_backward_solve(B), off.backend_backward_solve(B)) < 1e-11
    assert np.array_equal(on.backend_solve(B), on.backend_solve(B))        # still bit-reproducible


@pytest.mark.parametrize("name", ["matern64_coords", "cfg1_alpha3_65x65", "matern3d_10", "rand400", "tall_fronts", "dense70"])
def test_backward_step_one_workgroup_per_front_equals_two_launches(name, monkeypatch):
    """Round 5 (csrc/sweep_front.hip): the backward step of a front of at most 128 columns as ONE workgroup -- trailing rows of x
    gathered once through LDS, t = y - L21' x and x = L11^-T t in the same launch -- against the two launches it replaces
    (GMRFX_BWD_FRONT=0) and against the oracle: solves and backward-only solves (y in the second buffer / in place), full and
    ragged right-hand-side blocks. By default only levels with at least 192 such fronts take the new path (the 10^6-node
    problems of the full-size tests); GMRFX_BWD_FRONT=1 sends every eligible front of these small cases through it."""
    Q, kw = next((sp.csc_matrix(q), k) for n, q, k in CASES if n == name)
    n = Q.shape[0]
    rng = np.random.default_rng(11)
    monkeypatch.setenv("GMRFX_BWD_FRONT", "1")
    one = gmrfx.MI355XBackend(Q, **kw)
    monkeypatch.setenv("GMRFX_BWD_FRONT", "0")
    two = gmrfx.MI355XBackend(Q, ordering=one.ordering_permutation())
    monkeypatch.delenv("GMRFX_BWD_FRONT")
    F = orc.OracleFactor(Q, one.ordering_permutation())
    for nrhs in (1, 3, 17, 64, 70):
        B = rng.standard_normal((n, nrhs))
        assert relerr(one.backend_solve(B), F.solve(B)) < 1e-10
        assert relerr(one.backend_solve(B), two.backend_solve(B)) < 1e-11
        assert relerr(one.backend_backward_solve(B), F.backward_solve(B)) < 1e-10
        assert relerr(one.backend_backward_solve(B), two.backend_backward_solve(B)) < 1e-11
    assert np.array_equal(one.backend_solve(B), one.backend_solve(B))       # bit-reproducible


@pytest.mark.parametrize("name", ["matern64_coords", "cfg1_alpha3_65x65", "matern3d_10", "rand400", "tall_fronts", "dense70"])
def test_forward_step_one_workgroup_per_front_equals_three_launches(name, monkeypatch):
    """Round 6 (csrc/sweep_front.hip, k_fwd_front): the WHOLE forward step of a front of at most 128 columns as one workgroup -- own rows
    assembled from X and the children's update vectors, y = L11^-1 b, W = children - L21 y with the children's rows entering through the
    matrix pipe -- against the three launches it replaces (GMRFX_FWD_FRONT=0) and against the oracle: solves of 33 .. 70 right-hand
    sides (narrower passes keep the narrow kernels), ragged blocks, bit-reproducible. By default only levels with at least 192 such
    fronts take it (the 10^6-node problems of the full-size tests); GMRFX_FWD_FRONT=1 sends every eligible front of these cases."""
    Q, kw = next((sp.csc_matrix(q), k) for n, q, k in CASES if n == name)
    n = Q.shape[0]
    rng = np.random.default_rng(12)
    monkeypatch.setenv("GMRFX_FWD_FRONT", "1")
    one = gmrfx.MI355XBackend(Q, **kw)
    monkeypatch.setenv("GMRFX_FWD_FRONT", "0")
    three = gmrfx.MI355XBackend(Q, ordering=one.ordering_permutation())
    monkeypatch.delenv("GMRFX_FWD_FRONT")
    F = orc.OracleFactor(Q, one.ordering_permutation())
    for nrhs in (33, 40, 64, 70, 1, 17):
        B = rng.standard_normal((n, nrhs))
        assert relerr(one.backend_solve(B), F.solve(B)) < 1e-10
        assert relerr(one.backend_solve(B), three.backend_solve(B)) < 1e-11
    assert np.array_equal(one.backend_solve(B), one.backend_solve(B))       # bit-reproducible
    B = rng.standard_normal((n, 64))
    assert np.array_equal(one.backend_solve(B), one.backend_solve(B))
    # the pipelined call takes the same kernels: same bits as the two calls
    X1 = one.refactorize_solve(Q.data, B)
    assert np.array_equal(X1, one.backend_solve(B))


@pytest.mark.parametrize("cap", ["64", "128"])
def test_one_workgroup_backward_step_beside_blocked_substitution(cap, monkeypatch):
    """A level that has BOTH fronts wider than the inverse cap (blocked substitution, block 0's list = 'every big front') and a
    narrow tail k_bwd_front has already finished: the blocked loop must not run over that tail again (round-5 advisor finding:
    na = L.active[0] ignored the fronts taken off the list -- x overwritten with L11^-T of the already final x). Passes wider than
    16 columns, y in the second buffer (solve) and in place (backward-only), against the oracle."""
    monkeypatch.setenv("GMRFX_BWD_FRONT", "1")
    monkeypatch.setenv("GMRFX_FWD_FRONT", "1")          # (round 6: the forward twin has the same list arithmetic)
    monkeypatch.setenv("GMRFX_INV_CAP", cap)
    rng = np.random.default_rng(100 + int(cap))
    m3 = spde.grid_mesh_3d(13, 12, 11)
    # block-diagonal: one dense 330-column front beside thirty dense 100-column fronts (too many rows for the fused small-front
    # kernels, too wide for the sweep tasks, narrow enough for k_bwd_front at cap 128) -- wide and narrow on one level
    wide = np.cov(rng.standard_normal((330, 900))) + np.eye(330)
    blocks = [wide] + [np.cov(rng.standard_normal((100, 300))) + np.eye(100) for _ in range(30)]
    cases = [(sp.csc_matrix(sp.block_diag(blocks)), {}),
             (sp.csc_matrix(spde.matern_precision(spde.grid_mesh_2d(70, 66, jitter=0.2), 0, 0.3)), {}),
             (sp.csc_matrix(spde.matern_precision(m3, 0, 0.5)), {"coords": m3.points})]
    for Q, kw in cases:
        be = gmrfx.MI355XBackend(Q, **kw)
        F = orc.OracleFactor(Q, be.ordering_permutation())
        n = Q.shape[0]
        for nrhs in (17, 64, 70):
            B = rng.standard_normal((n, nrhs))
            assert relerr(be.backend_solve(B), F.solve(B)) < 1e-10
            assert relerr(be.backend_backward_solve(B), F.backward_solve(B)) < 1e-10
        be.close()


@pytest.mark.parametrize("name", ["matern64_coords", "cfg1_alpha3_65x65", "matern3d_10", "rand400"])
def test_tile_records_and_plain_grid_give_the_same_bits(name, monkeypatch):
    """The record-driven kernels (contribution-block SYRK and forward update: one self-contained record per tile, handed
    out in per-XCD runs, csrc/kernels.hip k_syrk_cb_rec / k_fwd_update_rec; the default) against the plain-grid forms
    they replaced (GMRFX_SYRK_XCD=0): the SAME factor bit for bit (same sums in the same order), the same solve bits,
    and the factor against the oracle entry by entry."""
    Q, kw = next((sp.csc_matrix(q), k) for n, q, k in CASES if n == name)
    n = Q.shape[0]
    rec = gmrfx.MI355XBackend(Q, **kw)
    monkeypatch.setenv("GMRFX_SYRK_XCD", "0")
    grid = gmrfx.MI355XBackend(Q, ordering=rec.ordering_permutation())
    monkeypatch.delenv("GMRFX_SYRK_XCD")
    a, b = rec.factor_values(), grid.factor_values()          # before any solve (solves fill the dense inverses in)
    assert np.array_equal(a, b)
    F = orc.OracleFactor(Q, rec.ordering_permutation())
    Lg, Lo = rec.factor_csc(), F.L()
    assert abs(Lg - Lo).max() <= 1e-10 * abs(Lo).max()
    B = np.random.default_rng(9).standard_normal((n, 37))     # odd count: the pair loads' last right-hand side is alone
    Xr, Xg = rec.backend_solve(B), grid.backend_solve(B)
    assert np.array_equal(Xr, Xg)
    assert relerr(Xr, F.solve(B)) < 1e-10


def test_tile_records_and_plain_grid_give_the_same_bits_on_a_mesh_with_many_tiles(monkeypatch):
    """The same on a 300 x 300-node mesh (9e4 unknowns): levels with hundreds of big fronts and more than 128 update
    tiles, so that k_syrk_cb_rec, k_fwd_update_rec and the 8-wave backward variant all run; factor and solves bit for
    bit equal to the plain-grid forms, residual against Q."""
    mesh = spde.grid_mesh_2d(300, 300, jitter=0.25, seed=3)
    Q = sp.csc_matrix(spde.matern_precision(mesh, 0, 0.2))
    n = Q.shape[0]
    rec = gmrfx.MI355XBackend(Q, coords=mesh.points)
    monkeypatch.setenv("GMRFX_SYRK_XCD", "0")
    grid = gmrfx.MI355XBackend(Q, ordering=rec.ordering_permutation())
    monkeypatch.delenv("GMRFX_SYRK_XCD")
    assert np.array_equal(rec.factor_values(), grid.factor_values())
    assert rec.compute_logdet() == grid.compute_logdet()
    B = np.random.default_rng(10).standard_normal((n, 64))
    Xr, Xg = rec.backend_solve(B), grid.backend_solve(B)
    assert np.array_equal(Xr, Xg)
    assert np.linalg.norm(Q @ Xr - B) / np.linalg.norm(B) < 1e-10
    Z = B[:, :7]
    assert np.array_equal(rec.backend_backward_solve(Z), grid.backend_backward_solve(Z))


@pytest.mark.parametrize("name", ["matern64_coords", "cfg1_alpha3_65x65", "matern3d_10", "rand400", "mesh300"])
def test_pipelined_syrk_product_loop_gives_the_same_bits(name, monkeypatch):
    """Round 6: k_syrk_cb_rec<true> runs the product of a contribution-block tile as three stages of two k-steps, each requested two
    stages ahead (levels whose widest front has >= 128 columns by default). GMRFX_SYRK_PIPED=0 sends EVERY level through it (fronts of
    1 .. 64 columns too: ranges shorter than one round, column counts that are no multiple of 4, the clamped requests behind the last
    k-step), a huge value none: the same factor bit for bit in all three, and against the oracle."""
    if name == "mesh300":
        mesh = spde.grid_mesh_2d(300, 300, jitter=0.25, seed=3)
        Q, kw = sp.csc_matrix(spde.matern_precision(mesh, 0, 0.2)), {"coords": mesh.points}
    else:
        Q, kw = next((sp.csc_matrix(q), k) for n, q, k in CASES if n == name)
    dflt = gmrfx.MI355XBackend(Q, **kw)
    perm = dflt.ordering_permutation()
    monkeypatch.setenv("GMRFX_SYRK_PIPED", "0")
    every = gmrfx.MI355XBackend(Q, ordering=perm)
    monkeypatch.setenv("GMRFX_SYRK_PIPED", "1000000000")
    none = gmrfx.MI355XBackend(Q, ordering=perm)
    monkeypatch.delenv("GMRFX_SYRK_PIPED")
    a = dflt.factor_values()
    assert np.array_equal(a, every.factor_values()) and np.array_equal(a, none.factor_values())
    assert dflt.compute_logdet() == every.compute_logdet() == none.compute_logdet()
    if name != "mesh300":
        F = orc.OracleFactor(Q, perm)
        Lg, Lo = every.factor_csc(), F.L()
        assert abs(Lg - Lo).max() <= 1e-10 * abs(Lo).max()
    for be in (dflt, every, none):
        be.close()


@pytest.mark.parametrize("mode", ["wg", "wave"])
@pytest.mark.parametrize("nrhs", [1, 17, 33, 64])
def test_sweep_task_forms_match_the_oracle(mode, nrhs, monkeypatch):
    """The two forms of the bottom-subtree sweep tasks -- 16-wave workgroups on a 32-column local vector (sweep_task.hip)
    and one wave per (task, 16 columns) with an op pipeline (sweep_wave.hip; the default up to 32 right-hand sides) --
    forced for every width through GMRFX_TASK_MODE: full solves and backward solves against the oracle on the same
    permutation, on a mesh whose tasks include wide fronts (17 .. 64 columns), partial column tiles and both LDS classes."""
    mesh = spde.grid_mesh_2d(150, 150, jitter=0.25, seed=11)
    Q = sp.csc_matrix(spde.matern_precision(mesh, 0, 0.2))
    n = Q.shape[0]
    monkeypatch.setenv("GMRFX_TASK_MODE", mode)
    be = gmrfx.MI355XBackend(Q, coords=mesh.points)
    monkeypatch.delenv("GMRFX_TASK_MODE")
    cap, first, last, _ = be.sweep_tasks()
    assert len(first) > 50 and cap == 288
    F = orc.OracleFactor(Q, be.ordering_permutation())
    B = np.random.default_rng(100 + nrhs).standard_normal((n, nrhs))
    assert relerr(be.backend_solve(B), F.solve(B)) < 1e-10
    Z = B[:, : min(nrhs, 5)]
    assert relerr(be.backend_backward_solve(Z), F.backward_solve(Z)) < 1e-10
    # bit-reproducible: the same call twice gives the same bits
    assert np.array_equal(be.backend_solve(B), be.backend_solve(B))


def test_narrow_passes_match_the_oracle_for_every_width():
    """Round 5: passes of at most 16 right-hand sides have kernels of their own on every level of the tree -- the bottom tasks with a
    local vector of 2 / 4 / 8 / 16 stored columns (sweep_wave.hip), one launch of all tasks up to 4 columns; the level fronts through
    k_fwd_update_wave / k_bwd_wave (one wave per tile, four waves sharing the K range of fronts wider than 128 columns / with more
    than 256 trailing rows), k_xmul_narrow, k_permute_narrow (up to 8 columns). Solve and backward solve VALUE BY VALUE against the
    oracle for every width class and its edges, on a mesh whose top fronts have several hundred columns and trailing rows; then the
    same columns through ONE 64-column pass (other kernels, other summation orders) to 1e-12."""
    mesh = spde.grid_mesh_2d(230, 210, jitter=0.25, seed=19)
    Q = sp.csc_matrix(spde.matern_precision(mesh, 0, 0.2))
    n = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, coords=mesh.points)
    sym = be.symbolic()
    cols = np.diff(sym.super_first)
    trail = np.diff(sym.row_ptr) - cols
    assert cols.max() > 128 and trail.max() > 256          # the four-wave forms of both update kernels are exercised
    F = orc.OracleFactor(Q, be.ordering_permutation())
    # (round 6: passes of 17 .. 32 columns run the narrow FORWARD kernels on two right-hand-side tiles, grid y / z = the tile; 33 is
    #  the first width on the 64-column kernels throughout)
    B = np.random.default_rng(77).standard_normal((n, 33))
    Xo, So = F.solve(B), F.backward_solve(B)
    X64 = be.backend_solve(np.hstack([B, B[:, :31]]))[:, :33]
    for k in (1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 18, 24, 31, 32, 33):
        Bk = B[:, 0] if k == 1 else B[:, :k]
        X = be.backend_solve(Bk).reshape(n, -1)
        assert relerr(X, Xo[:, :k]) < 1e-10, k
        assert relerr(X, X64[:, :k]) < 1e-12, k
        S = be.backend_backward_solve(Bk).reshape(n, -1)
        assert relerr(S, So[:, :k]) < 1e-10, k
        assert np.array_equal(X, be.backend_solve(Bk).reshape(n, -1)), k      # bit-reproducible
    be.close()


@pytest.mark.parametrize("mesh_kind", ["2d", "3d"])
def test_refactorize_solve_pipelined_equals_separate_calls(mesh_kind):
    """gmrfx_refactorize_solve[_dev] (workspace_solve on a workspace with new values, gmrf_workspace.jl:170-178 + 207-215; the
    forward sweep follows the factorisation up the tree on a second stream) against gmrfx_refactorize + gmrfx_solve on a second
    handle: factor, X and log-determinant bit for bit, for 1 / 17 / 64 / 100 right-hand sides (wave tasks, workgroup tasks, a
    second pass), repeated (events and buffers are reused), then the oracle's solve to 1e-10, a plain solve on the pipelined
    handle afterwards (its dense inverses must be complete) and the pivot report of an indefinite matrix."""
    import torch
    if mesh_kind == "2d":
        mesh = spde.grid_mesh_2d(150, 140, jitter=0.25, seed=5)
        Q = spde.matern_precision(mesh, 0, 0.2)
    else:
        mesh = spde.grid_mesh_3d(22, 21, 20)
        Q = spde.matern_precision(mesh, 0, 0.4)
    n = Q.shape[0]
    dev = torch.device("cuda", 0)
    a = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
    b = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
    rng = np.random.default_rng(8)
    for rep, nrhs in enumerate((64, 1, 17, 100, 64)):
        scale = 1.0 + 0.25 * rep                                  # new values every time
        d_nz = torch.from_numpy(np.ascontiguousarray(Q.data * scale)).to(dev)
        Bh = rng.standard_normal((nrhs, n))
        d_B = torch.from_numpy(Bh).to(dev)
        d_Xa, d_Xb = torch.zeros_like(d_B), torch.zeros_like(d_B)
        torch.cuda.synchronize()
        assert a.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, nrhs, d_Xa.data_ptr(), n) == 0
        assert b.refactorize_dev(d_nz.data_ptr()) == 0
        b.solve_dev(d_B.data_ptr(), n, nrhs, d_Xb.data_ptr(), n)
        torch.cuda.synchronize()
        assert torch.equal(d_Xa, d_Xb), f"nrhs={nrhs}: pipelined and separate solves differ by {float((d_Xa - d_Xb).abs().max()):.3e}"
        assert np.array_equal(a.factor_values(), b.factor_values())
        assert a.compute_logdet() == b.compute_logdet()
        resid = np.linalg.norm((Q * scale) @ d_Xa.cpu().numpy().T - Bh.T) / np.linalg.norm(Bh)
        assert resid < 1e-10
        # a plain solve on the pipelined handle: same bits again (the per-level inverses are all there)
        d_Xc = torch.zeros_like(d_B)
        a.solve_dev(d_B.data_ptr(), n, nrhs, d_Xc.data_ptr(), n)
        torch.cuda.synchronize()
        assert torch.equal(d_Xc, d_Xb)
    # selected inversion after a pipelined call (it needs the full inverses of the diagonal blocks)
    assert np.array_equal(a.get_selinv_diag(), b.get_selinv_diag())
    # host-array form, against the oracle
    F = orc.OracleFactor(Q, a.ordering_permutation())
    Bh = rng.standard_normal((n, 3))
    X = a.refactorize_solve(Q.data, Bh)
    Xo = F.solve(Bh)
    assert np.abs(X - Xo).max() <= 1e-10 * np.abs(Xo).max()
    # an indefinite matrix: the pivot is reported like gmrfx_refactorize reports it, nothing hangs
    bad = Q.copy().tocsc(); bad.sort_indices()
    vals = bad.data.copy()
    col = n // 2
    k = bad.indptr[col] + int(np.searchsorted(bad.indices[bad.indptr[col]:bad.indptr[col + 1]], col))
    vals[k] = -abs(vals[k])
    d_bad = torch.from_numpy(vals).to(dev)
    d_B = torch.from_numpy(rng.standard_normal((4, n))).to(dev); d_X = torch.zeros_like(d_B)
    torch.cuda.synchronize()
    ia = a.refactorize_solve_dev(d_bad.data_ptr(), d_B.data_ptr(), n, 4, d_X.data_ptr(), n)
    ib = b.refactorize_dev(d_bad.data_ptr())
    assert ia == ib > 0
    a.close(); b.close()


def test_refactorize_solve_host_io_paths_equal_device_call():
    """gmrfx_refactorize_solve with HOST right-hand sides (what the reference's seam hands over, backend.jl:207-209): the
    transfers are serial -- B goes up on a copy stream IN FRONT of the factorisation, the result leaves in slices behind the backward
    sweep (Device::host_upload / host_download) -- only the host-side staging is overlapped.
    Pageable arrays (staged through the handle's page-locked buffer by several host threads, more than one 32 MB slice),
    page-locked arrays (direct DMA) and a leading dimension larger than n must all give the bits of the device-resident call."""
    import torch
    mesh = spde.grid_mesh_2d(300, 290, jitter=0.25, seed=7)
    Q = spde.matern_precision(mesh, 0, 0.2)
    n = Q.shape[0]
    dev = torch.device("cuda", 0)
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
    rng = np.random.default_rng(3)
    nrhs = 100                                                      # 87 000 x 100 doubles = 70 MB: three staging slices, two passes of the sweeps
    nz = np.ascontiguousarray(Q.data * 1.5)
    Bf = np.asfortranarray(rng.standard_normal((n, nrhs)))
    d_nz = torch.from_numpy(nz).to(dev)
    d_B = torch.from_numpy(np.ascontiguousarray(Bf.T)).to(dev)
    d_X = torch.zeros_like(d_B)
    assert be.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, nrhs, d_X.data_ptr(), n) == 0
    torch.cuda.synchronize()
    ref = d_X.cpu().numpy().T                                       # n x nrhs
    assert np.linalg.norm((Q * 1.5) @ ref - Bf) / np.linalg.norm(Bf) < 1e-10
    # pageable, contiguous
    Xf = np.zeros_like(Bf, order="F")
    for _ in range(2):                                              # (second call: staging buffer and events reused)
        Xf[:] = 0.0
        assert be.refactorize_solve_ptr(nz.ctypes.data, Bf.ctypes.data, n, nrhs, Xf.ctypes.data, n) == 0
        assert np.array_equal(Xf, ref)
    # pageable, leading dimensions larger than n (views into taller arrays)
    Bt = np.asfortranarray(np.zeros((n + 5, nrhs))); Bt[:n] = Bf
    Xt = np.asfortranarray(np.full((n + 3, nrhs), -7.0))
    assert be.refactorize_solve_ptr(nz.ctypes.data, Bt.ctypes.data, n + 5, nrhs, Xt.ctypes.data, n + 3) == 0
    assert np.array_equal(Xt[:n], ref) and np.all(Xt[n:] == -7.0)
    # page-locked: (nrhs, n) row-major = column-major n x nrhs
    Bp = torch.from_numpy(np.ascontiguousarray(Bf.T)).pin_memory()
    Xp = torch.zeros_like(Bp).pin_memory()
    assert be.refactorize_solve_ptr(nz.ctypes.data, Bp.data_ptr(), n, nrhs, Xp.data_ptr(), n) == 0
    assert np.array_equal(Xp.numpy().T, ref)
    be.close()


def test_full_size_cfg2_pipelined_equals_separate_calls():
    """The call bench.py times, at the size it is timed (BASELINE cfg 2: 1000 x 1000 nodes): gmrfx_refactorize_solve_dev against
    gmrfx_refactorize_dev + gmrfx_solve_dev on a second handle -- factor, X and log-determinant bit for bit for 64 and 17
    right-hand sides (the gate that holds the bottom of the forward sweep back sits at a different level of an 18-level tree
    than of the 10-level trees of the small cases), and the residual of the pipelined X below 1e-9."""
    import torch
    m = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
    Q = spde.matern_precision(m, 0, 0.2)
    n = Q.shape[0]
    dev = torch.device("cuda", 0)
    a = gmrfx.MI355XBackend(Q, coords=m.points, factorize=False)
    b = gmrfx.MI355XBackend(Q, ordering=a.ordering_permutation(), factorize=False)
    g = torch.Generator(device="cpu").manual_seed(1)
    for rep, nrhs in enumerate((64, 17)):
        scale = 1.0 + 0.5 * rep
        d_nz = torch.from_numpy(np.ascontiguousarray(Q.data * scale)).to(dev)
        Bh = torch.randn((nrhs, n), generator=g, dtype=torch.float64)
        d_B = Bh.to(dev)
        d_Xa, d_Xb = torch.zeros_like(d_B), torch.zeros_like(d_B)
        torch.cuda.synchronize()
        for _ in range(2):                                          # twice: the second call reuses events, buffers and inverses' storage
            assert a.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, nrhs, d_Xa.data_ptr(), n) == 0
        assert b.refactorize_dev(d_nz.data_ptr()) == 0
        b.solve_dev(d_B.data_ptr(), n, nrhs, d_Xb.data_ptr(), n)
        torch.cuda.synchronize()
        assert torch.equal(d_Xa, d_Xb), f"nrhs={nrhs}: pipelined and separate solves differ by {float((d_Xa - d_Xb).abs().max()):.3e}"
        assert a.compute_logdet() == b.compute_logdet()
        assert np.array_equal(a.factor_values(), b.factor_values())
        X = d_Xa.cpu().numpy().T
        Bn = Bh.numpy().T
        assert np.linalg.norm((Q * scale) @ X - Bn) / np.linalg.norm(Bn) < 1e-9
    a.close(); b.close()


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80, 97, 128, 129, 150, 200])
def test_dense_fronts_of_every_width_through_the_diagonal_block_kernel(n, monkeypatch):
    """One dense supernode of n columns on the generic big-front path (GMRFX_SMALL_ROWS=0: no fused small-front kernel, no subtree
    tasks): the 64 x 64 diagonal-block kernel (k_potrf64_b: 16-column steps, one diagonal wave, helper waves;
    csrc/potrf64_blocked.h) sees every number of 16 x 16 steps and every partial last block (n mod 64, n mod 16). Against
    numpy's Cholesky of the permuted matrix: factor entry by entry, log-determinant, a solve; and a non-positive pivot planted in
    three different columns is reported at exactly that column (info = column + 1 in the elimination order, LAPACK / CHOLMOD style:
    /root/reference/src/workspace/backend.jl:184, cholesky!(...; check = false))."""
    rng = np.random.default_rng(100 + n)
    A = rng.standard_normal((n, n))
    Qd = A @ A.T + n * np.eye(n)
    Q = sp.csc_matrix(Qd)
    monkeypatch.setenv("GMRFX_SMALL_ROWS", "0")
    be = gmrfx.MI355XBackend(Q, ordering="natural")
    assert be.last_info == 0 and be.stats()["n_small_fronts"] == 0
    Lref = np.linalg.cholesky(Qd)
    Lg = be.factor_csc().toarray()
    assert abs(np.tril(Lg) - Lref).max() <= 1e-12 * abs(Lref).max()
    assert abs(be.compute_logdet() - 2.0 * np.log(np.diag(Lref)).sum()) <= 1e-12 * n * max(1.0, np.log(n + 1.0))
    B = rng.standard_normal((n, 3))
    assert relerr(be.backend_solve(B), np.linalg.solve(Qd, B)) < 1e-11
    # the first non-positive pivot: make the leading (k + 1) x (k + 1) minor indefinite by lowering entry (k, k) below its Schur bound
    for k in sorted({0, n // 2, n - 1}):
        Qb = Qd.copy()
        if k == 0:
            Qb[0, 0] = -1.0
        else:
            v = np.linalg.solve(Lref[:k, :k], Qd[:k, k])
            Qb[k, k] = float(v @ v) - 0.5          # pivot k becomes -0.5, the pivots before it are untouched
        be.refactorize(sp.csc_matrix(Qb))
        assert be.last_info == k + 1, (n, k, be.last_info)
    be.refactorize(Q)
    assert be.last_info == 0
    be.close()


def test_one_dense_front_wide_enough_for_the_staged_panel_update():
    """Round 5 (csrc/kernels.hip, k_gemm_nt_big): fronts with at least 4096 rows below a 256-column outer block take the K = 256
    panel update on 128 x 128 LDS-staged tiles (the top fronts of 3-D problems: cfg 4). A dense SPD matrix of 4700 unknowns is ONE
    front of that kind (its first outer blocks qualify, the rest fall back to the direct-operand tiles): solve and log-determinant
    against LAPACK."""
    n = 4700
    rng = np.random.default_rng(9)
    G = rng.standard_normal((n, 64))
    A = G @ G.T / 64.0 + np.diag(1.0 + rng.random(n))
    Q = sp.csc_matrix(A)
    be = gmrfx.MI355XBackend(Q, ordering="natural", device=0)
    assert be.last_info == 0
    B = rng.standard_normal((n, 3))
    Lc = np.linalg.cholesky(A)
    X = np.linalg.solve(Lc.T, np.linalg.solve(Lc, B))
    assert relerr(be.backend_solve(B), X) < 1e-10
    assert abs(be.compute_logdet() - 2.0 * np.log(np.diag(Lc)).sum()) < 1e-9 * n
    be.close()


def test_two_huge_fronts_take_the_two_pass_contribution_product():
    """Round 5 (csrc/kernels.hip, k_syrk_big): on a level whose fronts have >= 1024 columns and >= 4096 rows below them the
    contribution block is built in two passes -- the children's extend-add alone (k_syrk_cb_rec, noprod), then CB -= L21 L21' on
    128 x 128 LDS-staged tiles. Two dense 1024-column blocks that only meet through a dense 4100-column block (4100: a ragged
    last tile) are two such fronts under the natural ordering; solve and log-determinant against LAPACK."""
    c, m = 1024, 4100
    n = 2 * c + m
    rng = np.random.default_rng(10)
    G1 = rng.standard_normal((n, 24))
    G2 = rng.standard_normal((n, 24))
    G1[c:2 * c] = 0.0                           # the first block never meets the second: their coupling is exactly zero
    G2[:c] = 0.0
    A = (G1 @ G1.T + G2 @ G2.T) / 48.0 + np.diag(1.0 + rng.random(n))
    assert not A[:c, c:2 * c].any()
    Q = sp.csc_matrix(A)
    be = gmrfx.MI355XBackend(Q, ordering="natural", device=0)
    assert be.last_info == 0
    sym = be.symbolic()
    cols = np.diff(sym.super_first)
    trail = np.diff(sym.row_ptr) - cols
    assert np.count_nonzero((cols >= 1024) & (trail >= 4096)) == 2, (cols, trail)
    B = rng.standard_normal((n, 2))
    Lc = np.linalg.cholesky(A)
    X = np.linalg.solve(Lc.T, np.linalg.solve(Lc, B))
    assert relerr(be.backend_solve(B), X) < 1e-10
    assert abs(be.compute_logdet() - 2.0 * np.log(np.diag(Lc)).sum()) < 1e-9 * n
    # the selected inversion of the same fronts (csrc/selinv.hip, k_sel_z21_big: Z21 on 128 x 128 staged tiles): diagonal of A^-1
    be.compute_selinv()
    Li = np.linalg.solve(Lc, np.eye(n))
    assert relerr(be.get_selinv_diag(), np.einsum("ij,ij->j", Li, Li)) < 1e-8
    be.close()


def test_refactorize_logpdf_one_call_equals_three_calls():
    """gmrfx_refactorize_logpdf_dev (one evaluation of the hyper-parameter loop: factorisation, r'Qr beside it on the side stream,
    log-determinant behind it, one synchronisation) against gmrfx_refactorize_dev + gmrfx_quadform_dev + gmrfx_logdet: quadratic
    forms, log-determinant and factor bit for bit, with and without a mean, for 1 and 5 vectors; and the logpdf against the oracle."""
    import torch
    mesh = spde.grid_mesh_2d(140, 130, jitter=0.25, seed=6)
    Q = sp.csc_matrix(spde.matern_precision(mesh, 0, 0.2))
    n = Q.shape[0]
    dev = torch.device("cuda", 0)
    a = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
    b = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
    rng = np.random.default_rng(4)
    for rep, nvec in enumerate((1, 5, 1)):
        d_nz = torch.from_numpy(np.ascontiguousarray(Q.data * (1.0 + 0.5 * rep))).to(dev)
        Zh = rng.standard_normal((nvec, n))
        d_Z = torch.from_numpy(Zh).to(dev)
        mu = rng.standard_normal(n) if rep == 1 else None
        d_mu = torch.from_numpy(mu).to(dev) if mu is not None else None
        torch.cuda.synchronize()
        q1, ld1 = a.refactorize_logpdf_dev(d_nz.data_ptr(), d_Z.data_ptr(), n, nvec, d_mu.data_ptr() if d_mu is not None else 0)
        assert b.refactorize_dev(d_nz.data_ptr()) == 0
        q2 = b.quadform_dev(d_nz.data_ptr(), d_Z.data_ptr(), n, nvec, d_mu.data_ptr() if d_mu is not None else 0)
        ld2 = b.compute_logdet()
        assert a.last_info == 0 and np.array_equal(q1, q2) and ld1 == ld2
        assert np.array_equal(a.factor_values(), b.factor_values())
        assert a.compute_logdet() == ld1                          # kept for this factorisation
        Qk = Q * (1.0 + 0.5 * rep)
        F = orc.OracleFactor(sp.csc_matrix(Qk), a.ordering_permutation())
        for k in range(nvec):
            lp = -0.5 * q1[k] + 0.5 * ld1 - 0.5 * n * np.log(2.0 * np.pi)
            lpo = orc.logpdf(F, sp.csc_matrix(Qk), Zh[k], mu)
            assert abs(lp - lpo) <= 1e-10 * abs(lpo)
    a.close(); b.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_spacetime_rehearsal_on_one_gpu(world):
    """SURVEY 8 f3, "distributed separators": the space-time posterior precision kron(Q_t, Q_s) + diag(h) (block tridiagonal in
    time; SeparableModel hands it whole to the solver, separable.jl:143-172) factored ONCE over `world` processes on the space-time
    nested dissection -- time-slab / space separators at the top of the tree are distributed fronts (256-column blocks dealt over
    their group, look-ahead broadcasts). Panels and log-determinant bit for bit against the unsharded handle, solves, backward
    solves and the selected-inverse diagonal to rounding; all ranks on this one GPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_gpu_worker, args=(r, world, 29850 + world, q, "spacetime")) for r in range(world)]
    [p.start() for p in procs]
    got = []
    for _ in range(world):
        item = q.get(timeout=300)
        assert item[0] != "error", f"rank {item[1]} failed:\n{item[2]}"
        got.append(item)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ld, ld_ref, same, nmine, info in got:
        assert same and nmine > 0 and info["n_dist"] >= 1, f"rank {rank}: {info.get('why')}"
        assert abs(ld - ld_ref) <= 1e-12 * abs(ld_ref)
