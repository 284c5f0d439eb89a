"""Parity with the reference's OWN arithmetic: the vectors `julia bench/cholmod_baseline.jl <in> <out>` writes -- what
`cholesky(Symmetric(Q); perm)`, `F \\ B`, `logdet(F)`, `F.UP \\ z`, `sparse(F.L)` and `SelectedInversion.selinv_diag(F)`
(reference call sites: src/workspace/backend.jl:148-149, 208, 212, 283, 253) return on the deterministic inputs under
tests/golden/cholmod_inputs/ (made by tests/golden/make_cholmod_inputs.py with THIS backend's permutation).

No Julia exists in the authoring image or on the GPU boxes so far, so the output files are absent and these tests SKIP, saying
so: the oracle stays "parity unpinned"  until somebody with Julia runs the two commands in
make_cholmod_inputs.py's header and commits tests/golden/cholmod_outputs/. The harness itself (file formats, the comparison) is
exercised on every run against files written in the same format from the oracle -- that proves the reader, not parity.

Tolerances: north_star's 1e-8 relative for factor / solve / logdet / backward solve; selinv diagonal 1e-8 (reference's own)."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IN = os.path.join(ROOT, "tests", "golden", "cholmod_inputs")
OUT = os.path.join(ROOT, "tests", "golden", "cholmod_outputs")
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(IN, "*")) if os.path.isdir(p))
SKIP = ("CHOLMOD parity vectors absent (no Julia in this image): run `julia bench/cholmod_baseline.jl tests/golden/cholmod_inputs/{c} "
        "tests/golden/cholmod_outputs/{c}` where Julia exists and commit the output -- until then parity is UNPINNED")


def load_inputs(case):
    d = os.path.join(IN, case)
    n, nnz, nrhs = (int(x) for x in np.fromfile(os.path.join(d, "meta.bin"), np.int64))
    colptr = np.fromfile(os.path.join(d, "colptr.bin"), np.int64) - 1
    rowval = np.fromfile(os.path.join(d, "rowval.bin"), np.int64) - 1
    nz = np.fromfile(os.path.join(d, "nzval.bin"), np.float64)
    Q = sp.csc_matrix((nz, rowval, colptr), shape=(n, n))
    perm = np.fromfile(os.path.join(d, "perm.bin"), np.int64) - 1
    B = np.fromfile(os.path.join(d, "B.bin"), np.float64).reshape(nrhs, n).T.copy()
    return Q, perm, B


def load_outputs(d, n):
    sc = np.fromfile(os.path.join(d, "scalars.bin"), np.float64)
    k = int(sc[3])
    o = {"logdet": float(sc[0]), "nnz_L": int(sc[1]), "k": k,
         "p": np.fromfile(os.path.join(d, "p.bin"), np.int64) - 1,
         "X": np.fromfile(os.path.join(d, "X.bin"), np.float64).reshape(k, -1).T,
         "UPz": np.fromfile(os.path.join(d, "UPz.bin"), np.float64)}
    if os.path.exists(os.path.join(d, "Lnzval.bin")):
        cp = np.fromfile(os.path.join(d, "Lcolptr.bin"), np.int64) - 1
        rv = np.fromfile(os.path.join(d, "Lrowval.bin"), np.int64) - 1
        o["L"] = sp.csc_matrix((np.fromfile(os.path.join(d, "Lnzval.bin"), np.float64), rv, cp), shape=(n, n))
    if os.path.exists(os.path.join(d, "selinv_diag.bin")):
        o["selinv_diag"] = np.fromfile(os.path.join(d, "selinv_diag.bin"), np.float64)
    return o


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def compare(got, ref):
    """got: dict with the same keys computed by the implementation under test ON ref['p'] (the permutation CHOLMOD used)"""
    assert abs(got["logdet"] - ref["logdet"]) <= 1e-8 * abs(ref["logdet"])
    assert rel(got["X"], ref["X"]) <= 1e-8
    if "UPz" in ref:
        assert rel(got["UPz"], ref["UPz"]) <= 1e-8
    if "L" in ref:
        assert got["nnz_L"] <= ref["nnz_L"]      # CHOLMOD's supernodal L stores amalgamation zeros; the true fill is a subset
        D = (got["L"] - ref["L"]).tocsc()
        assert abs(D).max() <= 1e-8 * abs(ref["L"]).max()
    if "selinv_diag" in ref:
        assert rel(got["selinv_diag"], ref["selinv_diag"]) <= 1e-8


def oracle_results(Q, p, B, k):
    F = orc.OracleFactor(Q, p)
    return {"logdet": F.logdet(), "X": F.solve(B[:, :k]), "UPz": F.backward_solve(B[:, 0]), "nnz_L": int(F.nnz_L), "L": F.L(),
            "selinv_diag": F.selinv_diag()}


def write_in_cholmod_format(d, res, p, n, k):
    os.makedirs(d, exist_ok=True)
    np.array([res["logdet"], res["nnz_L"], n, k], np.float64).tofile(os.path.join(d, "scalars.bin"))
    (np.asarray(p, np.int64) + 1).tofile(os.path.join(d, "p.bin"))
    np.asfortranarray(res["X"]).T.copy().tofile(os.path.join(d, "X.bin"))
    np.asarray(res["UPz"], np.float64).tofile(os.path.join(d, "UPz.bin"))
    L = sp.csc_matrix(res["L"]); L.sort_indices()
    (L.indptr.astype(np.int64) + 1).tofile(os.path.join(d, "Lcolptr.bin"))
    (L.indices.astype(np.int64) + 1).tofile(os.path.join(d, "Lrowval.bin"))
    L.data.astype(np.float64).tofile(os.path.join(d, "Lnzval.bin"))
    np.asarray(res["selinv_diag"], np.float64).tofile(os.path.join(d, "selinv_diag.bin"))


def test_cases_exist_and_inputs_are_consistent():
    assert CASES, "tests/golden/cholmod_inputs is empty: run tests/golden/make_cholmod_inputs.py"
    for c in CASES:
        Q, perm, B = load_inputs(c)
        n = Q.shape[0]
        assert sorted(perm.tolist()) == list(range(n)) and B.shape[0] == n
        assert abs(Q - Q.T).max() < 1e-12 * abs(Q).max()


@pytest.mark.parametrize("case", CASES)
def test_harness_reads_the_documented_file_formats(case, tmp_path):
    """NOT parity: the oracle's own results written in the Julia script's format, read back and compared (with a dense
    cross-check of the same quantities) -- so that the day the CHOLMOD files arrive the comparison code is known to work."""
    Q, perm, B = load_inputs(case)
    n = Q.shape[0]
    res = oracle_results(Q, perm, B, 4)
    write_in_cholmod_format(str(tmp_path), res, perm, n, 4)
    ref = load_outputs(str(tmp_path), n)
    compare(oracle_results(Q, ref["p"], B, ref["k"]), ref)
    Qd = Q.toarray()
    assert abs(ref["logdet"] - np.linalg.slogdet(Qd)[1]) < 1e-10 * abs(ref["logdet"])
    assert rel(ref["X"], np.linalg.solve(Qd, B[:, :4])) < 1e-9
    assert rel(ref["selinv_diag"], np.diag(np.linalg.inv(Qd))) < 1e-8
    Lp = ref["L"].toarray()
    assert rel(Lp @ Lp.T, Qd[np.ix_(ref["p"], ref["p"])]) < 1e-12


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_cholmod(case):
    d = os.path.join(OUT, case)
    if not os.path.exists(os.path.join(d, "scalars.bin")):
        pytest.skip(SKIP.format(c=case))
    Q, perm, B = load_inputs(case)
    ref = load_outputs(d, Q.shape[0])
    compare(oracle_results(Q, ref["p"], B, ref["k"]), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_path_matches_cholmod(case):
    d = os.path.join(OUT, case)
    if not os.path.exists(os.path.join(d, "scalars.bin")):
        pytest.skip(SKIP.format(c=case))
    import gmrfx
    Q, perm, B = load_inputs(case)
    ref = load_outputs(d, Q.shape[0])
    be = gmrfx.MI355XBackend(Q, ordering=ref["p"], device=0)
    # integer work, exact: with an explicit ordering the backend only postorders the elimination tree -- same fill
    got = {"logdet": be.compute_logdet(), "X": be.backend_solve(B[:, :ref["k"]]), "UPz": be.backend_backward_solve(B[:, 0]),
           "selinv_diag": be.get_selinv_diag(), "nnz_L": int(be.stats()["nnz_l"])}
    pb = be.ordering_permutation()
    if "L" in ref and np.array_equal(pb, ref["p"]):
        got["L"] = be.factor_csc()          # entry by entry only when the pivots carry the same labels
        compare(got, ref)
    else:
        # relabelled pivots: L and P' L^-T z (z is indexed in elimination order) are not comparable entry by entry
        ref2 = {k: v for k, v in ref.items() if k not in ("L", "UPz")}
        compare(got, ref2)
        if "L" in ref:                       # same P Q P' up to a relabelling of the pivots: the diagonal of L is a permutation-
            Ld = np.sort(be.factor_csc().diagonal())          # invariant multiset only for equal orderings; logdet covers it
            assert abs(2 * np.log(Ld).sum() - ref["logdet"]) <= 1e-8 * abs(ref["logdet"])
