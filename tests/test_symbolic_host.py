"""CPU tests of the host logic of libgmrfx.so: ordering, supernodes, assembly maps, schedule.
The numeric results here come from tests/mf_hostsim.py (numpy walk over the exported symbolic
structure), checked against the oracle; the GPU kernels are tested in test_gpu_parity.py."""
import numpy as np
import pytest
import scipy.sparse as sp

import gmrfx
import orc
from gmrfx import spde
from mf_hostsim import HostSim


def _cases():
    m = spde.grid_mesh_2d(24, 24, jitter=0.25)
    yield "matern24_graph", spde.matern_precision(m, 0, 0.3), {}
    yield "matern24_coords", spde.matern_precision(m, 0, 0.3), {"coords": m.points}
    yield "matern_a3", spde.matern_precision(spde.grid_mesh_2d(17, 17), 1, 0.3), {}
    yield "rand60", spde.random_spd_precision(60), {}
    yield "rand300", spde.random_spd_precision(300, 0.02), {}
    m3 = spde.grid_mesh_3d(7, 7, 7)
    yield "matern3d", spde.matern_precision(m3, 0, 0.5), {"coords": m3.points}
    yield "natural", spde.matern_precision(spde.grid_mesh_2d(12, 12), 0, 0.3), {"ordering": "natural"}
    yield "tiny", sp.csc_matrix(np.array([[4.0, 1, 0], [1, 3, 0], [0, 0, 2]])), {}
    yield "scalar", sp.csc_matrix(np.array([[2.5]])), {}


@pytest.mark.parametrize("name,Q,kw", list(_cases()), ids=[c[0] for c in _cases()])
def test_symbolic_structure_and_hostsim(name, Q, kw):
    n = Q.shape[0]
    b = gmrfx.MI355XBackend(Q, symbolic_only=True, **kw)
    perm = b.ordering_permutation()
    assert sorted(perm.tolist()) == list(range(n))            # integer work: exact
    st = b.stats()
    F = orc.OracleFactor(Q, perm)
    assert st["nnz_l"] == F.nnz_L                              # column counts exact
    sy = b.symbolic()
    # supernode partition covers 0..n, rows sorted, own columns first, rel consistent
    assert sy.super_first[0] == 0 and sy.super_first[-1] == n
    Lo = F.L()
    for s in range(len(sy.super_parent)):
        rows = sy.rows[sy.row_ptr[s]:sy.row_ptr[s + 1]]
        c = sy.super_first[s + 1] - sy.super_first[s]
        assert np.array_equal(rows[:c], np.arange(sy.super_first[s], sy.super_first[s + 1]))
        assert np.all(np.diff(rows) > 0)
        p = sy.super_parent[s]
        rel = sy.rel[sy.row_ptr[s] + c:sy.row_ptr[s + 1]]
        if len(rows) > c:
            prow = sy.rows[sy.row_ptr[p]:sy.row_ptr[p + 1]]
            assert np.array_equal(prow[rel], rows[c:])
            assert sy.level[p] > sy.level[s]
        else:
            assert p == -1
        # the supernode structure contains the true pattern of every one of its columns
        for j in range(c):
            col = sy.super_first[s] + j
            true_rows = Lo.indices[Lo.indptr[col]:Lo.indptr[col + 1]]
            assert np.all(np.isin(true_rows, rows[j:]))
    # numeric walk == oracle
    hs = HostSim(sy, n, np.asarray(Q.data, dtype=float)).factor()
    Ld = hs.dense_from_panels(hs.L)
    assert np.allclose(Ld, Lo.toarray(), rtol=1e-10, atol=1e-12)
    rng = np.random.default_rng(1)
    B = rng.standard_normal((n, 3))
    X = np.empty_like(B)
    X[perm] = hs.solve(B[perm])
    assert np.allclose(X, F.solve(B), rtol=1e-9, atol=1e-12)
    Xb = np.empty_like(B)
    Xb[perm] = hs.solve(B, mode=1)
    assert np.allclose(Xb, F.backward_solve(B), rtol=1e-9, atol=1e-12)
    assert np.isclose(hs.logdet(), F.logdet(), rtol=1e-12)
    Z = hs.dense_from_panels(hs.selinv())
    Zo = F.selinv().toarray()[np.ix_(perm, perm)]
    mask = np.tril(Zo != 0)
    assert np.allclose(Z[mask], Zo[mask], rtol=1e-8, atol=1e-12)


def test_user_permutation_is_respected_up_to_postorder():
    Q = spde.matern_precision(spde.grid_mesh_2d(10, 10), 0, 0.3)
    rng = np.random.default_rng(0)
    p = rng.permutation(100)
    b = gmrfx.MI355XBackend(Q, ordering=p, symbolic_only=True)
    # same fill as the user's order (postordering never changes nnz(L))
    assert b.stats()["nnz_l"] == orc.OracleFactor(Q, p).nnz_L


def test_invalid_inputs_raise():
    Q = spde.random_spd_precision(20)
    with pytest.raises(ValueError):
        gmrfx.MI355XBackend(Q, ordering=np.zeros(20, dtype=np.int64), symbolic_only=True)
    with pytest.raises(ValueError):
        gmrfx.MI355XBackend(sp.csc_matrix(np.ones((3, 4))), symbolic_only=True)
    with pytest.raises(ValueError):
        gmrfx.MI355XBackend(Q, ordering="bogus", symbolic_only=True)


def test_one_triangle_input_equals_full_input():
    Q = spde.matern_precision(spde.grid_mesh_2d(9, 9, jitter=0.2), 0, 0.4)
    full = gmrfx.MI355XBackend(Q, symbolic_only=True)
    up = gmrfx.MI355XBackend(sp.triu(Q, format="csc"), ordering=full.ordering_permutation(), symbolic_only=True)
    lo = gmrfx.MI355XBackend(sp.tril(Q, format="csc"), ordering=full.ordering_permutation(), symbolic_only=True)
    assert full.stats()["nnz_l"] == up.stats()["nnz_l"] == lo.stats()["nnz_l"]
    for bk, M in ((full, Q), (up, sp.triu(Q, format="csc")), (lo, sp.tril(Q, format="csc"))):
        sy = bk.symbolic()
        hs = HostSim(sy, Q.shape[0], np.asarray(M.data, float)).factor()
        assert np.isclose(hs.logdet(), np.linalg.slogdet(Q.toarray())[1], rtol=1e-12)


def test_numeric_calls_fail_loudly_without_device():
    """Symbolic-only handles (and GPU-less boxes) must refuse numeric work: no CPU fallback."""
    Q = spde.random_spd_precision(20)
    b = gmrfx.MI355XBackend(Q, symbolic_only=True)
    with pytest.raises(gmrfx.NoDeviceError):
        b.refactorize(Q)
    with pytest.raises(gmrfx.NoDeviceError):
        b.backend_solve(np.ones(20))
    with pytest.raises(gmrfx.NoDeviceError):
        b.compute_logdet()
    with pytest.raises(gmrfx.NoDeviceError):
        b.get_selinv_diag()


@pytest.mark.parametrize("with_coords", [True, False])
def test_threaded_symbolic_analysis_equals_serial(with_coords, monkeypatch):
    # near the top of the dissection tree the two halves run as separate host tasks, and the scatter map is built
    # per supernode range on several threads: ordering, supernodes and maps must not depend on that (every rank of
    # a sharded factorisation runs the analysis for itself and has to arrive at the same structure)
    g = 460 if with_coords else 230        # 460 x 460: more than 2e6 scatter entries, the threaded map build as well
    mesh = spde.grid_mesh_2d(g, g, jitter=0.25, seed=4)
    Q = sp.csc_matrix(spde.matern_precision(mesh, 0, 0.2))
    kw = {"coords": mesh.points} if with_coords else {}
    got = []
    for threads in ("1", "8"):
        monkeypatch.setenv("GMRFX_ND_THREADS", threads)
        be = gmrfx.MI355XBackend(Q, symbolic_only=True, **kw)
        sy = be.symbolic()
        got.append((be.ordering_permutation(), np.asarray(sy.super_first), np.asarray(sy.row_ptr), np.asarray(sy.rows),
                    np.asarray(sy.q_src), np.asarray(sy.q_dst)))
    for a, b in zip(*got):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", ["2d", "3d", "rand"])
def test_sweep_tasks_structure_and_local_vector_walk(shape):
    """Sweep tasks (csrc/sweep_task.hip): maximal bottom subtrees whose triangular sweeps run on a local vector in
    LDS. Host-side invariants, and the task algorithm itself restated in numpy on the exported structures -- forward:
    V = [b of the subtree; 0], per front y = L11^-1 V[own], V[lrow] -= L21 y; backward the mirror image -- against the
    level-order multifrontal walk (HostSim) front by front."""
    from mf_hostsim import HostSim
    if shape == "2d":
        m = spde.grid_mesh_2d(60, 47, jitter=0.25, seed=4)
        Q, kw = sp.csc_matrix(spde.matern_precision(m, 0, 0.3)), {"coords": m.points}
    elif shape == "3d":
        m = spde.grid_mesh_3d(9, 8, 10)
        Q, kw = sp.csc_matrix(spde.matern_precision(m, 0, 0.5)), {"coords": m.points}
    else:
        Q, kw = spde.random_spd_precision(300, 0.01), {}
    n = Q.shape[0]
    b = gmrfx.MI355XBackend(Q, symbolic_only=True, **kw)
    sy = b.symbolic()
    cap, first, last, lrow = b.sweep_tasks()
    ns = len(sy.super_parent)
    c = np.diff(sy.super_first); r = np.diff(sy.row_ptr)
    assert 0 < cap <= 288
    seen = np.zeros(ns, bool)
    assert len(first) > 0
    for f, t in zip(first, last):
        assert t - f + 1 >= 2 and t - f + 1 <= 64 and not seen[f:t + 1].any()
        seen[f:t + 1] = True
        # a complete subtree: every front's parent is inside, except the root's; no outside front has a parent inside
        assert all(f < sy.super_parent[s] <= t for s in range(f, t))
        assert not (f <= sy.super_parent[t] <= t)
        assert c[f:t + 1].max() <= 64
        col0, col1 = sy.super_first[f], sy.super_first[t + 1]
        nt = col1 - col0
        assert nt + (r[t] - c[t]) <= cap
        root_trail = sy.rows[sy.row_ptr[t] + c[t]:sy.row_ptr[t + 1]]
        for s in range(f, t + 1):
            rows = sy.rows[sy.row_ptr[s]:sy.row_ptr[s + 1]]
            lr = lrow[sy.row_ptr[s]:sy.row_ptr[s + 1]]
            assert (lr[:c[s]] == -1).all()
            glob = np.concatenate([np.arange(col0, col1), root_trail])
            assert np.array_equal(glob[lr[c[s]:]], rows[c[s]:])          # local row -> the same global row
    outside = np.flatnonzero(~seen)
    assert all(not seen[sy.super_parent[s]] for s in outside if sy.super_parent[s] >= 0)
    assert (lrow[np.repeat(~seen, r)] == -1).all()
    # numeric walk of the task algorithm on the host factor
    perm = b.ordering_permutation()
    hs = HostSim(sy, n, Q.data).factor()
    rng = np.random.default_rng(0)
    Bp = rng.standard_normal((n, 3))
    # reference: plain column-oriented forward substitution y = L^-1 b and backward x = L^-T y on the dense factor
    Ld = hs.dense_from_panels(hs.L)
    Yref = np.linalg.solve(Ld, Bp)
    Xref = np.linalg.solve(Ld.T, Yref)
    for f, t in zip(first, last):
        col0, col1 = sy.super_first[f], sy.super_first[t + 1]
        nt = col1 - col0
        mroot = r[t] - c[t]
        # forward on the task: only valid for rows whose updates all come from inside the subtree = the own rows
        V = np.zeros((nt + mroot, 3))
        V[:nt] = Bp[col0:col1]
        for s in range(f, t + 1):
            P = hs.panel(hs.L, s)
            o = sy.super_first[s] - col0
            y = np.linalg.solve(np.tril(P[:c[s]]), V[o:o + c[s]])
            V[o:o + c[s]] = y
            V[lrow[sy.row_ptr[s] + c[s]:sy.row_ptr[s + 1]]] -= P[c[s]:] @ y
        assert np.allclose(V[:nt], Yref[col0:col1], rtol=1e-10, atol=1e-12)
        # backward on the task, given the final x of the root's trailing rows
        root_trail = sy.rows[sy.row_ptr[t] + c[t]:sy.row_ptr[t + 1]]
        V = np.zeros((nt + mroot, 3))
        V[:nt] = Yref[col0:col1]
        V[nt:] = Xref[root_trail]
        for s in range(t, f - 1, -1):
            P = hs.panel(hs.L, s)
            o = sy.super_first[s] - col0
            tt = V[o:o + c[s]] - P[c[s]:].T @ V[lrow[sy.row_ptr[s] + c[s]:sy.row_ptr[s + 1]]]
            V[o:o + c[s]] = np.linalg.solve(np.tril(P[:c[s]]).T, tt)
        assert np.allclose(V[:nt], Xref[col0:col1], rtol=1e-9, atol=1e-12)


def test_schedule_shape_levels_by_depth_and_wide_supernodes_end_at_branching_columns(monkeypatch):
    """Round 3 (csrc/symbolic.cpp): (a) every front sits exactly one level below its parent (levels by depth below the root; with
    GMRFX_TOP_BY_DEPTH=0 by height above the leaves: leaves on level 0) and both schedules factor to the same log-determinant on
    the host walk; (b) a run of 128 .. 4096 columns ends below a branching column: with the rule on, the root of a 2-D mesh is
    the root separator alone and the separators of BOTH halves are fronts of their own on the level below; with the rule off
    (GMRFX_MERGE_WIDE huge) one of them is absorbed by the root. Same fill, same flops, same log-determinant."""
    mesh = spde.grid_mesh_2d(150, 150, jitter=0.25, seed=3)
    Q = spde.matern_precision(mesh, 0, 0.2)
    n = Q.shape[0]

    def analyse():
        be = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True)
        sy, st = be.symbolic(), be.stats()
        be.close()
        return sy, st

    sy, st = analyse()
    par, lv = np.asarray(sy.super_parent), np.asarray(sy.level)
    c = np.diff(sy.super_first)
    has = par >= 0
    assert (lv[has] == lv[par[has]] - 1).all() and (lv[~has] == lv.max()).all()
    root = int(np.flatnonzero(~has)[-1])
    kids = np.flatnonzero(par == root)
    assert len(kids) == 2 and (c[kids] >= 128).all() and c[root] < c[kids].sum() + c[root]      # both halves' separators are separate fronts
    ld = HostSim(sy, n, np.asarray(Q.data)).factor().logdet()

    monkeypatch.setenv("GMRFX_TOP_BY_DEPTH", "0")
    sy0, st0 = analyse()
    lv0, par0 = np.asarray(sy0.level), np.asarray(sy0.super_parent)
    leaves = np.setdiff1d(np.arange(len(lv0)), par0[par0 >= 0])
    assert (lv0[leaves] == 0).all() and (lv0[par0 >= 0] < lv0[par0[par0 >= 0]]).all() and lv0.max() == lv.max()
    ld0 = HostSim(sy0, n, np.asarray(Q.data)).factor().logdet()
    assert abs(ld - ld0) <= 1e-12 * abs(ld)
    monkeypatch.delenv("GMRFX_TOP_BY_DEPTH")

    monkeypatch.setenv("GMRFX_MERGE_WIDE", "1000000000")
    sy1, st1 = analyse()
    c1, par1 = np.diff(sy1.super_first), np.asarray(sy1.super_parent)
    root1 = int(np.flatnonzero(par1 < 0)[-1])
    assert len(c1) == len(c) - 1 and c1[root1] == c[root] + c[kids].min() or c1[root1] == c[root] + c[kids].max()
    assert st1["nnz_l"] == st["nnz_l"] and abs(st1["factor_flops"] - st["factor_flops"]) <= 1e-9 * st["factor_flops"]
    ld1 = HostSim(sy1, n, np.asarray(Q.data)).factor().logdet()
    assert abs(ld - ld1) <= 1e-12 * abs(ld)


@pytest.mark.parametrize("shape", ["2d", "2d_wide", "3d", "rand", "tall"])
def test_sweep_chunks_programs_walk(shape, monkeypatch):
    """Round 5 (csrc/sweep_chunk.hip): the tasks' CHUNK programs restated in numpy on the exported structures. Forward: the
    chunks of a task one after the other, y = (diagonal 16 x 16 block)^-1 V[own], V[targets] -= L[targets, chunk] y with the
    padded target lists (padding goes to a spare row). Backward: the four slot programs with their barrier counts -- chunks
    between the same pair of barriers must be independent (nobody reads a row another one writes) -- t = V[own] -
    L[targets, chunk]' V[targets], x = (diagonal block)^-T t. Both against dense substitution with the host factor."""
    from mf_hostsim import HostSim
    if shape == "2d":
        m = spde.grid_mesh_2d(60, 47, jitter=0.25, seed=4)
        Q, kw = sp.csc_matrix(spde.matern_precision(m, 0, 0.3)), {"coords": m.points}
    elif shape == "2d_wide":          # alpha = 3 (37-point stencil) and a coarse amalgamation: task fronts of 2-4 chunks
        m = spde.grid_mesh_2d(40, 33, jitter=0.25, seed=5)
        Q, kw = sp.csc_matrix(spde.matern_precision(m, 1, 0.3)), {"coords": m.points, "relax_cols": 64, "relax_zeros": 0.4}
    elif shape == "3d":
        m = spde.grid_mesh_3d(9, 8, 10)
        Q, kw = sp.csc_matrix(spde.matern_precision(m, 0, 0.5)), {"coords": m.points}
    elif shape == "tall":             # fronts of 8 columns and 158 rows: chunks with more than 128 target rows (several forward records)
        Q, kw = spde.tall_front_precision(), {"ordering": "natural", "relax_cols": 1, "relax_zeros": 1e-9}
    else:
        Q, kw = spde.random_spd_precision(300, 0.01), {}
    n = Q.shape[0]
    b = gmrfx.MI355XBackend(Q, symbolic_only=True, **kw)
    sy = b.symbolic()
    cap, first, last, lrow = b.sweep_tasks()
    ck = b.sweep_chunks()
    if shape == "tall":
        assert len(ck["fwd"]) > len(ck["bwd"]) > 0
    c = np.diff(sy.super_first); r = np.diff(sy.row_ptr)
    hs = HostSim(sy, n, Q.data).factor()
    Ld = hs.dense_from_panels(hs.L)
    rng = np.random.default_rng(0)
    Bp = rng.standard_normal((n, 3))
    Yref = np.linalg.solve(Ld, Bp)
    Xref = np.linalg.solve(Ld.T, Yref)
    Lflat = hs.L
    SPARE = 288
    assert len(ck["task_ptr"]) == len(first) + 1 and ck["task_ptr"][-1, 0] == len(ck["fwd"]) and ck["task_ptr"][-1, 1] == len(ck["bwd"])
    assert len(ck["rows"]) % 32 == 0
    wide = 0
    for t, (f, l) in enumerate(zip(first, last)):
        col0, col1 = sy.super_first[f], sy.super_first[l + 1]
        nt = col1 - col0
        mroot = r[l] - c[l]
        (c0, b0), (c1, b1) = ck["task_ptr"][t], ck["task_ptr"][t + 1]
        assert b1 - b0 == sum((c[s] + 15) // 16 for s in range(f, l + 1)) and b1 - b0 <= c1 - c0 <= 96
        wide += int((c[f:l + 1] > 16).sum())

        def operands(rec, part=False):
            pa, ld, o, cc, ntg, lr, nbar, cid = (int(v) for v in rec)
            assert 1 <= cc <= 16 and 0 <= o and o + cc <= nt and 0 <= ntg
            npad = (ntg + 31) // 32 * 32
            rows = ck["rows"][lr:lr + npad].copy()
            assert (rows[:ntg] >= 0).all() and (rows[:ntg] < nt + mroot).all() and (rows[ntg:] == -1).all()
            D = None if part else np.array([[Lflat[pa - cc + i + j * ld] for j in range(cc)] for i in range(cc)])   # cc rows above the first target row
            A = np.array([[Lflat[pa + i + j * ld] for j in range(cc)] for i in range(ntg)]).reshape(ntg, cc)
            return o, cc, ntg, rows, (None if part else np.tril(D)), A, cid

        # ---- forward ----
        V = np.zeros((SPARE + 1, 3))
        V[:nt] = Bp[col0:col1]
        ids, prev = [], None
        for rec in ck["fwd"][c0:c1]:
            # a chunk with more than 128 target rows comes as several records: the same diagonal block (found through the
            # first record: the others start 128 rows further down), the next <= 128 target rows
            part = prev is not None and int(rec[7]) == prev[0]
            o, cc, ntg, rows, D, A, cid = operands(rec, part)
            assert ntg <= 128
            if part:
                assert (o, cc) == prev[1:3] and int(rec[0]) == prev[3] + 128
                y = prev[4]                                     # (the kernel recomputes it from rows no record of the chunk touches)
            else:
                ids.append(cid)
                y = np.linalg.solve(D, V[o:o + cc])
                V[o:o + cc] = y
            prev = (cid, o, cc, int(rec[0]), y)
            upd = np.full((len(rows), 3), 7.7e7)            # what the padding rows of a tile would produce: goes to the spare row
            upd[:ntg] = A @ y
            tgt = np.where(rows < 0, SPARE, rows)
            assert len(set(tgt[:ntg])) == ntg               # distinct rows inside a chunk
            np.subtract.at(V, tgt, upd)
        assert ids == sorted(ids) and len(set(ids)) == b1 - b0
        assert np.allclose(V[:nt], Yref[col0:col1], rtol=1e-10, atol=1e-12)
        # ---- backward: the slot programs ----
        slot = ck["slot"][t]
        assert slot[:4].sum() == b1 - b0
        progs, total = [], set()
        p0 = b0
        for q in range(4):
            phase, items = 0, []
            for rec in ck["bwd"][p0:p0 + slot[q]]:
                phase += int(rec[6])
                items.append((phase, rec))
            total.add(phase + int(slot[4 + q]))
            progs.append(items)
            p0 += slot[q]
        assert len(total) == 1 and min(slot[4:]) >= 1            # every slot passes the same number of barriers, one before the write-out
        ngroups = total.pop()
        root_trail = sy.rows[sy.row_ptr[l] + c[l]:sy.row_ptr[l + 1]]
        V = np.zeros((SPARE + 1, 3))
        V[:nt] = Yref[col0:col1]
        V[nt:nt + mroot] = Xref[root_trail]
        seen = []
        for g in range(ngroups):
            group = [rec for items in progs for (ph, rec) in items if ph == g]
            assert 1 <= len(group)
            writes, reads, results = [], [], []
            for rec in group:
                o, cc, ntg, rows, D, A, cid = operands(rec)
                seen.append(cid)
                tgt = np.where(rows < 0, SPARE, rows)
                Apad = np.vstack([A, np.full((len(rows) - ntg, cc), 3.3e3)])       # garbage operand rows meet the zero spare row
                tt = V[o:o + cc] - Apad.T @ V[tgt]
                results.append((o, cc, np.linalg.solve(D.T, tt)))
                writes.append(set(range(o, o + cc)))
                reads.append(set(tgt[:ntg]) | set(range(o, o + cc)))
            for a in range(len(group)):
                for bb in range(len(group)):
                    if a != bb:
                        assert not (writes[a] & reads[bb])
            for o, cc, x in results:
                V[o:o + cc] = x
        assert sorted(seen) == ids
        assert np.allclose(V[:nt], Xref[col0:col1], rtol=1e-9, atol=1e-12)
        assert (V[SPARE] == 0).all()
    if shape == "2d_wide":
        assert wide > 0
