/*
 * gmrf_oracle.c -- TEST INFRASTRUCTURE ONLY. CPU restatement ("oracle") of the GMRF
 * precision-matrix hot path of GaussianMarkovRandomFields.jl. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product
 * (libgmrfx.so) never links, loads or calls it.
 *
 * What it restates (semantics from the reference; arithmetic from the literature, because
 * the reference's arithmetic lives in un-vendored dependencies -- CHOLMOD via Julia's
 * SparseArrays stdlib, SelectedInversion.jl 0.2.1 -- see SURVEY.md section 8c):
 *   cholesky(Symmetric(Q); perm=p)      src/workspace/backend.jl:147-153   -> orc_factorize
 *   F \ b, F \ B                        src/workspace/backend.jl:191-209   -> orc_solve
 *   F.UP \ z  (= P' L^-T z)             src/workspace/backend.jl:281-284,
 *                                       src/solvers/backward_solve.jl:50-53 -> orc_backward_solve
 *   logdet(F) (= 2 sum log L_jj)        src/workspace/backend.jl:211-213   -> orc_logdet
 *   selinv(F; depermute=true), selinv_diag
 *                                       src/workspace/backend.jl:226-257,
 *                                       src/solvers/selinv.jl:70-125       -> orc_selinv_*
 * Algorithms: elimination tree (Liu 1990), up-looking simplicial LL' (row-subtree
 * "ereach" formulation, Davis 2006 ch.4), Takahashi recursion (Takahashi et al. 1973;
 * Rue & Held 2005 sec. 2.4; Erisman & Tinney 1975). Deliberately *simplicial* and scalar so
 * it shares no code or data layout with the supernodal/multifrontal product path.
 *
 * PINNING: the reference ships no golden vectors for this path (its tests compare against
 * dense inv/logdet/\ built on the spot: test/workspace/test_gmrf_workspace.jl:26-57).
 * This oracle is pinned by the same dense identities (tests/test_oracle.py) and by
 * committed fixtures cross-checked with scipy SuperLU (tests/golden/, make_golden.py).
 * Bit patterns of CHOLMOD's own L / permutations are PARITY-UNPINNED (no Julia here).
 *
 * All indices int64, 0-based, CSC.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

typedef struct {
    i64 n;
    i64 *perm;   /* perm[k]  = original index of the k-th pivot  (new -> old) */
    i64 *iperm;  /* iperm[i] = position of original i in the elimination order */
    i64 *parent; /* elimination tree of P Q P' */
    i64 *Lp, *Li;
    double *Lx;  /* L in CSC, permuted ordering, diagonal first in each column, rows ascending */
    /* selected inverse on pattern(L) (same Lp/Li), filled lazily */
    double *Zx;
    i64 fail_col; /* -1 if SPD, else first permuted column with non-positive pivot */
} orc_factor;

static void *xmalloc(size_t s) { void *p = malloc(s ? s : 1); if (!p) abort(); return p; }

/* Upper triangle of C = P A P' in CSC (row <= col), from the triangle of A selected by uplo
 * ('U': entries with row<=col define A -- Julia's Symmetric(Q) default, gmrf_workspace.jl:176;
 *  'L': entries with row>=col). */
static void permuted_upper(i64 n, const i64 *Ap, const i64 *Ai, const double *Ax, char uplo,
                           const i64 *iperm, i64 **Cp_, i64 **Ci_, double **Cx_) {
    i64 *cnt = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    for (i64 j = 0; j < n; j++)
        for (i64 p = Ap[j]; p < Ap[j + 1]; p++) {
            i64 i = Ai[p];
            if ((uplo == 'U' && i > j) || (uplo == 'L' && i < j)) continue;
            i64 a = iperm[i], b = iperm[j];
            i64 c = a > b ? a : b;
            cnt[c + 1]++;
        }
    i64 *Cp = (i64 *)xmalloc(((size_t)n + 1) * sizeof(i64));
    Cp[0] = 0;
    for (i64 j = 0; j < n; j++) Cp[j + 1] = Cp[j] + cnt[j + 1];
    i64 nz = Cp[n];
    i64 *Ci = (i64 *)xmalloc((size_t)nz * sizeof(i64));
    double *Cx = (double *)xmalloc((size_t)nz * sizeof(double));
    i64 *w = (i64 *)xmalloc((size_t)n * sizeof(i64));
    memcpy(w, Cp, (size_t)n * sizeof(i64));
    for (i64 j = 0; j < n; j++)
        for (i64 p = Ap[j]; p < Ap[j + 1]; p++) {
            i64 i = Ai[p];
            if ((uplo == 'U' && i > j) || (uplo == 'L' && i < j)) continue;
            i64 a = iperm[i], b = iperm[j];
            i64 r = a < b ? a : b, c = a > b ? a : b;
            i64 q = w[c]++;
            Ci[q] = r;
            Cx[q] = Ax[p];
        }
    free(w);
    free(cnt);
    *Cp_ = Cp; *Ci_ = Ci; *Cx_ = Cx;
}

/* Liu's elimination-tree algorithm with path compression (ancestor array). */
static void etree_upper(i64 n, const i64 *Cp, const i64 *Ci, i64 *parent) {
    i64 *anc = (i64 *)xmalloc((size_t)n * sizeof(i64));
    for (i64 k = 0; k < n; k++) {
        parent[k] = -1;
        anc[k] = -1;
        for (i64 p = Cp[k]; p < Cp[k + 1]; p++) {
            i64 i = Ci[p];
            while (i != -1 && i < k) {
                i64 nx = anc[i];
                anc[i] = k;
                if (nx == -1) parent[i] = k;
                i = nx;
            }
        }
    }
    free(anc);
}

/* Pattern of row k of L: nodes reachable in the etree from the entries of C(0:k-1,k), in
 * topological order in s[top..n-1]. Marks with flag value k in w. */
static i64 ereach(i64 n, const i64 *Cp, const i64 *Ci, i64 k, const i64 *parent, i64 *s, i64 *w) {
    i64 top = n;
    w[k] = k;
    for (i64 p = Cp[k]; p < Cp[k + 1]; p++) {
        i64 i = Ci[p];
        if (i >= k) continue;
        i64 len = 0;
        for (; w[i] != k; i = parent[i]) {
            s[len++] = i;
            w[i] = k;
        }
        while (len > 0) s[--top] = s[--len];
    }
    return top;
}

void orc_free(orc_factor *F) {
    if (!F) return;
    free(F->perm); free(F->iperm); free(F->parent);
    free(F->Lp); free(F->Li); free(F->Lx); free(F->Zx);
    free(F);
}

/* perm may be NULL (identity). Returns NULL only on invalid permutation. Indefinite input
 * does not abort (seam B never throws: backend.jl:184 `check=false`); fail_col records it. */
orc_factor *orc_factorize(i64 n, const i64 *Ap, const i64 *Ai, const double *Ax, int uplo,
                          const i64 *perm) {
    orc_factor *F = (orc_factor *)calloc(1, sizeof(orc_factor));
    F->n = n;
    F->fail_col = -1;
    F->perm = (i64 *)xmalloc((size_t)n * sizeof(i64));
    F->iperm = (i64 *)xmalloc((size_t)n * sizeof(i64));
    for (i64 i = 0; i < n; i++) F->iperm[i] = -1;
    for (i64 k = 0; k < n; k++) {
        i64 o = perm ? perm[k] : k;
        if (o < 0 || o >= n || F->iperm[o] != -1) { orc_free(F); return NULL; }
        F->perm[k] = o;
        F->iperm[o] = k;
    }
    i64 *Cp, *Ci; double *Cx;
    permuted_upper(n, Ap, Ai, Ax, (char)uplo, F->iperm, &Cp, &Ci, &Cx);
    F->parent = (i64 *)xmalloc((size_t)n * sizeof(i64));
    etree_upper(n, Cp, Ci, F->parent);

    i64 *s = (i64 *)xmalloc((size_t)n * sizeof(i64));
    i64 *w = (i64 *)xmalloc((size_t)n * sizeof(i64));
    /* symbolic: column counts by walking every row subtree */
    i64 *cc = (i64 *)calloc((size_t)n, sizeof(i64));
    for (i64 i = 0; i < n; i++) w[i] = -1;
    for (i64 k = 0; k < n; k++) {
        i64 top = ereach(n, Cp, Ci, k, F->parent, s, w);
        for (i64 t = top; t < n; t++) cc[s[t]]++;
        cc[k]++;
    }
    F->Lp = (i64 *)xmalloc(((size_t)n + 1) * sizeof(i64));
    F->Lp[0] = 0;
    for (i64 j = 0; j < n; j++) F->Lp[j + 1] = F->Lp[j] + cc[j];
    i64 lnz = F->Lp[n];
    F->Li = (i64 *)xmalloc((size_t)lnz * sizeof(i64));
    F->Lx = (double *)xmalloc((size_t)lnz * sizeof(double));
    /* numeric up-looking: row k of L solves L(0:k-1,0:k-1) l = C(0:k-1,k) */
    double *x = (double *)calloc((size_t)n, sizeof(double));
    i64 *fill = cc; /* reuse as next-free pointer per column */
    for (i64 j = 0; j < n; j++) fill[j] = F->Lp[j];
    for (i64 i = 0; i < n; i++) w[i] = -1;
    for (i64 k = 0; k < n; k++) {
        i64 top = ereach(n, Cp, Ci, k, F->parent, s, w);
        double d = 0.0;
        for (i64 p = Cp[k]; p < Cp[k + 1]; p++) {
            if (Ci[p] < k) x[Ci[p]] += Cx[p]; /* += : duplicate entries sum, like sparse() */
            else if (Ci[p] == k) d += Cx[p];
        }
        for (i64 t = top; t < n; t++) {
            i64 i = s[t];
            double lki = x[i] / F->Lx[F->Lp[i]];
            x[i] = 0.0;
            for (i64 p = F->Lp[i] + 1; p < fill[i]; p++) x[F->Li[p]] -= F->Lx[p] * lki;
            d -= lki * lki;
            i64 q = fill[i]++;
            F->Li[q] = k;
            F->Lx[q] = lki;
        }
        if (!(d > 0.0) && F->fail_col < 0) F->fail_col = k;
        i64 q = fill[k]++;
        F->Li[q] = k;
        F->Lx[q] = sqrt(d);
    }
    free(x); free(cc); free(s); free(w);
    free(Cp); free(Ci); free(Cx);
    return F;
}

i64 orc_n(const orc_factor *F) { return F->n; }
i64 orc_nnz_L(const orc_factor *F) { return F->Lp[F->n]; }
i64 orc_fail_col(const orc_factor *F) { return F->fail_col; }
void orc_get_L(const orc_factor *F, i64 *Lp, i64 *Li, double *Lx) {
    memcpy(Lp, F->Lp, ((size_t)F->n + 1) * sizeof(i64));
    memcpy(Li, F->Li, (size_t)F->Lp[F->n] * sizeof(i64));
    memcpy(Lx, F->Lx, (size_t)F->Lp[F->n] * sizeof(double));
}
void orc_get_parent(const orc_factor *F, i64 *parent) { memcpy(parent, F->parent, (size_t)F->n * sizeof(i64)); }

static void lsolve(const orc_factor *F, double *y) {
    for (i64 j = 0; j < F->n; j++) {
        double v = y[j] / F->Lx[F->Lp[j]];
        y[j] = v;
        for (i64 p = F->Lp[j] + 1; p < F->Lp[j + 1]; p++) y[F->Li[p]] -= F->Lx[p] * v;
    }
}
static void ltsolve(const orc_factor *F, double *y) {
    for (i64 j = F->n - 1; j >= 0; j--) {
        double v = y[j];
        for (i64 p = F->Lp[j] + 1; p < F->Lp[j + 1]; p++) v -= F->Lx[p] * y[F->Li[p]];
        y[j] = v / F->Lx[F->Lp[j]];
    }
}

/* X = Q^{-1} B, column-major n x nrhs with leading dimensions ldb/ldx. */
void orc_solve(const orc_factor *F, const double *B, i64 ldb, i64 nrhs, double *X, i64 ldx) {
    i64 n = F->n;
    double *y = (double *)xmalloc((size_t)n * sizeof(double));
    for (i64 r = 0; r < nrhs; r++) {
        for (i64 k = 0; k < n; k++) y[k] = B[F->perm[k] + r * ldb];
        lsolve(F, y);
        ltsolve(F, y);
        for (i64 k = 0; k < n; k++) X[F->perm[k] + r * ldx] = y[k];
    }
    free(y);
}

/* X = P' L^{-T} Z  (CHOLMOD's F.UP \ z): the permuted result of the back-substitution is
 * scattered back to original ordering; z itself is NOT permuted on the way in. */
void orc_backward_solve(const orc_factor *F, const double *Z, i64 ldz, i64 nrhs, double *X, i64 ldx) {
    i64 n = F->n;
    double *y = (double *)xmalloc((size_t)n * sizeof(double));
    for (i64 r = 0; r < nrhs; r++) {
        memcpy(y, Z + r * ldz, (size_t)n * sizeof(double));
        ltsolve(F, y);
        for (i64 k = 0; k < n; k++) X[F->perm[k] + r * ldx] = y[k];
    }
    free(y);
}

double orc_logdet(const orc_factor *F) {
    double s = 0.0;
    for (i64 j = 0; j < F->n; j++) s += log(F->Lx[F->Lp[j]]);
    return 2.0 * s;
}

/* (x - mu)' Symmetric(Q, uplo) (x - mu): `dot(r, d.precision * r)` of logpdf(::WorkspaceGMRF, z),
 * /root/reference/src/workspace/workspace_gmrf.jl:288-292, and sqmahal, /root/reference/src/gmrf.jl:94-97.
 * Plain restatement: y = Symmetric(Q) r column by column (only the `uplo` triangle is read, mirrored), then r'y.
 * mu may be NULL. */
double orc_sqmahal(i64 n, const i64 *Ap, const i64 *Ai, const double *Ax, int uplo, const double *x, const double *mu) {
    double *r = (double *)xmalloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    double *y = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    for (i64 j = 0; j < n; j++) r[j] = x[j] - (mu ? mu[j] : 0.0);
    for (i64 j = 0; j < n; j++)
        for (i64 p = Ap[j]; p < Ap[j + 1]; p++) {
            i64 i = Ai[p];
            if (i == j) y[i] += Ax[p] * r[j];
            else if ((uplo == 'U' && i < j) || (uplo == 'L' && i > j)) { y[i] += Ax[p] * r[j]; y[j] += Ax[p] * r[i]; }
        }
    double s = 0.0;
    for (i64 j = 0; j < n; j++) s += r[j] * y[j];
    free(r); free(y);
    return s;
}

/* Takahashi recursion on pattern(L), last column first. */
static void ensure_selinv(orc_factor *F) {
    if (F->Zx) return;
    i64 n = F->n;
    const i64 *Lp = F->Lp, *Li = F->Li;
    const double *Lx = F->Lx;
    double *Zx = (double *)xmalloc((size_t)Lp[n] * sizeof(double));
    double *lh = (double *)xmalloc((size_t)n * sizeof(double));
    double *acc = (double *)xmalloc((size_t)n * sizeof(double));
    for (i64 j = n - 1; j >= 0; j--) {
        i64 p0 = Lp[j] + 1, p1 = Lp[j + 1], m = p1 - p0;
        double dj = Lx[Lp[j]];
        for (i64 t = 0; t < m; t++) { lh[t] = Lx[p0 + t] / dj; acc[t] = 0.0; }
        /* acc[t] = sum_s Z[i_t, i_s] lh[s]; column i_s of Z holds Z[i_t,i_s] for t >= s */
        for (i64 s = 0; s < m; s++) {
            i64 c = Li[p0 + s];
            i64 q = Lp[c]; /* walk column c; its rows contain {i_t : t >= s} */
            for (i64 t = s; t < m; t++) {
                i64 r = Li[p0 + t];
                while (Li[q] != r) q++;
                double z = Zx[q];
                acc[t] += z * lh[s];
                if (t != s) acc[s] += z * lh[t];
            }
        }
        double zjj = 1.0 / (dj * dj);
        for (i64 t = 0; t < m; t++) {
            Zx[p0 + t] = -acc[t];
            zjj += lh[t] * acc[t];
        }
        Zx[Lp[j]] = zjj;
    }
    free(lh); free(acc);
    F->Zx = Zx;
}

void orc_selinv_diag(orc_factor *F, double *out) {
    ensure_selinv(F);
    for (i64 k = 0; k < F->n; k++) out[F->perm[k]] = F->Zx[F->Lp[k]];
}

i64 orc_selinv_nnz(const orc_factor *F) { return 2 * F->Lp[F->n] - F->n; }

/* De-permuted selected inverse, both triangles, CSC with sorted rows
 * (what sparse(selinv(F; depermute=true).Z) yields: backend.jl:238-246). */
void orc_selinv_csc(orc_factor *F, i64 *Zp, i64 *Zi, double *Zv) {
    ensure_selinv(F);
    i64 n = F->n;
    const i64 *Lp = F->Lp, *Li = F->Li;
    i64 *cnt = (i64 *)calloc((size_t)n + 1, sizeof(i64));
    for (i64 j = 0; j < n; j++)
        for (i64 p = Lp[j]; p < Lp[j + 1]; p++) {
            i64 a = F->perm[Li[p]], b = F->perm[j];
            cnt[b + 1]++;
            if (a != b) cnt[a + 1]++;
        }
    Zp[0] = 0;
    for (i64 j = 0; j < n; j++) Zp[j + 1] = Zp[j] + cnt[j + 1];
    i64 *w = (i64 *)xmalloc((size_t)n * sizeof(i64));
    memcpy(w, Zp, (size_t)n * sizeof(i64));
    for (i64 j = 0; j < n; j++)
        for (i64 p = Lp[j]; p < Lp[j + 1]; p++) {
            i64 a = F->perm[Li[p]], b = F->perm[j];
            i64 q = w[b]++;
            Zi[q] = a; Zv[q] = F->Zx[p];
            if (a != b) { q = w[a]++; Zi[q] = b; Zv[q] = F->Zx[p]; }
        }
    /* sort rows within each column (insertion sort; columns are short) */
    for (i64 j = 0; j < n; j++)
        for (i64 p = Zp[j] + 1; p < Zp[j + 1]; p++) {
            i64 r = Zi[p]; double v = Zv[p]; i64 q = p - 1;
            while (q >= Zp[j] && Zi[q] > r) { Zi[q + 1] = Zi[q]; Zv[q + 1] = Zv[q]; q--; }
            Zi[q + 1] = r; Zv[q + 1] = v;
        }
    free(w); free(cnt);
}
