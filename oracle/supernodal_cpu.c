/* supernodal_cpu.c -- TEST / BENCH INFRASTRUCTURE ONLY (never linked into libgmrfx.so).
 *
 * A multi-threaded SUPERNODAL multifrontal Cholesky + multi-RHS solve on the host, built from dense BLAS-3 /
 * LAPACK calls: the same algorithm family as the reference's CPU path -- CHOLMOD's supernodal numeric
 * factorisation and solve behind `cholesky!(F, S; check=false)` and `F \ B`
 * (/root/reference/src/workspace/backend.jl:178-209) -- which cannot run here (no Julia, no libcholmod).
 * It is the `cpu_baseline` of bench.py (kind "port"): the GPU path is timed beside it on the same Q, the same
 * permutation and the same supernode partition, on the GPU box's host cores (count stated in the bench line).
 *
 * Dense kernels: the OpenBLAS that ships inside the scipy wheel (LP64 symbols scipy_cblas_dgemm / dsyrk / dtrsm,
 * scipy_LAPACKE_dpotrf), passed in as function pointers by oracle/sncpu.py -- nothing is linked at build time.
 * Parallelism: tree levels with many fronts run one front per OpenMP thread (BLAS single-threaded); levels with
 * few, large fronts run front by front with the BLAS threaded.
 *
 * Symbolic structure (supernodes, row lists, relative indices, panel layout, Q scatter map, levels): arrays as
 * exported by gmrfx_symbolic_get (include/gmrfx.h); checked against the simplicial oracle in tests/test_oracle.py.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void (*dgemm_t)(int, int, int, int, int, int, double, const double *, int, const double *, int, double, double *, int);
typedef void (*dsyrk_t)(int, int, int, int, int, double, const double *, int, double, double *, int);
typedef void (*dtrsm_t)(int, int, int, int, int, int, int, double, const double *, int, double *, int);
typedef int (*dpotrf_t)(int, char, int, double *, int);
typedef void (*setthr_t)(int);

enum { ColMajor = 102, NoTrans = 111, Trans = 112, Lower = 122, NonUnit = 131, Left = 141, Right = 142 };

typedef struct {
    int ns, nlevels, nthreads;
    int64_t n;
    const int64_t *sfirst, *rowptr, *rows, *rel, *parent, *panelptr, *ld, *level;
    int64_t *lvptr, *lvlist;      /* fronts by level */
    int64_t *chptr, *chlist;      /* children lists */
    double *L;                    /* panels (caller's buffer) */
    dgemm_t gemm; dsyrk_t syrk; dtrsm_t trsm; dpotrf_t potrf; setthr_t setthr;
    int64_t fail;
} sn_t;

sn_t *sncpu_create(int ns, int64_t n, const int64_t *sfirst, const int64_t *rowptr, const int64_t *rows, const int64_t *rel,
                   const int64_t *parent, const int64_t *panelptr, const int64_t *ld, const int64_t *level, double *L,
                   void *gemm, void *syrk, void *trsm, void *potrf, void *setthr, int nthreads) {
    sn_t *S = (sn_t *)calloc(1, sizeof(sn_t));
    S->ns = ns; S->n = n; S->sfirst = sfirst; S->rowptr = rowptr; S->rows = rows; S->rel = rel; S->parent = parent;
    S->panelptr = panelptr; S->ld = ld; S->level = level; S->L = L; S->nthreads = nthreads;
    S->gemm = (dgemm_t)gemm; S->syrk = (dsyrk_t)syrk; S->trsm = (dtrsm_t)trsm; S->potrf = (dpotrf_t)potrf; S->setthr = (setthr_t)setthr;
    int nl = 0;
    for (int s = 0; s < ns; s++) if (level[s] + 1 > nl) nl = (int)level[s] + 1;
    S->nlevels = nl;
    S->lvptr = (int64_t *)calloc(nl + 1, sizeof(int64_t));
    S->lvlist = (int64_t *)malloc(sizeof(int64_t) * (ns > 0 ? ns : 1));
    for (int s = 0; s < ns; s++) S->lvptr[level[s] + 1]++;
    for (int l = 0; l < nl; l++) S->lvptr[l + 1] += S->lvptr[l];
    int64_t *w = (int64_t *)malloc(sizeof(int64_t) * (nl + 1));
    memcpy(w, S->lvptr, sizeof(int64_t) * (nl + 1));
    for (int s = 0; s < ns; s++) S->lvlist[w[level[s]]++] = s;
    S->chptr = (int64_t *)calloc(ns + 1, sizeof(int64_t));
    S->chlist = (int64_t *)malloc(sizeof(int64_t) * (ns > 0 ? ns : 1));
    for (int s = 0; s < ns; s++) if (parent[s] >= 0) S->chptr[parent[s] + 1]++;
    for (int s = 0; s < ns; s++) S->chptr[s + 1] += S->chptr[s];
    int64_t *w2 = (int64_t *)malloc(sizeof(int64_t) * (ns + 1));
    memcpy(w2, S->chptr, sizeof(int64_t) * (ns + 1));
    for (int s = 0; s < ns; s++) if (parent[s] >= 0) S->chlist[w2[parent[s]]++] = s;   /* ascending child order */
    free(w); free(w2);
    return S;
}
void sncpu_free(sn_t *S) { if (S) { free(S->lvptr); free(S->lvlist); free(S->chptr); free(S->chlist); free(S); } }

/* one front: assemble the children's contribution blocks, dense partial factorisation, own contribution block */
static void factor_front(sn_t *S, int s, double **cb) {
    const int64_t first = S->sfirst[s];
    const int c = (int)(S->sfirst[s + 1] - first), r = (int)(S->rowptr[s + 1] - S->rowptr[s]), m = r - c, ld = (int)S->ld[s];
    double *P = S->L + S->panelptr[s];
    double *F = m > 0 ? (double *)calloc((size_t)m * m, sizeof(double)) : NULL;
    for (int64_t q = S->chptr[s]; q < S->chptr[s + 1]; q++) {
        const int d = (int)S->chlist[q];
        const int cd = (int)(S->sfirst[d + 1] - S->sfirst[d]), md = (int)(S->rowptr[d + 1] - S->rowptr[d]) - cd;
        const int64_t *reld = S->rel + S->rowptr[d] + cd;
        const double *U = cb[d];
        for (int j = 0; j < md; j++) {
            const int tj = (int)reld[j];
            if (tj < c) { double *col = P + (size_t)tj * ld; for (int i = j; i < md; i++) col[reld[i]] += U[i + (size_t)j * md]; }
            else { double *col = F + (size_t)(tj - c) * m - c; for (int i = j; i < md; i++) col[reld[i]] += U[i + (size_t)j * md]; }
        }
        free(cb[d]); cb[d] = NULL;
    }
    const int info = S->potrf(ColMajor, 'L', c, P, ld);
    if (info != 0) {
#pragma omp critical
        { if (S->fail < 0 || first + info - 1 < S->fail) S->fail = first + info - 1; }
    }
    if (m > 0) {
        S->trsm(ColMajor, Right, Lower, Trans, NonUnit, m, c, 1.0, P, ld, P + c, ld);
        S->syrk(ColMajor, Lower, NoTrans, m, c, -1.0, P + c, ld, 1.0, F, m);
    }
    cb[s] = F;
}

/* numeric factorisation: L must hold zeros + Q's values scattered by the caller (L[qdst] = nz[qsrc]) */
int64_t sncpu_factor(sn_t *S) {
    double **cb = (double **)calloc(S->ns > 0 ? S->ns : 1, sizeof(double *));
    S->fail = -1;
    for (int l = 0; l < S->nlevels; l++) {
        const int64_t a = S->lvptr[l], b = S->lvptr[l + 1];
        if (b - a >= 2 * (int64_t)S->nthreads) {
            S->setthr(1);
#pragma omp parallel for schedule(dynamic, 4) num_threads(S->nthreads)
            for (int64_t k = a; k < b; k++) factor_front(S, (int)S->lvlist[k], cb);
        } else {
            S->setthr(S->nthreads);
            for (int64_t k = a; k < b; k++) factor_front(S, (int)S->lvlist[k], cb);
        }
    }
    for (int s = 0; s < S->ns; s++) free(cb[s]);
    free(cb);
    S->setthr(S->nthreads);
    return S->fail;
}

double sncpu_logdet(const sn_t *S) {
    double acc = 0.0;
    for (int s = 0; s < S->ns; s++) {
        const int c = (int)(S->sfirst[s + 1] - S->sfirst[s]), ld = (int)S->ld[s];
        const double *P = S->L + S->panelptr[s];
        for (int j = 0; j < c; j++) acc += log(P[j + (size_t)j * ld]);
    }
    return 2.0 * acc;
}

/* X: row-major n x nrhs in ELIMINATION order (= column-major nrhs x n), solved in place. mode 0: L L' x = b, 1: L' x = b */
static void fwd_front(sn_t *S, int s, double *X, int nr, double **W) {
    const int64_t first = S->sfirst[s];
    const int c = (int)(S->sfirst[s + 1] - first), r = (int)(S->rowptr[s + 1] - S->rowptr[s]), m = r - c, ld = (int)S->ld[s];
    const double *P = S->L + S->panelptr[s];
    double *Y = X + (size_t)first * nr;                                          /* nrhs x c, column-major */
    double *Ws = m > 0 ? (double *)calloc((size_t)m * nr, sizeof(double)) : NULL;    /* nrhs x m */
    for (int64_t q = S->chptr[s]; q < S->chptr[s + 1]; q++) {
        const int d = (int)S->chlist[q];
        const int cd = (int)(S->sfirst[d + 1] - S->sfirst[d]), md = (int)(S->rowptr[d + 1] - S->rowptr[d]) - cd;
        const int64_t *reld = S->rel + S->rowptr[d] + cd;
        for (int i = 0; i < md; i++) {
            const int t = (int)reld[i];
            double *dst = t < c ? Y + (size_t)t * nr : Ws + (size_t)(t - c) * nr;
            const double *src = W[d] + (size_t)i * nr;
            for (int j = 0; j < nr; j++) dst[j] += src[j];
        }
        free(W[d]); W[d] = NULL;
    }
    S->trsm(ColMajor, Right, Lower, Trans, NonUnit, nr, c, 1.0, P, ld, Y, nr);              /* Y <- Y L11^-T */
    if (m > 0) S->gemm(ColMajor, NoTrans, Trans, nr, m, c, -1.0, Y, nr, P + c, ld, 1.0, Ws, nr);  /* W -= Y L21' */
    W[s] = Ws;
}
static void bwd_front(sn_t *S, int s, double *X, int nr) {
    const int64_t first = S->sfirst[s];
    const int c = (int)(S->sfirst[s + 1] - first), r = (int)(S->rowptr[s + 1] - S->rowptr[s]), m = r - c, ld = (int)S->ld[s];
    const double *P = S->L + S->panelptr[s];
    double *Y = X + (size_t)first * nr;
    if (m > 0) {
        double *G = (double *)malloc((size_t)m * nr * sizeof(double));
        const int64_t *rows = S->rows + S->rowptr[s] + c;
        for (int i = 0; i < m; i++) memcpy(G + (size_t)i * nr, X + (size_t)rows[i] * nr, sizeof(double) * nr);
        S->gemm(ColMajor, NoTrans, NoTrans, nr, c, m, -1.0, G, nr, P + c, ld, 1.0, Y, nr);   /* Y -= X_R' L21 */
        free(G);
    }
    S->trsm(ColMajor, Right, Lower, NoTrans, NonUnit, nr, c, 1.0, P, ld, Y, nr);            /* Y <- Y L11^-1 */
}
void sncpu_solve(sn_t *S, double *X, int nr, int mode) {
    if (mode == 0) {
        double **W = (double **)calloc(S->ns > 0 ? S->ns : 1, sizeof(double *));
        for (int l = 0; l < S->nlevels; l++) {
            const int64_t a = S->lvptr[l], b = S->lvptr[l + 1];
            if (b - a >= 2 * (int64_t)S->nthreads) {
                S->setthr(1);
#pragma omp parallel for schedule(dynamic, 4) num_threads(S->nthreads)
                for (int64_t k = a; k < b; k++) fwd_front(S, (int)S->lvlist[k], X, nr, W);
            } else {
                S->setthr(S->nthreads);
                for (int64_t k = a; k < b; k++) fwd_front(S, (int)S->lvlist[k], X, nr, W);
            }
        }
        for (int s = 0; s < S->ns; s++) free(W[s]);
        free(W);
    }
    for (int l = S->nlevels - 1; l >= 0; l--) {
        const int64_t a = S->lvptr[l], b = S->lvptr[l + 1];
        if (b - a >= 2 * (int64_t)S->nthreads) {
            S->setthr(1);
#pragma omp parallel for schedule(dynamic, 4) num_threads(S->nthreads)
            for (int64_t k = a; k < b; k++) bwd_front(S, (int)S->lvlist[k], X, nr);
        } else {
            S->setthr(S->nthreads);
            for (int64_t k = a; k < b; k++) bwd_front(S, (int)S->lvlist[k], X, nr);
        }
    }
    S->setthr(S->nthreads);
}
