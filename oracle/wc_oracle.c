/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/qp_spec.py for the rules).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED: the reference pins no solver version and holds no fixtures; the
 * two third-party solvers it calls (OSQP via osqp-eigen, qpOASES) are absent here.
 *
 * What this file restates, in plain C99 (no BLAS, no Eigen):
 *
 *  (1) osqp_*   — the published OSQP algorithm (Stellato, Banjac, Goulart, Bemporad,
 *      Boyd, "OSQP: an operator splitting solver for quadratic programs", Math. Prog.
 *      Comp. 2020) with the LIBRARY-DEFAULT settings the reference runs it at (the
 *      reference only calls setVerbosity(false): MPCSolver.cpp:51; IK adds
 *      setLinearSystemSolver(0): WalkingQPInverseKinematics_osqp.cpp:129-130):
 *      rho 0.1, sigma 1e-6, alpha 1.6, eps_abs = eps_rel = 1e-3, max_iter 4000,
 *      Ruiz scaling 10 passes, rho_eq = 1e3 rho, termination check every 25
 *      iterations, adaptive rho, no polish, warm start.  Upstream adapts rho on a
 *      WALL-CLOCK schedule; to stay reproducible this restatement uses upstream's
 *      documented fixed fallback of every 100 iterations.
 *      The KKT system is solved with an up-looking sparse LDL' after a minimum-degree
 *      ordering, like QDLDL+AMD upstream.
 *
 *  (2) as_*     — a dense dual active-set method (Goldfarb & Idnani 1983) standing in
 *      for qpOASES::SQProblem::init on the IK problem (H, g, A, lbA = ubA, lb, ub;
 *      WalkingQPInverseKinematics_qpOASES.cpp:284-339).  qpOASES is an online
 *      active-set solver; both reach the same unique optimum.
 *
 *  (3) orc_mpc_batch_osqp / orc_ik_batch — the reference-side ASSEMBLY of the two QPs
 *      (same formulas as oracle/qp_spec.py, citing the same reference lines) followed
 *      by (1)/(2): exactly the contents of the reference's "MPC" and "IK" profiler
 *      brackets (WalkingModule.cpp:604-636, 684-770) minus forward kinematics.  These
 *      are the functions bench.py times as `cpu_baseline` (kind "port").
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OSQP_INFTY_ 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_TOL 1e-4
#define RHO_EQ_FACTOR 1e3
#define ADAPTIVE_RHO_FIXED 100

typedef struct {
    double rho, sigma, alpha, eps_abs, eps_rel, adaptive_rho_tolerance;
    int max_iter, scaling, check_termination, adaptive_rho, adaptive_rho_interval;
} osqp_settings;

#include <time.h>
static double now_seconds(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static void osqp_default_settings(osqp_settings* s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
    s->adaptive_rho_tolerance = 5.0; s->max_iter = 4000; s->scaling = 10; s->check_termination = 25;
    s->adaptive_rho = 1; s->adaptive_rho_interval = ADAPTIVE_RHO_FIXED;
}

/* column-compressed sparse matrix */
typedef struct { int m, n, nnz; int* p; int* i; double* x; } csc;

static csc* csc_alloc(int m, int n, int nnz) {
    csc* A = (csc*)calloc(1, sizeof(csc));
    A->m = m; A->n = n; A->nnz = nnz;
    A->p = (int*)calloc((size_t)n + 1, sizeof(int));
    A->i = (int*)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(int));
    A->x = (double*)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(double));
    return A;
}
static void csc_free(csc* A) { if (A) { free(A->p); free(A->i); free(A->x); free(A); } }

/* dense row-major -> csc, dropping exact zeros (what Eigen's sparseView() does, osqp.cpp:153);
 * upper != 0 keeps only i <= j (OSQP takes the upper triangle of P) */
static csc* csc_from_dense(const double* D, int m, int n, int upper) {
    int nnz = 0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i)
            if (D[(size_t)i * n + j] != 0.0 && (!upper || i <= j)) ++nnz;
    csc* A = csc_alloc(m, n, nnz);
    int k = 0;
    for (int j = 0; j < n; ++j) {
        A->p[j] = k;
        for (int i = 0; i < m; ++i)
            if (D[(size_t)i * n + j] != 0.0 && (!upper || i <= j)) { A->i[k] = i; A->x[k] = D[(size_t)i * n + j]; ++k; }
    }
    A->p[n] = k;
    return A;
}

/* y = A x / y = A' x / y = P x with P symmetric, upper triangle stored */
static void csc_mv(const csc* A, const double* x, double* y) {
    memset(y, 0, sizeof(double) * (size_t)A->m);
    for (int j = 0; j < A->n; ++j) { const double xj = x[j]; for (int k = A->p[j]; k < A->p[j + 1]; ++k) y[A->i[k]] += A->x[k] * xj; }
}
static void csc_mtv(const csc* A, const double* x, double* y) {
    for (int j = 0; j < A->n; ++j) { double s = 0; for (int k = A->p[j]; k < A->p[j + 1]; ++k) s += A->x[k] * x[A->i[k]]; y[j] = s; }
}
static void csc_symv(const csc* P, const double* x, double* y) {
    memset(y, 0, sizeof(double) * (size_t)P->n);
    for (int j = 0; j < P->n; ++j)
        for (int k = P->p[j]; k < P->p[j + 1]; ++k) {
            const int i = P->i[k];
            y[i] += P->x[k] * x[j];
            if (i != j) y[j] += P->x[k] * x[i];
        }
}
static double vnorm_inf(const double* v, int n) { double m = 0; for (int i = 0; i < n; ++i) { double a = fabs(v[i]); if (a > m) m = a; } return m; }

/* ---------------- sparse LDL' (up-looking, elimination tree) ---------------------- */
typedef struct {
    int n; int *Lp, *Li, *parent, *Lnz, *flag, *pattern, *perm, *iperm;
    double *Lx, *D, *Y, *work;
    /* permuted upper-triangular KKT and the map from original entries */
    csc* K; int* map; int nmap;
} ldl_t;

/* greedy minimum-degree ordering on the pattern of a symmetric matrix (upper csc) */
static void min_degree_order(const csc* K, int* perm) {
    const int n = K->n;
    const int W = (n + 63) / 64;
    uint64_t* adj = (uint64_t*)calloc((size_t)n * W, sizeof(uint64_t));
    char* done = (char*)calloc((size_t)n, 1);
    for (int j = 0; j < n; ++j)
        for (int k = K->p[j]; k < K->p[j + 1]; ++k) {
            const int i = K->i[k];
            if (i == j) continue;
            adj[(size_t)i * W + j / 64] |= 1ull << (j % 64);
            adj[(size_t)j * W + i / 64] |= 1ull << (i % 64);
        }
    for (int s = 0; s < n; ++s) {
        int best = -1, bestdeg = n + 1;
        for (int v = 0; v < n; ++v) {
            if (done[v]) continue;
            int d = 0;
            for (int w = 0; w < W; ++w) d += __builtin_popcountll(adj[(size_t)v * W + w]);
            if (d < bestdeg) { bestdeg = d; best = v; }
        }
        perm[s] = best;
        done[best] = 1;
        /* eliminate: neighbours become a clique, best leaves the graph */
        uint64_t* nb = adj + (size_t)best * W;
        for (int v = 0; v < n; ++v) {
            if (done[v] || !((nb[v / 64] >> (v % 64)) & 1ull)) continue;
            uint64_t* av = adj + (size_t)v * W;
            for (int w = 0; w < W; ++w) av[w] |= nb[w];
            av[v / 64] &= ~(1ull << (v % 64));
            av[best / 64] &= ~(1ull << (best % 64));
        }
        memset(nb, 0, sizeof(uint64_t) * (size_t)W);
    }
    free(adj); free(done);
}

/* builds the permuted upper-triangular copy of K (perm[new] = old) and the entry map */
static void ldl_symbolic(ldl_t* F, const csc* K0) {
    const int n = K0->n;
    F->n = n;
    F->perm = (int*)malloc(sizeof(int) * n); F->iperm = (int*)malloc(sizeof(int) * n);
    min_degree_order(K0, F->perm);
    for (int k = 0; k < n; ++k) F->iperm[F->perm[k]] = k;
    /* permuted upper pattern */
    int* cnt = (int*)calloc((size_t)n + 1, sizeof(int));
    for (int j = 0; j < n; ++j)
        for (int k = K0->p[j]; k < K0->p[j + 1]; ++k) {
            int a = F->iperm[K0->i[k]], b = F->iperm[j];
            int c = a > b ? a : b;
            cnt[c + 1]++;
        }
    F->K = csc_alloc(n, n, K0->p[n]);
    for (int j = 0; j < n; ++j) F->K->p[j + 1] = F->K->p[j] + cnt[j + 1];
    int* next = (int*)malloc(sizeof(int) * n);
    memcpy(next, F->K->p, sizeof(int) * n);
    F->nmap = K0->p[n];
    F->map = (int*)malloc(sizeof(int) * (size_t)(F->nmap > 0 ? F->nmap : 1));
    for (int j = 0; j < n; ++j)
        for (int k = K0->p[j]; k < K0->p[j + 1]; ++k) {
            int a = F->iperm[K0->i[k]], b = F->iperm[j];
            int r = a < b ? a : b, c = a > b ? a : b;
            int dst = next[c]++;
            F->K->i[dst] = r;
            F->map[k] = dst;
        }
    free(cnt); free(next);
    /* elimination tree and column counts of L */
    F->parent = (int*)malloc(sizeof(int) * n); F->Lnz = (int*)calloc((size_t)n, sizeof(int));
    F->flag = (int*)malloc(sizeof(int) * n); F->pattern = (int*)malloc(sizeof(int) * n);
    F->Lp = (int*)calloc((size_t)n + 1, sizeof(int));
    for (int k = 0; k < n; ++k) {
        F->parent[k] = -1; F->flag[k] = k;
        for (int p = F->K->p[k]; p < F->K->p[k + 1]; ++p) {
            int i = F->K->i[p];
            for (; i < k && F->flag[i] != k; i = F->parent[i]) {
                if (F->parent[i] == -1) F->parent[i] = k;
                F->Lnz[i]++;
                F->flag[i] = k;
            }
        }
    }
    for (int k = 0; k < n; ++k) F->Lp[k + 1] = F->Lp[k] + F->Lnz[k];
    F->Li = (int*)malloc(sizeof(int) * (size_t)(F->Lp[n] > 0 ? F->Lp[n] : 1));
    F->Lx = (double*)malloc(sizeof(double) * (size_t)(F->Lp[n] > 0 ? F->Lp[n] : 1));
    F->D = (double*)malloc(sizeof(double) * n); F->Y = (double*)calloc((size_t)n, sizeof(double));
    F->work = (double*)malloc(sizeof(double) * n);
}

static void ldl_load(ldl_t* F, const csc* K0) { for (int k = 0; k < F->nmap; ++k) F->K->x[F->map[k]] = K0->x[k]; }

static int ldl_numeric(ldl_t* F) {
    const int n = F->n;
    const csc* K = F->K;
    for (int k = 0; k < n; ++k) {
        int top = n;
        F->Y[k] = 0.0; F->flag[k] = k; F->Lnz[k] = 0;
        for (int p = K->p[k]; p < K->p[k + 1]; ++p) {
            int i = K->i[p];
            F->Y[i] += K->x[p];
            int len = 0;
            for (; i < k && F->flag[i] != k; i = F->parent[i]) { F->pattern[len++] = i; F->flag[i] = k; }
            while (len > 0) F->pattern[--top] = F->pattern[--len];
        }
        F->D[k] = F->Y[k]; F->Y[k] = 0.0;
        for (; top < n; ++top) {
            const int i = F->pattern[top];
            const double yi = F->Y[i];
            F->Y[i] = 0.0;
            const int p2 = F->Lp[i] + F->Lnz[i];
            for (int p = F->Lp[i]; p < p2; ++p) F->Y[F->Li[p]] -= F->Lx[p] * yi;
            const double lki = yi / F->D[i];
            F->D[k] -= lki * yi;
            F->Li[p2] = k; F->Lx[p2] = lki; F->Lnz[i]++;
        }
        if (F->D[k] == 0.0) return -1;
    }
    return 0;
}

static void ldl_solve(const ldl_t* F, double* b) {
    const int n = F->n;
    double* x = F->work;
    for (int k = 0; k < n; ++k) x[k] = b[F->perm[k]];
    for (int j = 0; j < n; ++j) { const double xj = x[j]; for (int p = F->Lp[j]; p < F->Lp[j] + F->Lnz[j]; ++p) x[F->Li[p]] -= F->Lx[p] * xj; }
    for (int j = 0; j < n; ++j) x[j] /= F->D[j];
    for (int j = n - 1; j >= 0; --j) { double s = x[j]; for (int p = F->Lp[j]; p < F->Lp[j] + F->Lnz[j]; ++p) s -= F->Lx[p] * x[F->Li[p]]; x[j] = s; }
    for (int k = 0; k < n; ++k) b[F->perm[k]] = x[k];
}

static void ldl_free(ldl_t* F) {
    free(F->Lp); free(F->Li); free(F->parent); free(F->Lnz); free(F->flag); free(F->pattern);
    free(F->perm); free(F->iperm); free(F->Lx); free(F->D); free(F->Y); free(F->work); free(F->map);
    csc_free(F->K);
}

/* ---------------- OSQP workspace -------------------------------------------------- */
typedef struct {
    int n, m;
    osqp_settings st;
    csc *P, *A;               /* scaled copies (P upper triangular) */
    double *q, *l, *u;        /* scaled */
    double *D, *E, c;         /* scaling */
    double *rho_vec; int* ctype;
    double *x, *z, *y, *x_prev, *z_prev, *xt, *zt, *rhs, *tn, *tn2, *tm;
    csc* K; int* Kdiag_rho;   /* KKT (upper) and positions of the -1/rho diagonal */
    int* Krow_of_A;           /* unused placeholder */
    ldl_t F;
    double rho;
    int iters, status, rho_updates;
} osqp_work;

static double limit_scaling(double v) { v = v < MIN_SCALING ? 1.0 : v; return v > MAX_SCALING ? MAX_SCALING : v; }

static void osqp_scale(osqp_work* w) {
    const int n = w->n, m = w->m;
    double* Dt = (double*)malloc(sizeof(double) * n);
    double* Et = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    for (int i = 0; i < n; ++i) w->D[i] = 1.0;
    for (int i = 0; i < m; ++i) w->E[i] = 1.0;
    w->c = 1.0;
    for (int it = 0; it < w->st.scaling; ++it) {
        /* inf-norms of the columns of [P A'; A 0] */
        for (int j = 0; j < n; ++j) Dt[j] = 0.0;
        for (int i = 0; i < m; ++i) Et[i] = 0.0;
        for (int j = 0; j < n; ++j)
            for (int k = w->P->p[j]; k < w->P->p[j + 1]; ++k) {
                const double a = fabs(w->P->x[k]); const int i = w->P->i[k];
                if (a > Dt[j]) Dt[j] = a;
                if (a > Dt[i]) Dt[i] = a;
            }
        for (int j = 0; j < n; ++j)
            for (int k = w->A->p[j]; k < w->A->p[j + 1]; ++k) {
                const double a = fabs(w->A->x[k]); const int i = w->A->i[k];
                if (a > Dt[j]) Dt[j] = a;
                if (a > Et[i]) Et[i] = a;
            }
        for (int j = 0; j < n; ++j) Dt[j] = 1.0 / sqrt(limit_scaling(Dt[j]));
        for (int i = 0; i < m; ++i) Et[i] = 1.0 / sqrt(limit_scaling(Et[i]));
        for (int j = 0; j < n; ++j)
            for (int k = w->P->p[j]; k < w->P->p[j + 1]; ++k) w->P->x[k] *= Dt[j] * Dt[w->P->i[k]];
        for (int j = 0; j < n; ++j)
            for (int k = w->A->p[j]; k < w->A->p[j + 1]; ++k) w->A->x[k] *= Dt[j] * Et[w->A->i[k]];
        for (int j = 0; j < n; ++j) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
        for (int i = 0; i < m; ++i) w->E[i] *= Et[i];
        /* cost scaling */
        for (int j = 0; j < n; ++j) Dt[j] = 0.0;
        for (int j = 0; j < n; ++j)
            for (int k = w->P->p[j]; k < w->P->p[j + 1]; ++k) {
                const double a = fabs(w->P->x[k]); const int i = w->P->i[k];
                if (a > Dt[j]) Dt[j] = a;
                if (a > Dt[i]) Dt[i] = a;
            }
        double ct = 0.0;
        for (int j = 0; j < n; ++j) ct += Dt[j];
        ct /= n;
        double nq = limit_scaling(vnorm_inf(w->q, n));
        ct = limit_scaling(ct > nq ? ct : nq);
        ct = 1.0 / ct;
        for (int k = 0; k < w->P->p[n]; ++k) w->P->x[k] *= ct;
        for (int j = 0; j < n; ++j) w->q[j] *= ct;
        w->c *= ct;
    }
    for (int i = 0; i < m; ++i) { w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
    free(Dt); free(Et);
}

static void osqp_set_rho_vec(osqp_work* w) {
    for (int i = 0; i < w->m; ++i) {
        if (w->l[i] < -OSQP_INFTY_ * MIN_SCALING && w->u[i] > OSQP_INFTY_ * MIN_SCALING) { w->ctype[i] = -1; w->rho_vec[i] = RHO_MIN; }
        else if (w->u[i] - w->l[i] < RHO_TOL) { w->ctype[i] = 1; w->rho_vec[i] = RHO_EQ_FACTOR * w->rho; }
        else { w->ctype[i] = 0; w->rho_vec[i] = w->rho; }
    }
}

/* upper-triangular KKT = [P + sigma I, A'; . , -diag(1/rho)] */
static void osqp_build_kkt(osqp_work* w) {
    const int n = w->n, m = w->m;
    /* A in row form: count per row */
    int* rcount = (int*)calloc((size_t)m + 1, sizeof(int));
    for (int k = 0; k < w->A->p[n]; ++k) rcount[w->A->i[k] + 1]++;
    int nnz = 0;
    /* P columns with guaranteed diagonal */
    int* hasdiag = (int*)calloc((size_t)n, sizeof(int));
    for (int j = 0; j < n; ++j)
        for (int k = w->P->p[j]; k < w->P->p[j + 1]; ++k) if (w->P->i[k] == j) hasdiag[j] = 1;
    nnz = w->P->p[n];
    for (int j = 0; j < n; ++j) if (!hasdiag[j]) ++nnz;
    nnz += w->A->p[n] + m;
    w->K = csc_alloc(n + m, n + m, nnz);
    int k2 = 0;
    for (int j = 0; j < n; ++j) {
        w->K->p[j] = k2;
        int dpos = -1;
        for (int k = w->P->p[j]; k < w->P->p[j + 1]; ++k) {
            w->K->i[k2] = w->P->i[k]; w->K->x[k2] = w->P->x[k];
            if (w->P->i[k] == j) dpos = k2;
            ++k2;
        }
        if (dpos < 0) { w->K->i[k2] = j; w->K->x[k2] = 0.0; dpos = k2; ++k2; }
        w->K->x[dpos] += w->st.sigma;
    }
    /* columns n..n+m-1: row i of A, then the diagonal */
    int* rptr = (int*)malloc(sizeof(int) * ((size_t)m + 1));
    rptr[0] = 0;
    for (int i = 0; i < m; ++i) rptr[i + 1] = rptr[i] + rcount[i + 1];
    int* rj = (int*)malloc(sizeof(int) * (size_t)(w->A->p[n] > 0 ? w->A->p[n] : 1));
    double* rx = (double*)malloc(sizeof(double) * (size_t)(w->A->p[n] > 0 ? w->A->p[n] : 1));
    int* fill = (int*)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
    for (int j = 0; j < n; ++j)
        for (int k = w->A->p[j]; k < w->A->p[j + 1]; ++k) {
            const int i = w->A->i[k];
            rj[rptr[i] + fill[i]] = j; rx[rptr[i] + fill[i]] = w->A->x[k]; fill[i]++;
        }
    for (int i = 0; i < m; ++i) {
        w->K->p[n + i] = k2;
        for (int k = rptr[i]; k < rptr[i + 1]; ++k) { w->K->i[k2] = rj[k]; w->K->x[k2] = rx[k]; ++k2; }
        w->K->i[k2] = n + i; w->K->x[k2] = -1.0 / w->rho_vec[i]; w->Kdiag_rho[i] = k2; ++k2;
    }
    w->K->p[n + m] = k2;
    free(rcount); free(hasdiag); free(rptr); free(rj); free(rx); free(fill);
}

static osqp_work* osqp_setup(int n, int m, const csc* P, const double* q, const csc* A,
                             const double* l, const double* u, const osqp_settings* st) {
    osqp_work* w = (osqp_work*)calloc(1, sizeof(osqp_work));
    w->n = n; w->m = m; w->st = *st;
    w->P = csc_alloc(n, n, P->p[n]); memcpy(w->P->p, P->p, sizeof(int) * ((size_t)n + 1));
    memcpy(w->P->i, P->i, sizeof(int) * (size_t)P->p[n]); memcpy(w->P->x, P->x, sizeof(double) * (size_t)P->p[n]);
    w->A = csc_alloc(m, n, A->p[n]); memcpy(w->A->p, A->p, sizeof(int) * ((size_t)n + 1));
    memcpy(w->A->i, A->i, sizeof(int) * (size_t)A->p[n]); memcpy(w->A->x, A->x, sizeof(double) * (size_t)A->p[n]);
    const int mm = m > 0 ? m : 1;
    w->q = (double*)malloc(sizeof(double) * n); memcpy(w->q, q, sizeof(double) * n);
    w->l = (double*)malloc(sizeof(double) * mm); w->u = (double*)malloc(sizeof(double) * mm);
    for (int i = 0; i < m; ++i) {                      /* osqp clamps infinite bounds */
        w->l[i] = l[i] < -OSQP_INFTY_ ? -OSQP_INFTY_ : l[i];
        w->u[i] = u[i] > OSQP_INFTY_ ? OSQP_INFTY_ : u[i];
    }
    w->D = (double*)malloc(sizeof(double) * n); w->E = (double*)malloc(sizeof(double) * mm);
    w->rho_vec = (double*)malloc(sizeof(double) * mm); w->ctype = (int*)malloc(sizeof(int) * mm);
    w->x = (double*)calloc((size_t)n, sizeof(double)); w->z = (double*)calloc((size_t)mm, sizeof(double));
    w->y = (double*)calloc((size_t)mm, sizeof(double));
    w->x_prev = (double*)malloc(sizeof(double) * n); w->z_prev = (double*)malloc(sizeof(double) * mm);
    w->xt = (double*)malloc(sizeof(double) * n); w->zt = (double*)malloc(sizeof(double) * mm);
    w->rhs = (double*)malloc(sizeof(double) * ((size_t)n + m));
    w->tn = (double*)malloc(sizeof(double) * n); w->tn2 = (double*)malloc(sizeof(double) * n);
    w->tm = (double*)malloc(sizeof(double) * mm);
    w->Kdiag_rho = (int*)malloc(sizeof(int) * mm);
    if (st->scaling > 0) osqp_scale(w);
    else { for (int i = 0; i < n; ++i) w->D[i] = 1.0; for (int i = 0; i < m; ++i) w->E[i] = 1.0; w->c = 1.0; }
    w->rho = st->rho;
    osqp_set_rho_vec(w);
    osqp_build_kkt(w);
    ldl_symbolic(&w->F, w->K);
    ldl_load(&w->F, w->K);
    w->status = ldl_numeric(&w->F) == 0 ? 0 : -1;
    return w;
}

static void osqp_cleanup(osqp_work* w) {
    if (!w) return;
    ldl_free(&w->F);
    csc_free(w->P); csc_free(w->A); csc_free(w->K);
    free(w->q); free(w->l); free(w->u); free(w->D); free(w->E); free(w->rho_vec); free(w->ctype);
    free(w->x); free(w->z); free(w->y); free(w->x_prev); free(w->z_prev); free(w->xt); free(w->zt);
    free(w->rhs); free(w->tn); free(w->tn2); free(w->tm); free(w->Kdiag_rho);
    free(w);
}

static void osqp_update_rho(osqp_work* w, double rho_new) {
    w->rho = rho_new < RHO_MIN ? RHO_MIN : (rho_new > RHO_MAX ? RHO_MAX : rho_new);
    for (int i = 0; i < w->m; ++i) {
        if (w->ctype[i] == 0) w->rho_vec[i] = w->rho;
        else if (w->ctype[i] == 1) w->rho_vec[i] = RHO_EQ_FACTOR * w->rho;
        w->K->x[w->Kdiag_rho[i]] = -1.0 / w->rho_vec[i];
    }
    ldl_load(&w->F, w->K);
    ldl_numeric(&w->F);
    w->rho_updates++;
}

/* residuals in the SCALED space; normalisers returned for the rho estimate */
static void osqp_residuals(osqp_work* w, int unscaled, double* pr, double* dr, double* pn, double* dn) {
    const int n = w->n, m = w->m;
    /* primal: A x - z */
    csc_mv(w->A, w->x, w->tm);
    double r = 0, nax = 0, nz = 0;
    for (int i = 0; i < m; ++i) {
        const double s = unscaled ? 1.0 / w->E[i] : 1.0;
        const double a = fabs(s * (w->tm[i] - w->z[i])); if (a > r) r = a;
        const double b = fabs(s * w->tm[i]); if (b > nax) nax = b;
        const double c = fabs(s * w->z[i]); if (c > nz) nz = c;
    }
    *pr = r; *pn = nax > nz ? nax : nz;
    /* dual: P x + q + A' y */
    csc_symv(w->P, w->x, w->tn);
    csc_mtv(w->A, w->y, w->tn2);
    double d = 0, npx = 0, naty = 0, nq = 0;
    const double cs = unscaled ? 1.0 / w->c : 1.0;
    for (int j = 0; j < n; ++j) {
        const double s = unscaled ? cs / w->D[j] : 1.0;
        const double a = fabs(s * (w->tn[j] + w->q[j] + w->tn2[j])); if (a > d) d = a;
        const double b = fabs(s * w->tn[j]); if (b > npx) npx = b;
        const double c = fabs(s * w->tn2[j]); if (c > naty) naty = c;
        const double e = fabs(s * w->q[j]); if (e > nq) nq = e;
    }
    *dr = d;
    double mx = npx > naty ? npx : naty;
    *dn = mx > nq ? mx : nq;
}

/* returns 0 = solved, 1 = max iterations */
static int osqp_solve(osqp_work* w) {
    const int n = w->n, m = w->m;
    const double alpha = w->st.alpha, sigma = w->st.sigma;
    int it;
    w->status = 1;
    for (it = 1; it <= w->st.max_iter; ++it) {
        memcpy(w->x_prev, w->x, sizeof(double) * n);
        memcpy(w->z_prev, w->z, sizeof(double) * (size_t)m);
        for (int j = 0; j < n; ++j) w->rhs[j] = sigma * w->x_prev[j] - w->q[j];
        for (int i = 0; i < m; ++i) w->rhs[n + i] = w->z_prev[i] - w->y[i] / w->rho_vec[i];
        ldl_solve(&w->F, w->rhs);
        for (int j = 0; j < n; ++j) w->xt[j] = w->rhs[j];
        for (int i = 0; i < m; ++i) w->zt[i] = w->z_prev[i] + (w->rhs[n + i] - w->y[i]) / w->rho_vec[i];
        for (int j = 0; j < n; ++j) w->x[j] = alpha * w->xt[j] + (1.0 - alpha) * w->x_prev[j];
        for (int i = 0; i < m; ++i) {
            const double zr = alpha * w->zt[i] + (1.0 - alpha) * w->z_prev[i];
            double zn = zr + w->y[i] / w->rho_vec[i];
            zn = zn < w->l[i] ? w->l[i] : (zn > w->u[i] ? w->u[i] : zn);
            w->z[i] = zn;
            w->y[i] += w->rho_vec[i] * (zr - zn);
        }
        const int check = w->st.check_termination && (it % w->st.check_termination == 0);
        const int adapt = w->st.adaptive_rho && w->st.adaptive_rho_interval && (it % w->st.adaptive_rho_interval == 0);
        if (check) {
            double pr, dr, pn, dn;
            osqp_residuals(w, 1, &pr, &dr, &pn, &dn);
            const double ep = w->st.eps_abs + w->st.eps_rel * pn;
            const double ed = w->st.eps_abs + w->st.eps_rel * dn;
            if (pr <= ep && dr <= ed) { w->status = 0; break; }
        }
        if (adapt) {
            double pr, dr, pn, dn;
            osqp_residuals(w, 0, &pr, &dr, &pn, &dn);
            pr /= (pn + 1e-10); dr /= (dn + 1e-10);
            double rn = w->rho * sqrt(pr / (dr + 1e-10));
            rn = rn < RHO_MIN ? RHO_MIN : (rn > RHO_MAX ? RHO_MAX : rn);
            if (rn > w->rho * w->st.adaptive_rho_tolerance || rn < w->rho / w->st.adaptive_rho_tolerance) osqp_update_rho(w, rn);
        }
    }
    w->iters = it > w->st.max_iter ? w->st.max_iter : it;
    return w->status;
}

static void osqp_get_x(const osqp_work* w, double* x) { for (int j = 0; j < w->n; ++j) x[j] = w->D[j] * w->x[j]; }

/* ---------------- generic entry for tests: dense in, OSQP-restatement out ----------- */
int orc_osqp_dense(int n, int m, const double* P, const double* q, const double* A, const double* l, const double* u,
                   double eps_abs, double eps_rel, int max_iter, double* x_out, int* iters_out) {
    osqp_settings st; osqp_default_settings(&st);
    if (eps_abs > 0) st.eps_abs = eps_abs;
    if (eps_rel > 0) st.eps_rel = eps_rel;
    if (max_iter > 0) st.max_iter = max_iter;
    csc* Pc = csc_from_dense(P, n, n, 1);
    csc* Ac = csc_from_dense(A, m, n, 0);
    osqp_work* w = osqp_setup(n, m, Pc, q, Ac, l, u, &st);
    int rc = w->status < 0 ? -1 : osqp_solve(w);
    osqp_get_x(w, x_out);
    if (iters_out) *iters_out = w->iters;
    osqp_cleanup(w); csc_free(Pc); csc_free(Ac);
    return rc;
}

/* =====================================================================================
 * MPC batch through the OSQP restatement (cold start per instance = BASELINE config 2:
 * a new MPCSolver/OSQP workspace, WalkingDCMModelPredictiveController.cpp:415-420).
 * ===================================================================================== */
typedef struct { int N; double dT, com_height, gravity, Q[4], R[4]; } orc_mpc_params;

int orc_mpc_batch_osqp(const orc_mpc_params* p, int batch,
                       const double* x0, const double* ref, int ref_len, const double* u_prev,
                       const double* hull_A, const double* hull_b, const int* hull_nc,
                       double* u0_out, int* iters_out, int* status_out, int nthreads) {
    const int N = p->N, nx = 2 * (N + 1), nu = 2 * N, n = nx + nu;
    /* P = blkdiag(Qtilde, Theta' Rtilde Theta)  (…PredictiveController.cpp:38-77,126-144) */
    double* Pd = (double*)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i <= N; ++i)
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) Pd[(size_t)(2 * i + r) * n + 2 * i + c] = p->Q[2 * r + c];
    for (int i = 0; i < N; ++i)
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) {
            const double Rrc = p->R[2 * r + c];
            Pd[(size_t)(nx + 2 * i + r) * n + nx + 2 * i + c] += Rrc;                     /* u_i' R u_i        */
            if (i + 1 < N) {
                Pd[(size_t)(nx + 2 * i + r) * n + nx + 2 * i + c] += Rrc;                 /* from stage i+1    */
                Pd[(size_t)(nx + 2 * i + r) * n + nx + 2 * (i + 1) + c] -= Rrc;           /* cross terms       */
                Pd[(size_t)(nx + 2 * (i + 1) + r) * n + nx + 2 * i + c] -= Rrc;
            }
        }
    const double omega = sqrt(p->gravity / p->com_height);
    const double a = exp(omega * p->dT), b = 1.0 - a;                                      /* :230-237 */
    csc* Pc = csc_from_dense(Pd, n, n, 1);
    free(Pd);
    int nfail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(+ : nfail)
#endif
    for (int inst = 0; inst < batch; ++inst) {
        int nc = hull_nc[inst]; nc = nc < 0 ? 0 : (nc > 8 ? 8 : nc);
        const int m = nx + nc;
        double* Ad = (double*)calloc((size_t)m * n, sizeof(double));
        for (int i = 0; i < nx; ++i) Ad[(size_t)i * n + i] = -1.0;                        /* :102-103 */
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < 2; ++r) {
                Ad[(size_t)(2 * (i + 1) + r) * n + 2 * i + r] = a;                         /* :105-109 */
                Ad[(size_t)(2 * (i + 1) + r) * n + nx + 2 * i + r] = b;                    /* :118-122 */
            }
        for (int k = 0; k < nc; ++k) {                                                     /* MPCSolver.cpp:82-86 */
            Ad[(size_t)(nx + k) * n + nx] = hull_A[(size_t)inst * 16 + 2 * k];
            Ad[(size_t)(nx + k) * n + nx + 1] = hull_A[(size_t)inst * 16 + 2 * k + 1];
        }
        double* q = (double*)calloc((size_t)n, sizeof(double));
        double* l = (double*)calloc((size_t)m, sizeof(double));
        double* u = (double*)calloc((size_t)m, sizeof(double));
        for (int i = 0; i <= N; ++i) {                                                     /* MPCSolver.cpp:188-215 */
            const int ir = i < ref_len ? i : ref_len - 1;
            const double* r = ref + ((size_t)inst * ref_len + ir) * 2;
            q[2 * i] = -(p->Q[0] * r[0] + p->Q[1] * r[1]);
            q[2 * i + 1] = -(p->Q[2] * r[0] + p->Q[3] * r[1]);
        }
        q[nx] = -(p->R[0] * u_prev[2 * inst] + p->R[1] * u_prev[2 * inst + 1]);            /* :244-245 */
        q[nx + 1] = -(p->R[2] * u_prev[2 * inst] + p->R[3] * u_prev[2 * inst + 1]);
        l[0] = u[0] = -x0[2 * inst]; l[1] = u[1] = -x0[2 * inst + 1];                      /* :143-146 */
        for (int k = 0; k < nc; ++k) { l[nx + k] = -OSQP_INFTY_; u[nx + k] = hull_b[(size_t)inst * 8 + k]; }  /* :48-49,152-153 */
        csc* Ac = csc_from_dense(Ad, m, n, 0);
        osqp_settings st; osqp_default_settings(&st);
        osqp_work* w = osqp_setup(n, m, Pc, q, Ac, l, u, &st);
        const int rc = w->status < 0 ? -1 : osqp_solve(w);
        double* xs = (double*)malloc(sizeof(double) * n);
        osqp_get_x(w, xs);
        u0_out[2 * inst] = xs[nx]; u0_out[2 * inst + 1] = xs[nx + 1];                      /* cpp:510-511 */
        if (iters_out) iters_out[inst] = w->iters;
        if (status_out) status_out[inst] = rc;
        if (rc != 0) nfail++;
        free(xs); osqp_cleanup(w); csc_free(Ac); free(Ad); free(q); free(l); free(u);
    }
    csc_free(Pc);
    return nfail;
}

/* MPCSolver::setBounds / setGradient on an EXISTING workspace (MPCSolver.cpp:157-173, 249-258: updateBounds,
 * updateGradient - osqp_update_bounds / osqp_update_lin_cost upstream): new data in, scaled like the old. */
static void osqp_update_q(osqp_work* w, const double* q) { for (int j = 0; j < w->n; ++j) w->q[j] = w->c * w->D[j] * q[j]; }
static void osqp_update_bounds(osqp_work* w, const double* l, const double* u) {
    for (int i = 0; i < w->m; ++i) {
        const double li = l[i] < -OSQP_INFTY_ ? -OSQP_INFTY_ : l[i], ui = u[i] > OSQP_INFTY_ ? OSQP_INFTY_ : u[i];
        w->l[i] = w->E[i] * li; w->u[i] = w->E[i] * ui;
    }
}

/* =====================================================================================
 * MPC batch, WARM ticks: what the reference does on every tick without a contact change (>= 97 % of them): the
 * OSQP workspace of the previous tick is kept, only the bounds (x0, hull b) and the gradient (reference window
 * shifted by one stage, u_prev) are updated, and solve() warm-starts from the previous (x, y)
 * (MPCSolver.cpp:157-173, 216-258, 297-314).  Per instance: one cold set-up + solve (not timed by the caller's
 * `warm_seconds`), then `warm_ticks` warm solves on a window that advances one stage per tick
 * (ref must hold ref_len >= N + 1 + warm_ticks stages; x0 follows the LIPM under the applied input).
 * Returns the number of non-converged warm solves; *warm_seconds = wall time of the warm solves of all threads' work.
 * ===================================================================================== */
int orc_mpc_batch_osqp_warm(const orc_mpc_params* p, int batch, int warm_ticks,
                            const double* x0, const double* ref, int ref_len, const double* u_prev,
                            const double* hull_A, const double* hull_b, const int* hull_nc,
                            double* u0_out, double* mean_iters_warm, double* warm_seconds, int nthreads) {
    const int N = p->N, nx = 2 * (N + 1), nu = 2 * N, n = nx + nu;
    if (ref_len < N + 1 + warm_ticks) return -1;
    double* Pd = (double*)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i <= N; ++i)
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) Pd[(size_t)(2 * i + r) * n + 2 * i + c] = p->Q[2 * r + c];
    for (int i = 0; i < N; ++i)
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) {
            const double Rrc = p->R[2 * r + c];
            Pd[(size_t)(nx + 2 * i + r) * n + nx + 2 * i + c] += Rrc;
            if (i + 1 < N) {
                Pd[(size_t)(nx + 2 * i + r) * n + nx + 2 * i + c] += Rrc;
                Pd[(size_t)(nx + 2 * i + r) * n + nx + 2 * (i + 1) + c] -= Rrc;
                Pd[(size_t)(nx + 2 * (i + 1) + r) * n + nx + 2 * i + c] -= Rrc;
            }
        }
    const double omega = sqrt(p->gravity / p->com_height);
    const double a = exp(omega * p->dT), b = 1.0 - a;
    csc* Pc = csc_from_dense(Pd, n, n, 1);
    free(Pd);
    int nfail = 0;
    long iters_total = 0;
    double t_warm = 0.0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(+ : nfail, iters_total, t_warm)
#endif
    for (int inst = 0; inst < batch; ++inst) {
        int nc = hull_nc[inst]; nc = nc < 0 ? 0 : (nc > 8 ? 8 : nc);
        const int m = nx + nc;
        double* Ad = (double*)calloc((size_t)m * n, sizeof(double));
        for (int i = 0; i < nx; ++i) Ad[(size_t)i * n + i] = -1.0;
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < 2; ++r) {
                Ad[(size_t)(2 * (i + 1) + r) * n + 2 * i + r] = a;
                Ad[(size_t)(2 * (i + 1) + r) * n + nx + 2 * i + r] = b;
            }
        for (int k = 0; k < nc; ++k) {
            Ad[(size_t)(nx + k) * n + nx] = hull_A[(size_t)inst * 16 + 2 * k];
            Ad[(size_t)(nx + k) * n + nx + 1] = hull_A[(size_t)inst * 16 + 2 * k + 1];
        }
        double* q = (double*)calloc((size_t)n, sizeof(double));
        double* l = (double*)calloc((size_t)m, sizeof(double));
        double* u = (double*)calloc((size_t)m, sizeof(double));
        double* xs = (double*)malloc(sizeof(double) * n);
        double xm[2] = {x0[2 * inst], x0[2 * inst + 1]}, up[2] = {u_prev[2 * inst], u_prev[2 * inst + 1]};
        osqp_work* w = NULL;
        csc* Ac = csc_from_dense(Ad, m, n, 0);
        for (int t = 0; t <= warm_ticks; ++t) {
            for (int i = 0; i <= N; ++i) {
                const double* r = ref + ((size_t)inst * ref_len + t + i) * 2;
                q[2 * i] = -(p->Q[0] * r[0] + p->Q[1] * r[1]);
                q[2 * i + 1] = -(p->Q[2] * r[0] + p->Q[3] * r[1]);
            }
            q[nx] = -(p->R[0] * up[0] + p->R[1] * up[1]);
            q[nx + 1] = -(p->R[2] * up[0] + p->R[3] * up[1]);
            l[0] = u[0] = -xm[0]; l[1] = u[1] = -xm[1];
            for (int k = 0; k < nc; ++k) { l[nx + k] = -OSQP_INFTY_; u[nx + k] = hull_b[(size_t)inst * 8 + k]; }
            int rc;
            if (t == 0) {
                osqp_settings st; osqp_default_settings(&st);
                w = osqp_setup(n, m, Pc, q, Ac, l, u, &st);
                rc = w->status < 0 ? -1 : osqp_solve(w);
            } else {
                const double t0 = now_seconds();
                osqp_update_bounds(w, l, u);
                osqp_update_q(w, q);
                rc = osqp_solve(w);                     /* (x, y, z) of the previous tick are the starting point */
                t_warm += now_seconds() - t0;
                iters_total += w->iters;
                if (rc != 0) nfail++;
            }
            osqp_get_x(w, xs);
            up[0] = xs[nx]; up[1] = xs[nx + 1];
            xm[0] = a * xm[0] + b * up[0]; xm[1] = a * xm[1] + b * up[1];          /* the plant follows the LIPM */
        }
        u0_out[2 * inst] = up[0]; u0_out[2 * inst + 1] = up[1];
        free(xs); osqp_cleanup(w); csc_free(Ac); free(Ad); free(q); free(l); free(u);
    }
    csc_free(Pc);
    if (mean_iters_warm) *mean_iters_warm = warm_ticks > 0 ? (double)iters_total / ((double)batch * warm_ticks) : 0.0;
    if (warm_seconds) *warm_seconds = t_warm;
    return nfail;
}

/* =====================================================================================
 * IK assembly (both back-ends) + solve
 * ===================================================================================== */
typedef struct {
    int dof, use_com, form;                 /* form 0 = qpOASES, 1 = osqp */
    double Wc[9], Wn[9];
    double w[32], gains[32], qreg[32], vmin[32], vmax[32];
    double k_pos_com, k_pos_foot, k_att_foot, k_neck;
} orc_ik_params;

static void rot_err3(const double* R, const double* Rd, double* e) {
    double E[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        double s = 0; for (int k = 0; k < 3; ++k) s += R[3 * a + k] * Rd[3 * b + k];
        E[3 * a + b] = s;                                  /* R * Rd^-1, Rd^-1 = Rd' */
    }
    e[0] = 0.5 * (E[7] - E[5]); e[1] = 0.5 * (E[2] - E[6]); e[2] = 0.5 * (E[3] - E[1]);   /* Utils.cpp:22-27 + unskew */
}

/* builds H (n x n), g (n), A (meq x n), b (meq) for one instance */
static void ik_assemble(const orc_ik_params* p, const double* JL, const double* JR, const double* JN, const double* JC,
                        const double* q, const double* s, double* H, double* g, double* A, double* b) {
    const int dof = p->dof, n = dof + 6, meq = p->use_com ? 15 : 12;
    memset(H, 0, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < dof; ++j) H[(size_t)(6 + j) * n + 6 + j] = p->w[j];                /* base.cpp:64-67 */
    double WJ[3 * 64];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < n; ++c)
        WJ[r * n + c] = p->Wn[3 * r] * JN[c] + p->Wn[3 * r + 1] * JN[n + c] + p->Wn[3 * r + 2] * JN[2 * n + c];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)                                 /* osqp.cpp:142-145 */
        H[(size_t)i * n + j] += JN[i] * WJ[j] + JN[n + i] * WJ[n + j] + JN[2 * n + i] * WJ[2 * n + j];
    double en[3], t[3], y[3];
    rot_err3(s + 48, s + 57, en);
    const double kappa = p->form == 1 ? p->k_att_foot : 1.0;                               /* osqp.cpp:183 vs qp.cpp:164 */
    for (int k = 0; k < 3; ++k) t[k] = kappa * (-p->k_neck * en[k]);
    for (int r = 0; r < 3; ++r) y[r] = p->Wn[3 * r] * t[0] + p->Wn[3 * r + 1] * t[1] + p->Wn[3 * r + 2] * t[2];
    for (int i = 0; i < n; ++i) g[i] = -(JN[i] * y[0] + JN[n + i] * y[1] + JN[2 * n + i] * y[2]);
    for (int j = 0; j < dof; ++j) g[6 + j] -= p->w[j] * p->gains[j] * (p->qreg[j] - q[j]); /* base.cpp:70-72 */
    if (!p->use_com) {
        double WC[3 * 64], wv[3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < n; ++c)
            WC[r * n + c] = p->Wc[3 * r] * JC[c] + p->Wc[3 * r + 1] * JC[n + c] + p->Wc[3 * r + 2] * JC[2 * n + c];
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)                             /* osqp.cpp:147-151 */
            H[(size_t)i * n + j] += JC[i] * WC[j] + JC[n + i] * WC[n + j] + JC[2 * n + i] * WC[2 * n + j];
        for (int r = 0; r < 3; ++r) wv[r] = p->Wc[3 * r] * s[72] + p->Wc[3 * r + 1] * s[73] + p->Wc[3 * r + 2] * s[74];
        for (int i = 0; i < n; ++i) g[i] -= JC[i] * wv[0] + JC[n + i] * wv[1] + JC[2 * n + i] * wv[2];
    }
    memcpy(A, JL, sizeof(double) * 6 * (size_t)n);                                          /* osqp.cpp:221-226 */
    memcpy(A + 6 * (size_t)n, JR, sizeof(double) * 6 * (size_t)n);
    if (p->use_com) memcpy(A + 12 * (size_t)n, JC, sizeof(double) * 3 * (size_t)n);
    for (int foot = 0; foot < 2; ++foot) {                                                  /* osqp.cpp:268-306 */
        const double* pp = s + (foot ? 12 : 0); const double* R = s + (foot ? 15 : 3);
        const double* pd = s + (foot ? 36 : 24); const double* Rd = s + (foot ? 39 : 27);
        const double* tw = s + (foot ? 81 : 75);
        double e[3];
        rot_err3(R, Rd, e);
        const int skip = p->form == 1 && tw[0] == tw[1] && tw[0] == 0.0;
        for (int k = 0; k < 3; ++k) {
            b[6 * foot + k] = skip ? tw[k] : tw[k] - p->k_pos_foot * (pp[k] - pd[k]);
            b[6 * foot + 3 + k] = skip ? tw[3 + k] : tw[3 + k] - p->k_att_foot * e[k];
        }
    }
    if (p->use_com) for (int k = 0; k < 3; ++k) b[12 + k] = s[72 + k] - p->k_pos_com * (s[66 + k] - s[69 + k]);
    (void)meq;
}

/* dense Cholesky helpers */
static int chol_dense(double* M, int n) {
    for (int j = 0; j < n; ++j) {
        double d = M[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= M[(size_t)j * n + k] * M[(size_t)j * n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d); M[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = M[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= M[(size_t)i * n + k] * M[(size_t)j * n + k];
            M[(size_t)i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const double* L, int n, double* b) {
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
}

/* Goldfarb-Idnani dual active set on  min 1/2 x'Hx + g'x, A x = b, lb <= x <= ub (dense).
 * returns 0 solved, 1 max iterations, 2 infeasible, 4 numeric */
static int as_solve(int n, int meq, const double* H, const double* g, const double* A, const double* b,
                    const double* lb, const double* ub, int max_iter, double* x, uint32_t* act_lo, uint32_t* act_up, int* iters) {
    double* M = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* P = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* G = (double*)malloc(sizeof(double) * (size_t)n * (meq > 0 ? meq : 1));
    double* S = (double*)malloc(sizeof(double) * (size_t)(meq > 0 ? meq * meq : 1));
    double* gt = (double*)malloc(sizeof(double) * n);
    double* tmp = (double*)malloc(sizeof(double) * n);
    int rc = 0;
    memcpy(M, H, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int r = 0; r < meq; ++r) s += A[(size_t)r * n + i] * A[(size_t)r * n + j]; M[(size_t)i * n + j] += s; }
    for (int i = 0; i < n; ++i) { double s = 0; for (int r = 0; r < meq; ++r) s += A[(size_t)r * n + i] * b[r]; gt[i] = g[i] - s; }
    if (chol_dense(M, n) != 0) { rc = 4; goto done; }
    /* Minv columns -> P (start as Minv) */
    for (int j = 0; j < n; ++j) { memset(tmp, 0, sizeof(double) * n); tmp[j] = 1.0; chol_solve(M, n, tmp); for (int i = 0; i < n; ++i) P[(size_t)i * n + j] = tmp[i]; }
    for (int r = 0; r < meq; ++r) for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += P[(size_t)i * n + j] * A[(size_t)r * n + j]; G[(size_t)i * meq + r] = s; }
    for (int r = 0; r < meq; ++r) for (int c = 0; c < meq; ++c) { double s = 0; for (int i = 0; i < n; ++i) s += A[(size_t)r * n + i] * G[(size_t)i * meq + c]; S[(size_t)r * meq + c] = s; }
    if (meq && chol_dense(S, meq) != 0) { rc = 4; goto done; }
    {
        /* x = -Minv gt - G lam, lam = -Sinv (G'gt + b) */
        double lam[32], u[64];
        for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += P[(size_t)i * n + j] * gt[j]; u[i] = s; }
        for (int r = 0; r < meq; ++r) { double s = b[r]; for (int i = 0; i < n; ++i) s += A[(size_t)r * n + i] * u[i]; lam[r] = -s; }
        if (meq) chol_solve(S, meq, lam);
        for (int i = 0; i < n; ++i) { double s = -u[i]; for (int r = 0; r < meq; ++r) s -= G[(size_t)i * meq + r] * lam[r]; x[i] = s; }
        /* P = Minv - G Sinv G' */
        double col[32];
        for (int j = 0; j < n; ++j) {
            for (int r = 0; r < meq; ++r) col[r] = G[(size_t)j * meq + r];
            if (meq) chol_solve(S, meq, col);
            for (int i = 0; i < n; ++i) { double s = 0; for (int r = 0; r < meq; ++r) s += G[(size_t)i * meq + r] * col[r]; P[(size_t)i * n + j] -= s; }
        }
    }
    {
        int W[64], nW = 0, it = 0;
        double sg[64], mu[64], R[64 * 64], r[64], c[64], z[64];
        char inW[64];
        memset(inW, 0, sizeof(inW));
        for (;;) {
            int p = -1; double s = 1e-12, sig = 1.0;
            for (int i = 0; i < n; ++i) {
                if (inW[i]) continue;
                const double vh = x[i] - ub[i], vl = lb[i] - x[i];
                const double v = vh > vl ? vh : vl;
                if (v > s) { s = v; p = i; sig = vh >= vl ? 1.0 : -1.0; }
            }
            if (p < 0) break;
            if (it >= max_iter) { rc = 1; break; }
            ++it;
            double mu_p = 0.0;
            int inner_guard = 0;
            for (;;) {
                for (int a2 = 0; a2 < nW; ++a2) {
                    for (int b2 = 0; b2 < nW; ++b2) R[a2 * nW + b2] = sg[a2] * sg[b2] * P[(size_t)W[a2] * n + W[b2]];
                    c[a2] = sg[a2] * sig * P[(size_t)W[a2] * n + p];
                }
                if (nW) { if (chol_dense(R, nW) != 0) { rc = 4; goto done; } memcpy(r, c, sizeof(double) * nW); chol_solve(R, nW, r); }
                for (int i = 0; i < n; ++i) { double zi = sig * P[(size_t)i * n + p]; for (int a2 = 0; a2 < nW; ++a2) zi -= r[a2] * sg[a2] * P[(size_t)i * n + W[a2]]; z[i] = zi; }
                const double nz = sig * z[p];
                const double t2 = nz > 1e-12 * P[(size_t)p * n + p] ? s / nz : INFINITY;
                double t1 = INFINITY; int jd = -1;
                for (int a2 = 0; a2 < nW; ++a2) if (r[a2] > 0 && mu[a2] / r[a2] < t1) { t1 = mu[a2] / r[a2]; jd = a2; }
                const double t = t1 < t2 ? t1 : t2;
                if (!(t < INFINITY)) { rc = 2; goto done; }
                for (int i = 0; i < n; ++i) x[i] -= t * z[i];
                for (int a2 = 0; a2 < nW; ++a2) mu[a2] -= t * r[a2];
                mu_p += t; s -= t * nz;
                if (t2 <= t1) { W[nW] = p; sg[nW] = sig; mu[nW] = mu_p; inW[p] = 1; ++nW; break; }
                inW[W[jd]] = 0;
                for (int a2 = jd; a2 < nW - 1; ++a2) { W[a2] = W[a2 + 1]; sg[a2] = sg[a2 + 1]; mu[a2] = mu[a2 + 1]; }
                --nW; ++it;
                if (++inner_guard > n + 2) { rc = 1; goto done; }
            }
        }
        *act_lo = 0; *act_up = 0;
        for (int a2 = 0; a2 < nW; ++a2) { if (W[a2] >= 6) { if (sg[a2] > 0) *act_up |= 1u << (W[a2] - 6); else *act_lo |= 1u << (W[a2] - 6); } }
        if (iters) *iters = it;
    }
done:
    free(M); free(P); free(G); free(S); free(gt); free(tmp);
    return rc;
}

int orc_ik_batch(const orc_ik_params* p, int batch,
                 const double* JL, const double* JR, const double* JN, const double* JC,
                 const double* q, const double* state,
                 double* dq_out, int* status_out, uint32_t* act_lo, uint32_t* act_up, int* iters_out, int nthreads) {
    const int dof = p->dof, n = dof + 6, meq = p->use_com ? 15 : 12;
    int nfail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(+ : nfail)
#endif
    for (int inst = 0; inst < batch; ++inst) {
        double H[64 * 64], g[64], A[15 * 64], b[16], x[64];
        ik_assemble(p, JL + (size_t)inst * 6 * n, JR + (size_t)inst * 6 * n, JN + (size_t)inst * 3 * n, JC + (size_t)inst * 3 * n,
                    q + (size_t)inst * dof, state + (size_t)inst * 87, H, g, A, b);
        int rc, it = 0; uint32_t lo = 0, up = 0;
        if (p->form == 0) {
            /* qpOASES form: variable bounds, base +-DBL_MAX (qp.cpp:39-49), nWSR = 100 (qp.cpp:312) */
            double lb[64], ub[64];
            for (int i = 0; i < 6; ++i) { lb[i] = -DBL_MAX; ub[i] = DBL_MAX; }
            for (int j = 0; j < dof; ++j) { lb[6 + j] = p->vmin[j]; ub[6 + j] = p->vmax[j]; }
            rc = as_solve(n, meq, H, g, A, b, lb, ub, 100, x, &lo, &up, &it);
        } else {
            /* osqp form: m = meq + dof, the last dof rows of A are ZERO (osqp.hpp:18, osqp.cpp:227-235) */
            const int m = meq + dof;
            double* Ad = (double*)calloc((size_t)m * n, sizeof(double));
            double l[64], u[64];
            memcpy(Ad, A, sizeof(double) * (size_t)meq * n);
            for (int r = 0; r < meq; ++r) l[r] = u[r] = b[r];
            for (int j = 0; j < dof; ++j) { l[meq + j] = p->vmin[j]; u[meq + j] = p->vmax[j]; }   /* osqp.cpp:45-49 */
            csc* Pc = csc_from_dense(H, n, n, 1);        /* dense -> sparseView (osqp.cpp:153) */
            csc* Ac = csc_from_dense(Ad, m, n, 0);
            osqp_settings st; osqp_default_settings(&st);
            osqp_work* w = osqp_setup(n, m, Pc, g, Ac, l, u, &st);
            rc = w->status < 0 ? 4 : osqp_solve(w);
            osqp_get_x(w, x);
            it = w->iters;
            osqp_cleanup(w); csc_free(Pc); csc_free(Ac); free(Ad);
        }
        for (int j = 0; j < dof; ++j) dq_out[(size_t)inst * dof + j] = x[6 + j];           /* osqp.cpp:424-425, qp.cpp:357-358 */
        if (status_out) status_out[inst] = rc;
        if (act_lo) act_lo[inst] = lo;
        if (act_up) act_up[inst] = up;
        if (iters_out) iters_out[inst] = it;
        if (rc != 0) nfail++;
    }
    return nfail;
}

/* =====================================================================================
 * (4) The SAME direct methods the device kernels run, in plain C on the host cores: bench.py times them as
 * `cpu_baseline.same_algorithm_qps`, so that the record separates "a better algorithm" (condensing instead of ADMM, range
 * space instead of a 29-variable active set) from "an MI355X".  Restated from this repository's own DESIGN.md 2 / 4.2
 * (tools/ik4_proto.py is the numpy prototype of the IK part); checked against oracle/qp_spec.py in tests/test_cpu_oracle.py.
 * ===================================================================================== */

/* DCM-MPC, condensed: u0_unc = sum_i Gr_i r_i + Gx x0 + Gu u_prev (rows of the inverse of the constant equality KKT, computed by
 * the caller), then the projection of u0_unc onto the support polygon in the Sigma0^-1 metric by enumeration of {no row, one
 * row, two rows} active.  Returns the number of instances that are not SOLVED (status 2 = infeasible). */
int orc_mpc_batch_condensed(int N, const double* Gr, const double* Gx, const double* Gu, const double* S0, double feas_tol, int batch,
                            const double* x0, const double* ref, int ref_len, const double* u_prev,
                            const double* hull_A, const double* hull_b, const int* hull_nc,
                            double* u0_out, uint32_t* active_out, int* status_out, int nthreads) {
    int nfail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(+ : nfail)
#endif
    for (int inst = 0; inst < batch; ++inst) {
        const double* r = ref + (size_t)inst * ref_len * 2;
        double ux = 0.0, uy = 0.0;
        for (int i = 0; i <= N; ++i) {
            const double* ri = r + 2 * (i < ref_len ? i : ref_len - 1);            /* MPCSolver.cpp:200-214 (constant tail) */
            const double* g = Gr + 4 * i;
            ux += g[0] * ri[0] + g[1] * ri[1]; uy += g[2] * ri[0] + g[3] * ri[1];
        }
        const double* xs = x0 + 2 * (size_t)inst; const double* up = u_prev + 2 * (size_t)inst;
        ux += Gx[0] * xs[0] + Gx[1] * xs[1] + Gu[0] * up[0] + Gu[1] * up[1];
        uy += Gx[2] * xs[0] + Gx[3] * xs[1] + Gu[2] * up[0] + Gu[3] * up[1];
        const double* A = hull_A + (size_t)inst * 16; const double* b = hull_b + (size_t)inst * 8;
        int nc = hull_nc[inst]; nc = nc < 0 ? 0 : (nc > 8 ? 8 : nc);
        double best = INFINITY, bx = ux, by = uy; uint32_t bm = 0; int found = 0;
        /* candidates: (e, f) with f == e meaning "row e alone" and e == -1 "no row"; first hit among equal costs wins */
        for (int e = -1; e < nc; ++e) {
            for (int f = e; f < nc; ++f) {
                const int single = f == e;
                double px = ux, py = uy, cost = 0.0; uint32_t mask = 0; int ok = 1;
                if (e >= 0) {
                    const double sex = S0[0] * A[2 * e] + S0[1] * A[2 * e + 1], sey = S0[2] * A[2 * e] + S0[3] * A[2 * e + 1];
                    const double ree = A[2 * e] * sex + A[2 * e + 1] * sey, re = A[2 * e] * ux + A[2 * e + 1] * uy - b[e];
                    mask = 1u << e;
                    if (single) { ok = ree > 0.0; const double mu = ok ? re / ree : 0.0; px = ux - sex * mu; py = uy - sey * mu; cost = mu * re; }
                    else {
                        const double sfx = S0[0] * A[2 * f] + S0[1] * A[2 * f + 1], sfy = S0[2] * A[2 * f] + S0[3] * A[2 * f + 1];
                        const double rff = A[2 * f] * sfx + A[2 * f + 1] * sfy, ref_ = A[2 * e] * sfx + A[2 * e + 1] * sfy;
                        const double rf = A[2 * f] * ux + A[2 * f + 1] * uy - b[f], det = ree * rff - ref_ * ref_;
                        ok = det > 1e-12 * ree * rff;                          /* parallel rows have no vertex */
                        const double mue = ok ? (rff * re - ref_ * rf) / det : 0.0, muf = ok ? (ree * rf - ref_ * re) / det : 0.0;
                        px = ux - sex * mue - sfx * muf; py = uy - sey * mue - sfy * muf; cost = mue * re + muf * rf; mask |= 1u << f;
                    }
                }
                for (int k = 0; k < nc && ok; ++k) if (k != e && k != f && A[2 * k] * px + A[2 * k + 1] * py - b[k] > feas_tol) ok = 0;
                if (ok && cost < best) { best = cost; bx = px; by = py; bm = mask; found = 1; }
                if (e < 0) break;                                              /* "no row" is one candidate */
            }
        }
        u0_out[2 * (size_t)inst] = bx; u0_out[2 * (size_t)inst + 1] = by;
        if (active_out) active_out[inst] = bm;
        if (status_out) status_out[inst] = found ? 0 : 2;
        if (!found) nfail++;
    }
    return nfail;
}

/* QP-IK, base unknowns eliminated in closed form through the left-foot rows (MIXED free-floating Jacobians: base blocks
 * [I B; 0 I]), the remaining 23-variable QP in range space: C = [L'(J_Nq - J_Lq,ang); A] (12 x 23), D = Lam^-1,
 * M = C D C' + diag(I3, 0), M y = -(C D gq + [t; b']), x = -D (gq + C' y); joint-velocity bounds (qpOASES form) by the
 * Goldfarb-Idnani dual active set on columns of P = D - D C' M^-1 C D.  CoM as constraint, every joint weight > 0.
 * status: 0 solved, 1 max iterations, 2 infeasible, 4 numeric. */
int orc_ik_batch_range_space(const orc_ik_params* p, int batch,
                             const double* JLa, const double* JRa, const double* JNa, const double* JCa,
                             const double* qa, const double* state,
                             double* dq_out, int* status_out, uint32_t* act_lo, uint32_t* act_up, int* iters_out, int nthreads) {
    enum { NJ = 23, NV = 29, NRW = 12 };
    if (p->dof != NJ || !p->use_com) return -1;
    double Lc[9];                                                          /* neck weight W = L L' */
    memcpy(Lc, p->Wn, sizeof(Lc));
    if (chol_dense(Lc, 3) != 0) return -1;
    int nfail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(+ : nfail)
#endif
    for (int inst = 0; inst < batch; ++inst) {
        const double* JL = JLa + (size_t)inst * 6 * NV; const double* JR = JRa + (size_t)inst * 6 * NV;
        const double* JN = JNa + (size_t)inst * 3 * NV; const double* JC = JCa + (size_t)inst * 3 * NV;
        const double* q = qa + (size_t)inst * NJ; const double* s = state + (size_t)inst * 87;
        double bt[18];                                                     /* [b_L; b_R; b_C; e_neck] */
        for (int foot = 0; foot < 2; ++foot) {
            const double* pp = s + (foot ? 12 : 0); const double* R = s + (foot ? 15 : 3);
            const double* pd = s + (foot ? 36 : 24); const double* Rd = s + (foot ? 39 : 27); const double* tw = s + (foot ? 81 : 75);
            double e[3]; rot_err3(R, Rd, e);
            const int skip = p->form == 1 && tw[0] == tw[1] && tw[0] == 0.0;
            for (int k = 0; k < 3; ++k) {
                bt[6 * foot + k] = skip ? tw[k] : tw[k] - p->k_pos_foot * (pp[k] - pd[k]);
                bt[6 * foot + 3 + k] = skip ? tw[3 + k] : tw[3 + k] - p->k_att_foot * e[k];
            }
        }
        for (int k = 0; k < 3; ++k) bt[12 + k] = s[72 + k] - p->k_pos_com * (s[66 + k] - s[69 + k]);
        { double en[3]; rot_err3(s + 48, s + 57, en); const double kap = (p->form == 1 ? p->k_att_foot : 1.0) * (-p->k_neck); for (int k = 0; k < 3; ++k) bt[15 + k] = kap * en[k]; }
        double dBR[9], dBC[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { dBR[3 * r + c] = JR[r * NV + 3 + c] - JL[r * NV + 3 + c]; dBC[3 * r + c] = JC[r * NV + 3 + c] - JL[r * NV + 3 + c]; }
        double C[NRW][NJ + 1], D[NJ], gq[NJ];
        for (int c = 0; c <= NJ; ++c) {                                    /* column c of [J_L; J_R; J_C; J_N | rhs] -> column c of [C | d] */
            double a[18];
            if (c < NJ) { for (int r = 0; r < 6; ++r) { a[r] = JL[r * NV + 6 + c]; a[6 + r] = JR[r * NV + 6 + c]; } for (int r = 0; r < 3; ++r) { a[12 + r] = JC[r * NV + 6 + c]; a[15 + r] = JN[r * NV + 6 + c]; } }
            else memcpy(a, bt, sizeof(a));
            const double* w = a + 3; const double n3[3] = {a[15] - w[0], a[16] - w[1], a[17] - w[2]};
            for (int r = 0; r < 3; ++r) { double t = 0; for (int k = r; k < 3; ++k) t += Lc[3 * k + r] * n3[k]; C[r][c] = t; }        /* L' n */
            for (int r = 0; r < 3; ++r) {
                C[3 + r][c] = a[6 + r] - a[r] - (dBR[3 * r] * w[0] + dBR[3 * r + 1] * w[1] + dBR[3 * r + 2] * w[2]);
                C[6 + r][c] = a[9 + r] - w[r];
                C[9 + r][c] = a[12 + r] - a[r] - (dBC[3 * r] * w[0] + dBC[3 * r + 1] * w[1] + dBC[3 * r + 2] * w[2]);
            }
        }
        for (int j = 0; j < NJ; ++j) { D[j] = 1.0 / p->w[j]; gq[j] = -p->w[j] * p->gains[j] * (p->qreg[j] - q[j]); }
        double M[NRW * NRW], Mi[NRW * NRW], y[NRW], x[NJ];
        for (int i = 0; i < NRW; ++i) for (int k = 0; k < NRW; ++k) { double t = (i == k && i < 3) ? 1.0 : 0.0; for (int j = 0; j < NJ; ++j) t += C[i][j] * D[j] * C[k][j]; M[i * NRW + k] = t; }
        int rc = chol_dense(M, NRW) != 0 ? 4 : 0, it = 0; uint32_t lo = 0, up = 0;
        if (rc == 0) {
            for (int i = 0; i < NRW; ++i) { double t = C[i][NJ]; for (int j = 0; j < NJ; ++j) t += C[i][j] * D[j] * gq[j]; y[i] = -t; }
            chol_solve(M, NRW, y);
            for (int j = 0; j < NJ; ++j) { double t = gq[j]; for (int i = 0; i < NRW; ++i) t += C[i][j] * y[i]; x[j] = -D[j] * t; }
            int need = 0;
            if (p->form == 0) for (int j = 0; j < NJ; ++j) if (x[j] - p->vmax[j] > 1e-12 || p->vmin[j] - x[j] > 1e-12) need = 1;
            if (need) {
                double P[NJ][NJ], T[NRW][NJ];
                for (int k = 0; k < NRW; ++k) { double e[NRW]; memset(e, 0, sizeof(e)); e[k] = 1.0; chol_solve(M, NRW, e); for (int i = 0; i < NRW; ++i) Mi[i * NRW + k] = e[i]; }
                for (int i = 0; i < NRW; ++i) for (int j = 0; j < NJ; ++j) { double t = 0; for (int k = 0; k < NRW; ++k) t += Mi[i * NRW + k] * C[k][j]; T[i][j] = t * D[j]; }
                for (int a = 0; a < NJ; ++a) for (int b2 = 0; b2 < NJ; ++b2) { double t = a == b2 ? D[a] : 0.0; for (int i = 0; i < NRW; ++i) t -= D[a] * C[i][a] * T[i][b2]; P[a][b2] = t; }
                int W[NJ], nW = 0; double sg[NJ], mu[NJ], R[NJ * NJ], r[NJ], c[NJ], z[NJ]; char inW[NJ]; memset(inW, 0, sizeof(inW));
                for (;;) {
                    int pv = -1; double sv = 1e-12, sig = 1.0;
                    for (int i = 0; i < NJ; ++i) { if (inW[i]) continue; const double vh = x[i] - p->vmax[i], vl = p->vmin[i] - x[i], v = vh > vl ? vh : vl; if (v > sv) { sv = v; pv = i; sig = vh >= vl ? 1.0 : -1.0; } }
                    if (pv < 0) break;
                    if (it >= 100) { rc = 1; break; }
                    ++it;
                    double mu_p = 0.0; int guard = 0;
                    for (;;) {
                        for (int a = 0; a < nW; ++a) { for (int b2 = 0; b2 < nW; ++b2) R[a * nW + b2] = sg[a] * sg[b2] * P[W[a]][W[b2]]; c[a] = sg[a] * sig * P[W[a]][pv]; }
                        if (nW) { if (chol_dense(R, nW) != 0) { rc = 4; break; } memcpy(r, c, sizeof(double) * nW); chol_solve(R, nW, r); }
                        for (int i = 0; i < NJ; ++i) { double zi = sig * P[i][pv]; for (int a = 0; a < nW; ++a) zi -= r[a] * sg[a] * P[i][W[a]]; z[i] = zi; }
                        const double nz = sig * z[pv], t2 = (nW < NJ - 9 && nz > 1e-10 * P[pv][pv]) ? sv / nz : INFINITY;
                        double t1 = INFINITY; int jd = -1;
                        for (int a = 0; a < nW; ++a) if (r[a] > 0 && mu[a] / r[a] < t1) { t1 = mu[a] / r[a]; jd = a; }
                        const double t = t1 < t2 ? t1 : t2;
                        if (!(t < INFINITY)) { rc = 2; break; }
                        for (int i = 0; i < NJ; ++i) x[i] -= t * z[i];
                        for (int a = 0; a < nW; ++a) mu[a] -= t * r[a];
                        mu_p += t; sv -= t * nz;
                        if (t2 <= t1) { W[nW] = pv; sg[nW] = sig; mu[nW] = mu_p; inW[pv] = 1; ++nW; break; }
                        inW[W[jd]] = 0;
                        for (int a = jd; a < nW - 1; ++a) { W[a] = W[a + 1]; sg[a] = sg[a + 1]; mu[a] = mu[a + 1]; }
                        --nW; ++it;
                        if (++guard > NJ + 2) { rc = 1; break; }
                    }
                    if (rc != 0) break;
                }
                for (int a = 0; a < nW; ++a) { if (sg[a] > 0) { up |= 1u << W[a]; if (rc == 0) x[W[a]] = p->vmax[W[a]]; } else { lo |= 1u << W[a]; if (rc == 0) x[W[a]] = p->vmin[W[a]]; } }
            }
        }
        for (int j = 0; j < NJ; ++j) dq_out[(size_t)inst * NJ + j] = rc == 0 ? x[j] : 0.0;
        if (status_out) status_out[inst] = rc;
        if (act_lo) act_lo[inst] = lo;
        if (act_up) act_up[inst] = up;
        if (iters_out) iters_out[inst] = it;
        if (rc != 0) nfail++;
    }
    return nfail;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
