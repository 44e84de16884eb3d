/* TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * CPU restatement of libmpc++'s linear-MPC solve path, one instance at a time:
 *   - the QP the reference builds        (include/mpc/LMPC/ProblemBuilder.hpp:184-633,642-825)
 *   - dense -> CSC conversion per solve  (ProblemBuilder.hpp:54-67, LOptimizer.hpp:425-478)
 *   - OSQP set-up + solve per call       (LOptimizer.hpp:189-368) -> osqp_restate.c
 *   - unpacking into Result/OptSequence  (LOptimizer.hpp:305-361), status map (:386-415)
 * All matrices are column-major doubles (Types.hpp:42).  Index conventions and
 * quirks follow SURVEY.md section 8(a) (the +1 column shift of weights/bounds,
 * bounds on x_u(i)=u(i-1), "i > ch" delta-u pinning, ...).
 *
 * PARITY STATUS: pinned by the reference's known answers
 * (test/LMPC/test_common.cpp:230-236, test/LMPC/test_constraints.cpp:183-294)
 * in tests/test_oracle.py.
 */
#include "osqp_restate.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int nx, nu, ndu, ny, ph, ch, na;
    int nvar, neq, nineq, ncon;
    double *ssA, *ssB, *ssC, *ssBv, *ssDv;       /* na x na, na x nu, (ny+nu) x na, na x ndu, (ny+nu) x ndu */
    double *wOutput, *wU, *wDeltaU;              /* ny x (ph+1), nu x (ph+1), nu x ph */
    double *minX, *maxX, *minY, *maxY, *minU, *maxU;
    double *sMin, *sMax, *sX, *sU;
    double *P, *A;                               /* dense, column-major */
    double *lineq, *uineq;
    double *q, *l, *u;
    oq_cache *cache;
    double *warm_x, *warm_y; int have_warm;
    double last_cmd[64];
} lmpc_t;

#define M2(a, ld, i, j) (a)[(size_t)(i) + (size_t)(j) * (size_t)(ld)]

static double *dalloc(size_t n) { return (double *)calloc(n > 0 ? n : 1, sizeof(double)); }
static void fill(double *a, size_t n, double v) { for (size_t i = 0; i < n; i++) a[i] = v; }

static void build_time_invariant(lmpc_t *h);

void *oracle_lmpc_create(int nx, int nu, int ndu, int ny, int ph, int ch)
{
    lmpc_t *h = (lmpc_t *)calloc(1, sizeof(lmpc_t));
    h->nx = nx; h->nu = nu; h->ndu = ndu; h->ny = ny; h->ph = ph; h->ch = ch; h->na = nx + nu;
    int na = h->na;
    h->nvar = (ph + 1) * na + ph * nu;
    h->neq = (ph + 1) * na;
    h->nineq = (ph + 1) * na + (ph + 1) * ny + ph * nu + (ph + 1);
    h->ncon = h->neq + h->nineq;
    h->ssA = dalloc((size_t)na * na); h->ssB = dalloc((size_t)na * nu);
    h->ssC = dalloc((size_t)(ny + nu) * na);
    h->ssBv = dalloc((size_t)na * ndu); h->ssDv = dalloc((size_t)(ny + nu) * ndu);
    h->wOutput = dalloc((size_t)ny * (ph + 1)); h->wU = dalloc((size_t)nu * (ph + 1));
    h->wDeltaU = dalloc((size_t)nu * ph);
    h->minX = dalloc((size_t)nx * (ph + 1)); h->maxX = dalloc((size_t)nx * (ph + 1));
    h->minY = dalloc((size_t)ny * (ph + 1)); h->maxY = dalloc((size_t)ny * (ph + 1));
    h->minU = dalloc((size_t)nu * ph); h->maxU = dalloc((size_t)nu * ph);
    fill(h->minX, (size_t)nx * (ph + 1), -INFINITY); fill(h->maxX, (size_t)nx * (ph + 1), INFINITY);
    fill(h->minY, (size_t)ny * (ph + 1), -INFINITY); fill(h->maxY, (size_t)ny * (ph + 1), INFINITY);
    fill(h->minU, (size_t)nu * ph, -INFINITY); fill(h->maxU, (size_t)nu * ph, INFINITY);
    h->sMin = dalloc((size_t)ph + 1); h->sMax = dalloc((size_t)ph + 1);
    fill(h->sMin, (size_t)ph + 1, -INFINITY); fill(h->sMax, (size_t)ph + 1, INFINITY);
    h->sX = dalloc((size_t)nx); h->sU = dalloc((size_t)nu);
    h->P = dalloc((size_t)h->nvar * h->nvar); h->A = dalloc((size_t)h->ncon * h->nvar);
    h->lineq = dalloc((size_t)h->nineq); h->uineq = dalloc((size_t)h->nineq);
    h->q = dalloc((size_t)h->nvar); h->l = dalloc((size_t)h->ncon); h->u = dalloc((size_t)h->ncon);
    h->warm_x = dalloc((size_t)h->nvar); h->warm_y = dalloc((size_t)h->ncon);
    h->cache = oq_cache_new();
    build_time_invariant(h);
    return h;
}

void oracle_lmpc_destroy(void *hv)
{
    lmpc_t *h = (lmpc_t *)hv;
    if (!h) return;
    free(h->ssA); free(h->ssB); free(h->ssC); free(h->ssBv); free(h->ssDv);
    free(h->wOutput); free(h->wU); free(h->wDeltaU);
    free(h->minX); free(h->maxX); free(h->minY); free(h->maxY); free(h->minU); free(h->maxU);
    free(h->sMin); free(h->sMax); free(h->sX); free(h->sU);
    free(h->P); free(h->A); free(h->lineq); free(h->uineq); free(h->q); free(h->l); free(h->u);
    free(h->warm_x); free(h->warm_y);
    oq_cache_free(h->cache);
    free(h);
}

void oracle_lmpc_sizes(void *hv, int *nvar, int *ncon, int *neq)
{
    lmpc_t *h = (lmpc_t *)hv; *nvar = h->nvar; *ncon = h->ncon; *neq = h->neq;
}

/* ---- setters: ProblemBuilder.hpp:184-504 --------------------------------- */
int oracle_lmpc_set_model(void *hv, const double *A, const double *B, const double *C)
{
    lmpc_t *h = (lmpc_t *)hv;
    int nx = h->nx, nu = h->nu, ny = h->ny, na = h->na;
    memset(h->ssA, 0, sizeof(double) * na * na);
    memset(h->ssB, 0, sizeof(double) * na * nu);
    memset(h->ssC, 0, sizeof(double) * (ny + nu) * na);
    for (int j = 0; j < nx; j++) for (int i = 0; i < nx; i++) M2(h->ssA, na, i, j) = M2(A, nx, i, j);
    for (int j = 0; j < nu; j++) for (int i = 0; i < nx; i++) {
        M2(h->ssA, na, i, nx + j) = M2(B, nx, i, j);
        M2(h->ssB, na, i, j) = M2(B, nx, i, j);
    }
    for (int j = 0; j < nu; j++) { M2(h->ssA, na, nx + j, nx + j) = 1.0; M2(h->ssB, na, nx + j, j) = 1.0; }
    for (int j = 0; j < nx; j++) for (int i = 0; i < ny; i++) M2(h->ssC, ny + nu, i, j) = M2(C, ny, i, j);
    for (int j = 0; j < nu; j++) M2(h->ssC, ny + nu, ny + j, nx + j) = 1.0;
    build_time_invariant(h);
    return 1;
}

int oracle_lmpc_set_exogenous(void *hv, const double *Bd, const double *Dd)
{
    lmpc_t *h = (lmpc_t *)hv;
    int nx = h->nx, nu = h->nu, ny = h->ny, na = h->na, ndu = h->ndu;
    memset(h->ssBv, 0, sizeof(double) * na * ndu);
    memset(h->ssDv, 0, sizeof(double) * (ny + nu) * ndu);
    for (int j = 0; j < ndu; j++) {
        for (int i = 0; i < nx; i++) M2(h->ssBv, na, i, j) = M2(Bd, nx, i, j);
        for (int i = 0; i < ny; i++) M2(h->ssDv, ny + nu, i, j) = M2(Dd, ny, i, j);
    }
    build_time_invariant(h);
    return 1;
}

/* matrix form: user column k -> internal k+1; internal 0 := user 0 */
static void shift_in(double *dst, const double *src, int rows, int ph)
{
    for (int k = 0; k < ph; k++) memcpy(dst + (size_t)(k + 1) * rows, src + (size_t)k * rows, sizeof(double) * rows);
    memcpy(dst, src, sizeof(double) * rows);
}

int oracle_lmpc_set_objective(void *hv, const double *OW, const double *UW, const double *DUW)
{
    lmpc_t *h = (lmpc_t *)hv;
    shift_in(h->wOutput, OW, h->ny, h->ph);
    shift_in(h->wU, UW, h->nu, h->ph);
    memcpy(h->wDeltaU, DUW, sizeof(double) * h->nu * h->ph);
    build_time_invariant(h);
    return 1;
}

int oracle_lmpc_set_objective_idx(void *hv, int idx, const double *ow, const double *uw, const double *duw)
{
    lmpc_t *h = (lmpc_t *)hv;
    memcpy(h->wOutput + (size_t)(idx + 1) * h->ny, ow, sizeof(double) * h->ny);
    memcpy(h->wU + (size_t)(idx + 1) * h->nu, uw, sizeof(double) * h->nu);
    if (idx == 0) { memcpy(h->wOutput, ow, sizeof(double) * h->ny); memcpy(h->wU, uw, sizeof(double) * h->nu); }
    memcpy(h->wDeltaU + (size_t)idx * h->nu, duw, sizeof(double) * h->nu);
    build_time_invariant(h);
    return 1;
}

int oracle_lmpc_set_state_bounds(void *hv, const double *lo, const double *hi)
{
    lmpc_t *h = (lmpc_t *)hv;
    shift_in(h->minX, lo, h->nx, h->ph); shift_in(h->maxX, hi, h->nx, h->ph);
    build_time_invariant(h); return 1;
}
int oracle_lmpc_set_state_bounds_idx(void *hv, int idx, const double *lo, const double *hi)
{
    lmpc_t *h = (lmpc_t *)hv;
    memcpy(h->minX + (size_t)(idx + 1) * h->nx, lo, sizeof(double) * h->nx);
    memcpy(h->maxX + (size_t)(idx + 1) * h->nx, hi, sizeof(double) * h->nx);
    if (idx == 0) { memcpy(h->minX, lo, sizeof(double) * h->nx); memcpy(h->maxX, hi, sizeof(double) * h->nx); }
    build_time_invariant(h); return 1;
}
int oracle_lmpc_set_output_bounds(void *hv, const double *lo, const double *hi)
{
    lmpc_t *h = (lmpc_t *)hv;
    shift_in(h->minY, lo, h->ny, h->ph); shift_in(h->maxY, hi, h->ny, h->ph);
    build_time_invariant(h); return 1;
}
int oracle_lmpc_set_output_bounds_idx(void *hv, int idx, const double *lo, const double *hi)
{
    lmpc_t *h = (lmpc_t *)hv;
    memcpy(h->minY + (size_t)(idx + 1) * h->ny, lo, sizeof(double) * h->ny);
    memcpy(h->maxY + (size_t)(idx + 1) * h->ny, hi, sizeof(double) * h->ny);
    if (idx == 0) { memcpy(h->minY, lo, sizeof(double) * h->ny); memcpy(h->maxY, hi, sizeof(double) * h->ny); }
    build_time_invariant(h); return 1;
}
/* nu x ch in; columns past ch replicate column ch-1 (ProblemBuilder.hpp:402-410) */
int oracle_lmpc_set_input_bounds(void *hv, const double *lo, const double *hi)
{
    lmpc_t *h = (lmpc_t *)hv;
    int nu = h->nu, ch = h->ch, ph = h->ph;
    memcpy(h->minU, lo, sizeof(double) * nu * ch); memcpy(h->maxU, hi, sizeof(double) * nu * ch);
    for (int k = ch; k < ph; k++) {
        memcpy(h->minU + (size_t)k * nu, lo + (size_t)(ch - 1) * nu, sizeof(double) * nu);
        memcpy(h->maxU + (size_t)k * nu, hi + (size_t)(ch - 1) * nu, sizeof(double) * nu);
    }
    build_time_invariant(h); return 1;
}
int oracle_lmpc_set_input_bounds_idx(void *hv, int idx, const double *lo, const double *hi)
{
    lmpc_t *h = (lmpc_t *)hv;
    memcpy(h->minU + (size_t)idx * h->nu, lo, sizeof(double) * h->nu);
    memcpy(h->maxU + (size_t)idx * h->nu, hi, sizeof(double) * h->nu);
    build_time_invariant(h); return 1;
}
int oracle_lmpc_set_scalar(void *hv, const double *smin, const double *smax, const double *X, const double *U)
{
    lmpc_t *h = (lmpc_t *)hv;
    memcpy(h->sMin + 1, smin, sizeof(double) * h->ph); h->sMin[0] = smin[0];
    memcpy(h->sMax + 1, smax, sizeof(double) * h->ph); h->sMax[0] = smax[0];
    memcpy(h->sX, X, sizeof(double) * h->nx); memcpy(h->sU, U, sizeof(double) * h->nu);
    build_time_invariant(h); return 1;
}
int oracle_lmpc_set_scalar_idx(void *hv, int idx, double smin, double smax, const double *X, const double *U)
{
    lmpc_t *h = (lmpc_t *)hv;
    h->sMin[idx + 1] = smin; h->sMax[idx + 1] = smax;
    if (idx == 0) { h->sMin[0] = smin; h->sMax[0] = smax; }
    memcpy(h->sX, X, sizeof(double) * h->nx); memcpy(h->sU, U, sizeof(double) * h->nu);
    build_time_invariant(h); return 1;
}

/* ---- time-invariant terms: ProblemBuilder.hpp:642-825 --------------------- */
static void build_time_invariant(lmpc_t *h)
{
    int nx = h->nx, nu = h->nu, ny = h->ny, ph = h->ph, ch = h->ch, na = h->na;
    int nv = h->nvar, nc = h->ncon, nyu = ny + nu;
    memset(h->P, 0, sizeof(double) * (size_t)nv * nv);
    memset(h->A, 0, sizeof(double) * (size_t)nc * nv);
    double *blk = dalloc((size_t)na * na);
    for (int i = 0; i <= ph; i++) {
        /* ssC' * diag(wOutput_i, wU_i) * ssC */
        for (int a = 0; a < na; a++)
            for (int b = 0; b < na; b++) {
                double s = 0;
                for (int k = 0; k < nyu; k++) {
                    double wk = k < ny ? M2(h->wOutput, ny, k, i) : M2(h->wU, nu, k - ny, i);
                    s += M2(h->ssC, nyu, k, a) * wk * M2(h->ssC, nyu, k, b);
                }
                M2(h->P, nv, i * na + a, i * na + b) = s;
            }
        if (i < ph)
            for (int j = 0; j < nu; j++) {
                int o = (ph + 1) * na + i * nu + j;
                M2(h->P, nv, o, o) = M2(h->wDeltaU, nu, j, i);
            }
    }
    free(blk);
    /* dynamics equalities */
    for (int i = 0; i <= ph; i++) {
        for (int a = 0; a < na; a++) M2(h->A, nc, i * na + a, i * na + a) = -1.0;
        if (i > 0) {
            for (int a = 0; a < na; a++) {
                for (int b = 0; b < na; b++) M2(h->A, nc, i * na + a, (i - 1) * na + b) += M2(h->ssA, na, a, b);
                for (int b = 0; b < nu; b++) M2(h->A, nc, i * na + a, (ph + 1) * na + (i - 1) * nu + b) = M2(h->ssB, na, a, b);
            }
        }
    }
    int r0 = h->neq;
    int off_y = (ph + 1) * na, off_du = off_y + (ph + 1) * ny, off_s = off_du + ph * nu;
    for (int k = 0; k < (ph + 1) * na; k++) M2(h->A, nc, r0 + k, k) = 1.0;
    for (int i = 0; i <= ph; i++)
        for (int a = 0; a < ny; a++)
            for (int b = 0; b < na; b++) M2(h->A, nc, r0 + off_y + i * ny + a, i * na + b) = M2(h->ssC, nyu, a, b);
    for (int k = 0; k < ph * nu; k++) M2(h->A, nc, r0 + off_du + k, (ph + 1) * na + k) = 1.0;
    for (int i = 0; i <= ph; i++) {
        for (int b = 0; b < nx; b++) M2(h->A, nc, r0 + off_s + i, i * na + b) = h->sX[b];
        for (int b = 0; b < nu; b++) M2(h->A, nc, r0 + off_s + i, i * na + nx + b) = h->sU[b];
    }
    for (int i = 0; i <= ph; i++) {
        int k = i == ph ? i - 1 : i;       /* x_u(i) bounded by input column min(i, ph-1) */
        for (int a = 0; a < nx; a++) { h->lineq[i * na + a] = M2(h->minX, nx, a, i); h->uineq[i * na + a] = M2(h->maxX, nx, a, i); }
        for (int a = 0; a < nu; a++) { h->lineq[i * na + nx + a] = M2(h->minU, nu, a, k); h->uineq[i * na + nx + a] = M2(h->maxU, nu, a, k); }
    }
    for (int k = 0; k < (ph + 1) * ny; k++) { h->lineq[off_y + k] = h->minY[k]; h->uineq[off_y + k] = h->maxY[k]; }
    for (int i = 0; i < ph; i++)
        for (int a = 0; a < nu; a++) {
            h->lineq[off_du + i * nu + a] = (i > ch) ? 0.0 : -INFINITY;
            h->uineq[off_du + i * nu + a] = (i > ch) ? 0.0 : INFINITY;
        }
    for (int i = 0; i <= ph; i++) { h->lineq[off_s + i] = h->sMin[i]; h->uineq[off_s + i] = h->sMax[i]; }
}

/* ---- per-solve vectors: ProblemBuilder.hpp:528-633 ------------------------ */
static void get_problem(lmpc_t *h, const double *x0, const double *u0, const double *yRef,
                        const double *uRef, const double *duRef, const double *dMeas)
{
    int nx = h->nx, nu = h->nu, ny = h->ny, ph = h->ph, na = h->na, ndu = h->ndu, nyu = ny + nu;
    int off_y = (ph + 1) * na;
    memset(h->q, 0, sizeof(double) * h->nvar);
    double *leq = h->l;                    /* first neq entries of l */
    memset(h->l, 0, sizeof(double) * h->ncon);
    double *off = dalloc((size_t)h->nineq);
    double *e = dalloc((size_t)nyu);
    for (int i = 0; i <= ph; i++) {
        int k = i == 0 ? 0 : i - 1;
        for (int a = 0; a < nyu; a++) {
            double r = a < ny ? M2(yRef, ny, a, k) : M2(uRef, nu, a - ny, k);
            double dv = 0;
            for (int d = 0; d < ndu; d++) dv += M2(h->ssDv, nyu, a, d) * M2(dMeas, ndu, d, k);
            double wk = a < ny ? M2(h->wOutput, ny, a, i) : M2(h->wU, nu, a - ny, i);
            e[a] = wk * (-r + dv);
        }
        for (int b = 0; b < na; b++) {
            double s = 0;
            for (int a = 0; a < nyu; a++) s += M2(h->ssC, nyu, a, b) * e[a];
            h->q[i * na + b] = s;
        }
        if (i < ph)
            for (int a = 0; a < nu; a++)
                h->q[(ph + 1) * na + i * nu + a] = -(M2(h->wDeltaU, nu, a, i) * M2(duRef, nu, a, k));
        if (i > 0)
            for (int a = 0; a < na; a++) {
                double s = 0;
                for (int d = 0; d < ndu; d++) s += M2(h->ssBv, na, a, d) * M2(dMeas, ndu, d, k);
                leq[i * na + a] = -s;
            }
        for (int a = 0; a < ny; a++) {
            double s = 0;
            for (int d = 0; d < ndu; d++) s += M2(h->ssDv, nyu, a, d) * M2(dMeas, ndu, d, k);
            off[off_y + i * ny + a] = -s;
        }
    }
    for (int a = 0; a < nx; a++) leq[a] = -x0[a];
    for (int a = 0; a < nu; a++) leq[nx + a] = -u0[a];
    memcpy(h->u, h->l, sizeof(double) * h->neq);
    for (int k = 0; k < h->nineq; k++) {
        h->l[h->neq + k] = h->lineq[k] + off[k];
        h->u[h->neq + k] = h->uineq[k] + off[k];
    }
    free(off); free(e);
}

/* QP-layout checks (test/LMPC/test_constraints.cpp:169-294): expose q, l, u, and dense P, A */
int oracle_lmpc_get_problem(void *hv, const double *x0, const double *u0, const double *yRef,
                            const double *uRef, const double *duRef, const double *dMeas,
                            double *P, double *q, double *A, double *l, double *u)
{
    lmpc_t *h = (lmpc_t *)hv;
    get_problem(h, x0, u0, yRef, uRef, duRef, dMeas);
    if (P) memcpy(P, h->P, sizeof(double) * (size_t)h->nvar * h->nvar);
    if (A) memcpy(A, h->A, sizeof(double) * (size_t)h->ncon * h->nvar);
    if (q) memcpy(q, h->q, sizeof(double) * h->nvar);
    if (l) memcpy(l, h->l, sizeof(double) * h->ncon);
    if (u) memcpy(u, h->u, sizeof(double) * h->ncon);
    return 1;
}

/* dense -> CSC, the sparseView() scan the reference repeats every solve */
static csc_t *dense_to_csc(const double *M, int rows, int cols, int upper)
{
    int nz = 0;
    for (int j = 0; j < cols; j++) {
        int lim = upper ? (j + 1 < rows ? j + 1 : rows) : rows;
        for (int i = 0; i < lim; i++) if (M2(M, rows, i, j) != 0.0) nz++;
    }
    csc_t *S = csc_alloc(rows, cols, nz);
    int q = 0;
    for (int j = 0; j < cols; j++) {
        S->p[j] = q;
        int lim = upper ? (j + 1 < rows ? j + 1 : rows) : rows;
        for (int i = 0; i < lim; i++) {
            double v = M2(M, rows, i, j);
            if (v != 0.0) { S->i[q] = i; S->x[q] = v; q++; }
        }
    }
    S->p[cols] = q;
    return S;
}

typedef struct {
    /* LParameters (Types.hpp:99-161) */
    int maximum_iteration; double time_limit; int enable_warm_start;
    double alpha, rho, eps_rel, eps_abs, eps_prim_inf, eps_dual_inf;
    int verbose, adaptive_rho, polish;
    /* oracle-only knobs */
    int adaptive_rho_interval;   /* 0 -> 25 */
    int nan_faithful;
} oracle_lparams;

void oracle_lparams_default(oracle_lparams *p)
{
    p->maximum_iteration = 100; p->time_limit = 0; p->enable_warm_start = 0;
    p->alpha = 1.6; p->rho = 1e-6; p->eps_rel = 1e-4; p->eps_abs = 1e-4;
    p->eps_prim_inf = 1e-3; p->eps_dual_inf = 1e-3;
    p->verbose = 0; p->adaptive_rho = 1; p->polish = 1;
    p->adaptive_rho_interval = 0; p->nan_faithful = 1;
}

/* ResultStatus (Types.hpp:87-94) */
enum { RS_SUCCESS = 0, RS_MAX_ITERATION = 1, RS_INFEASIBLE = 2, RS_ERROR = 3, RS_UNKNOWN = 4 };

static int to_result_status(int s)   /* LOptimizer.hpp:386-415 */
{
    switch (s) {
    case OQ_SOLVED: return RS_SUCCESS;
    case OQ_MAX_ITER_REACHED: return RS_MAX_ITERATION;
    case OQ_PRIMAL_INFEASIBLE: return RS_INFEASIBLE;
    case OQ_DUAL_INFEASIBLE: return RS_INFEASIBLE;
    case OQ_SOLVED_INACCURATE: return RS_SUCCESS;
    case OQ_PRIMAL_INFEASIBLE_INACCURATE: return RS_SUCCESS;
    case OQ_DUAL_INFEASIBLE_INACCURATE: return RS_SUCCESS;
    case OQ_SIGINT: return RS_ERROR;
    case OQ_TIME_LIMIT_REACHED: return RS_UNKNOWN;
    case OQ_NON_CVX: return RS_ERROR;
    case OQ_UNSOLVED: return RS_UNKNOWN;
    default: return RS_UNKNOWN;
    }
}

typedef struct {
    int solver_status, status, is_feasible, iters, polished, rho_updates;
    double cost, rho;
} oracle_result;

/* One LOptimizer::run.  Outputs (any may be NULL): cmd[nu], z[nvar], y[ncon],
 * seq_state[(ph+1) x nx], seq_output[(ph+1) x ny], seq_input[(ph+1) x nu]
 * (column-major, row i = step i, as OptSequence), act_lo/act_up[ncon] bytes. */
int oracle_lmpc_solve(void *hv, const oracle_lparams *prm,
                      const double *x0, const double *u0, const double *yRef, const double *uRef,
                      const double *duRef, const double *dMeas,
                      oracle_result *res, double *cmd, double *z_out, double *y_out,
                      double *seq_state, double *seq_output, double *seq_input,
                      unsigned char *act_lo, unsigned char *act_up)
{
    lmpc_t *h = (lmpc_t *)hv;
    int nx = h->nx, nu = h->nu, ny = h->ny, ph = h->ph, na = h->na, ndu = h->ndu, nyu = ny + nu;
    get_problem(h, x0, u0, yRef, uRef, duRef, dMeas);
    csc_t *Ps = dense_to_csc(h->P, h->nvar, h->nvar, 1);
    csc_t *As = dense_to_csc(h->A, h->ncon, h->nvar, 0);
    oq_settings s; oq_default_settings(&s);
    s.alpha = prm->alpha; s.rho = prm->rho; s.adaptive_rho = prm->adaptive_rho;
    s.eps_rel = prm->eps_rel; s.eps_abs = prm->eps_abs;
    s.eps_prim_inf = prm->eps_prim_inf; s.eps_dual_inf = prm->eps_dual_inf;
    s.max_iter = prm->maximum_iteration; s.polish = prm->polish;
    s.warm_start = prm->enable_warm_start && h->have_warm;
    if (prm->adaptive_rho_interval > 0) s.adaptive_rho_interval = prm->adaptive_rho_interval;
    s.nan_faithful = prm->nan_faithful;
    double *z = dalloc((size_t)h->nvar), *y = dalloc((size_t)h->ncon);
    oq_info info;
    oq_solve(Ps, h->q, As, h->l, h->u, &s, h->cache, h->warm_x, h->warm_y, z, y, &info, act_lo, act_up);
    memcpy(h->warm_x, z, sizeof(double) * h->nvar);
    memcpy(h->warm_y, y, sizeof(double) * h->ncon);
    h->have_warm = 1;
    res->solver_status = info.status; res->status = to_result_status(info.status);
    res->is_feasible = (info.status == OQ_SOLVED || info.status == OQ_SOLVED_INACCURATE || info.status == OQ_MAX_ITER_REACHED);
    res->iters = info.iters; res->polished = info.polished; res->rho_updates = info.rho_updates;
    res->cost = info.obj; res->rho = info.rho;
    /* LOptimizer.hpp:305-341 */
    for (int i = 0; i <= ph; i++) {
        int j = (i + 1 < ph + 1) ? i + 1 : i;
        int k = i == 0 ? 0 : i - 1;
        if (seq_state) for (int a = 0; a < nx; a++) M2(seq_state, ph + 1, i, a) = z[i * na + a];
        if (seq_input) for (int a = 0; a < nu; a++) M2(seq_input, ph + 1, i, a) = z[j * na + nx + a];
        if (seq_output)
            for (int a = 0; a < ny; a++) {
                double sv = 0;
                for (int b = 0; b < nx; b++) sv += M2(h->ssC, nyu, a, b) * z[i * na + b];
                for (int d = 0; d < ndu; d++) sv += M2(h->ssDv, nyu, a, d) * M2(dMeas, ndu, d, k);
                M2(seq_output, ph + 1, i, a) = sv;
            }
    }
    for (int a = 0; a < nu; a++) { double c = z[1 * na + nx + a]; if (ph < 1) c = z[nx + a]; if (cmd) cmd[a] = c; if (a < 64) h->last_cmd[a] = c; }
    if (z_out) memcpy(z_out, z, sizeof(double) * h->nvar);
    if (y_out) memcpy(y_out, y, sizeof(double) * h->ncon);
    free(z); free(y); csc_free(Ps); csc_free(As);
    return 0;
}

/* Batch driver for parity tests and the CPU baseline: instance b uses
 * x0[b*nx..], u0[b*nu..] and a per-instance output reference yref[b*ny..] held
 * constant along the horizon (the synthetic workload of SURVEY.md 8(d));
 * uRef = duRef = dMeas = 0.  Cold start every instance (fresh controller). */
double oracle_lmpc_solve_batch_constref(void *hv, const oracle_lparams *prm, int B,
                                        const double *x0, const double *u0, const double *yref,
                                        double *cmd, double *cost, int *status, int *solver_status,
                                        int *iters, int *polished, unsigned char *act_lo, unsigned char *act_up,
                                        double *per_solve_seconds)
{
    lmpc_t *h = (lmpc_t *)hv;
    int nx = h->nx, nu = h->nu, ny = h->ny, ph = h->ph, ndu = h->ndu;
    double *yR = dalloc((size_t)ny * ph), *uR = dalloc((size_t)nu * ph), *dM = dalloc((size_t)ndu * ph);
    struct timespec t0, t1, ta, tb;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int b = 0; b < B; b++) {
        for (int k = 0; k < ph; k++) memcpy(yR + (size_t)k * ny, yref + (size_t)b * ny, sizeof(double) * ny);
        oracle_result r;
        h->have_warm = 0;
        clock_gettime(CLOCK_MONOTONIC, &ta);
        oracle_lmpc_solve(hv, prm, x0 + (size_t)b * nx, u0 + (size_t)b * nu, yR, uR, uR, dM, &r,
                          cmd ? cmd + (size_t)b * nu : NULL, NULL, NULL, NULL, NULL, NULL,
                          act_lo ? act_lo + (size_t)b * h->ncon : NULL, act_up ? act_up + (size_t)b * h->ncon : NULL);
        clock_gettime(CLOCK_MONOTONIC, &tb);
        if (per_solve_seconds) per_solve_seconds[b] = (double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec);
        if (cost) cost[b] = r.cost;
        if (status) status[b] = r.status;
        if (solver_status) solver_status[b] = r.solver_status;
        if (iters) iters[b] = r.iters;
        if (polished) polished[b] = r.polished;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(yR); free(uR); free(dM);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
