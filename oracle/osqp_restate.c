/* TEST INFRASTRUCTURE ONLY (oracle).  See osqp_restate.h for provenance. */
#include "osqp_restate.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define OQ_INFTY 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_EQ_OVER_RHO_INEQ 1e3
#define RHO_TOL 1e-4
#define DIV_TOL (1.0 / OQ_INFTY)

void oq_default_settings(oq_settings *s)
{
    /* libmpc++ defaults: Types.hpp:108-114,150-160 */
    s->alpha = 1.6; s->rho = 1e-6; s->eps_rel = 1e-4; s->eps_abs = 1e-4;
    s->eps_prim_inf = 1e-3; s->eps_dual_inf = 1e-3;
    s->max_iter = 100; s->adaptive_rho = 1; s->polish = 1; s->warm_start = 0;
    /* OSQP v0.6.3 defaults */
    s->sigma = 1e-6; s->delta = 1e-6; s->adaptive_rho_tolerance = 5.0;
    s->scaling = 10; s->adaptive_rho_interval = 25; s->check_termination = 25;
    s->polish_refine_iter = 3; s->nan_faithful = 0;
}

struct oq_cache {
    uint64_t key;
    int n;
    int *perm;
};

oq_cache *oq_cache_new(void) { return (oq_cache *)calloc(1, sizeof(oq_cache)); }
void oq_cache_free(oq_cache *c) { if (c) { free(c->perm); free(c); } }

static uint64_t hash_ints(uint64_t h, const int *v, int n)
{
    for (int i = 0; i < n; i++) { h ^= (uint64_t)(unsigned)v[i]; h *= 0x100000001b3ull; }
    return h;
}

/* ---- small vector / matrix helpers ---------------------------------------- */
static double norm_inf(const double *v, int n)
{
    double r = 0; for (int i = 0; i < n; i++) { double a = fabs(v[i]); if (a > r) r = a; } return r;
}
static double scaled_norm_inf(const double *s, const double *v, int n)
{
    double r = 0; for (int i = 0; i < n; i++) { double a = fabs(s[i] * v[i]); if (a > r) r = a; } return r;
}
static void sym_triu_matvec(const csc_t *P, const double *x, double *y)
{
    int n = P->n;
    for (int i = 0; i < n; i++) y[i] = 0;
    for (int j = 0; j < n; j++)
        for (int p = P->p[j]; p < P->p[j + 1]; p++) {
            int i = P->i[p];
            y[i] += P->x[p] * x[j];
            if (i != j) y[j] += P->x[p] * x[i];
        }
}
static void mat_vec(const csc_t *A, const double *x, double *y)
{
    for (int i = 0; i < A->m; i++) y[i] = 0;
    for (int j = 0; j < A->n; j++) {
        double xj = x[j];
        for (int p = A->p[j]; p < A->p[j + 1]; p++) y[A->i[p]] += A->x[p] * xj;
    }
}
static void mat_tvec(const csc_t *A, const double *x, double *y)
{
    for (int j = 0; j < A->n; j++) {
        double s = 0;
        for (int p = A->p[j]; p < A->p[j + 1]; p++) s += A->x[p] * x[A->i[p]];
        y[j] = s;
    }
}
static csc_t *csc_copy(const csc_t *A)
{
    csc_t *B = csc_alloc(A->m, A->n, A->p[A->n]);
    memcpy(B->p, A->p, ((size_t)A->n + 1) * sizeof(int));
    memcpy(B->i, A->i, (size_t)A->p[A->n] * sizeof(int));
    memcpy(B->x, A->x, (size_t)A->p[A->n] * sizeof(double));
    return B;
}
static csc_t *csc_transpose(const csc_t *A)
{
    int nz = A->p[A->n];
    csc_t *T = csc_alloc(A->n, A->m, nz);
    int *w = (int *)calloc((size_t)A->m + 1, sizeof(int));
    for (int p = 0; p < nz; p++) w[A->i[p]]++;
    T->p[0] = 0;
    for (int i = 0; i < A->m; i++) { T->p[i + 1] = T->p[i] + w[i]; w[i] = T->p[i]; }
    for (int j = 0; j < A->n; j++)
        for (int p = A->p[j]; p < A->p[j + 1]; p++) {
            int q = w[A->i[p]]++;
            T->i[q] = j; T->x[q] = A->x[p];
        }
    free(w);
    return T;
}
static double limit1(double v)
{
    v = v < MIN_SCALING ? 1.0 : v;
    return v > MAX_SCALING ? MAX_SCALING : v;
}

/* ---- workspace ------------------------------------------------------------ */
typedef struct {
    int n, m;
    csc_t *P, *A, *At;        /* scaled data; At = A^T (row access) */
    double *q, *l, *u;
    double *D, *E, *Dinv, *Einv, c, cinv;
    int *ctype;
    double *rho_vec, *rho_inv;
    double rho;
    const oq_settings *s;
    csc_t *K; int *Kdiag;     /* KKT upper-tri + positions of the -1/rho diagonal */
    ldl_t *F;
    double *x, *z, *y, *x_prev, *z_prev, *xz, *delta_x, *delta_y;
    double *Ax, *Px, *Aty, *tn, *tm;
} work_t;

static void ruiz_scale(work_t *w)
{
    int n = w->n, m = w->m;
    double *Dt = w->tn, *Et = w->tm;
    double *Dta = (double *)malloc((size_t)n * sizeof(double));
    for (int it = 0; it < w->s->scaling; it++) {
        for (int j = 0; j < n; j++) { Dt[j] = 0; Dta[j] = 0; }
        for (int i = 0; i < m; i++) Et[i] = 0;
        for (int j = 0; j < n; j++)
            for (int p = w->P->p[j]; p < w->P->p[j + 1]; p++) {
                int i = w->P->i[p]; double a = fabs(w->P->x[p]);
                if (a > Dt[j]) Dt[j] = a;
                if (i != j && a > Dt[i]) Dt[i] = a;
            }
        for (int j = 0; j < n; j++)
            for (int p = w->A->p[j]; p < w->A->p[j + 1]; p++) {
                double a = fabs(w->A->x[p]);
                if (a > Dta[j]) Dta[j] = a;
                if (a > Et[w->A->i[p]]) Et[w->A->i[p]] = a;
            }
        for (int j = 0; j < n; j++) { double d = Dt[j] > Dta[j] ? Dt[j] : Dta[j]; Dt[j] = 1.0 / sqrt(limit1(d)); }
        for (int i = 0; i < m; i++) Et[i] = 1.0 / sqrt(limit1(Et[i]));
        for (int j = 0; j < n; j++)
            for (int p = w->P->p[j]; p < w->P->p[j + 1]; p++) w->P->x[p] *= Dt[j] * Dt[w->P->i[p]];
        for (int j = 0; j < n; j++)
            for (int p = w->A->p[j]; p < w->A->p[j + 1]; p++) w->A->x[p] *= Dt[j] * Et[w->A->i[p]];
        for (int j = 0; j < n; j++) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
        for (int i = 0; i < m; i++) w->E[i] *= Et[i];
        /* cost scaling */
        for (int j = 0; j < n; j++) Dt[j] = 0;
        for (int j = 0; j < n; j++)
            for (int p = w->P->p[j]; p < w->P->p[j + 1]; p++) {
                int i = w->P->i[p]; double a = fabs(w->P->x[p]);
                if (a > Dt[j]) Dt[j] = a;
                if (i != j && a > Dt[i]) Dt[i] = a;
            }
        double cm = 0; for (int j = 0; j < n; j++) cm += Dt[j];
        cm /= (double)n;
        double qn = limit1(norm_inf(w->q, n));
        double ct = cm > qn ? cm : qn;
        ct = 1.0 / limit1(ct);
        for (int p = 0; p < w->P->p[n]; p++) w->P->x[p] *= ct;
        for (int j = 0; j < n; j++) w->q[j] *= ct;
        w->c *= ct;
    }
    free(Dta);
    for (int j = 0; j < n; j++) w->Dinv[j] = 1.0 / w->D[j];
    for (int i = 0; i < m; i++) { w->Einv[i] = 1.0 / w->E[i]; w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
    w->cinv = 1.0 / w->c;
}

static void set_rho_vec(work_t *w, int first)
{
    for (int i = 0; i < w->m; i++) {
        if (first) {
            if (w->l[i] < -OQ_INFTY * MIN_SCALING && w->u[i] > OQ_INFTY * MIN_SCALING) w->ctype[i] = -1;
            else if (w->u[i] - w->l[i] < RHO_TOL) w->ctype[i] = 1;
            else w->ctype[i] = 0;
        }
        if (w->ctype[i] == -1) { if (first) w->rho_vec[i] = RHO_MIN; }
        else if (w->ctype[i] == 1) w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->rho;
        else w->rho_vec[i] = w->rho;
        w->rho_inv[i] = 1.0 / w->rho_vec[i];
    }
}

/* KKT = [P + sigma I, A'; A, -diag(1/rho)] upper triangle, CSC */
static void build_kkt(work_t *w)
{
    int n = w->n, m = w->m;
    int nz = w->P->p[n] + n + w->A->p[n] + m;
    w->K = csc_alloc(n + m, n + m, nz);
    w->Kdiag = (int *)malloc((size_t)(m > 0 ? m : 1) * sizeof(int));
    int q = 0;
    for (int j = 0; j < n; j++) {
        w->K->p[j] = q;
        double dj = w->s->sigma;
        for (int p = w->P->p[j]; p < w->P->p[j + 1]; p++) {
            int i = w->P->i[p];
            if (i == j) dj += w->P->x[p];
            else if (i < j) { w->K->i[q] = i; w->K->x[q] = w->P->x[p]; q++; }
        }
        w->K->i[q] = j; w->K->x[q] = dj; q++;
    }
    for (int r = 0; r < m; r++) {
        w->K->p[n + r] = q;
        for (int p = w->At->p[r]; p < w->At->p[r + 1]; p++) { w->K->i[q] = w->At->i[p]; w->K->x[q] = w->At->x[p]; q++; }
        w->Kdiag[r] = q;
        w->K->i[q] = n + r; w->K->x[q] = -w->rho_inv[r]; q++;
    }
    w->K->p[n + m] = q;
}

static void update_kkt_rho(work_t *w)
{
    for (int r = 0; r < w->m; r++) w->K->x[w->Kdiag[r]] = -w->rho_inv[r];
}

typedef struct { double pri, dua, eps_pri_norm, eps_dua_norm; } resid_t;

static void compute_residuals(work_t *w, const double *x, const double *z, const double *y, resid_t *r)
{
    int n = w->n, m = w->m;
    mat_vec(w->A, x, w->Ax);
    sym_triu_matvec(w->P, x, w->Px);
    mat_tvec(w->A, y, w->Aty);
    double pri = 0, nz_ = 0, nax = 0;
    for (int i = 0; i < m; i++) {
        double a = fabs(w->Einv[i] * (w->Ax[i] - z[i])); if (a > pri) pri = a;
        a = fabs(w->Einv[i] * z[i]); if (a > nz_) nz_ = a;
        a = fabs(w->Einv[i] * w->Ax[i]); if (a > nax) nax = a;
    }
    double dua = 0, nq = 0, naty = 0, npx = 0;
    for (int j = 0; j < n; j++) {
        double a = fabs(w->Dinv[j] * (w->Px[j] + w->q[j] + w->Aty[j])); if (a > dua) dua = a;
        a = fabs(w->Dinv[j] * w->q[j]); if (a > nq) nq = a;
        a = fabs(w->Dinv[j] * w->Aty[j]); if (a > naty) naty = a;
        a = fabs(w->Dinv[j] * w->Px[j]); if (a > npx) npx = a;
    }
    r->pri = pri; r->dua = w->cinv * dua;
    r->eps_pri_norm = nz_ > nax ? nz_ : nax;
    double t = nq > naty ? nq : naty; t = t > npx ? t : npx;
    r->eps_dua_norm = w->cinv * t;
}

static int is_primal_infeasible(work_t *w, double eps)
{
    int n = w->n, m = w->m;
    double *dy = w->delta_y;
    for (int i = 0; i < m; i++) {
        if (w->u[i] > OQ_INFTY * MIN_SCALING) {
            if (w->l[i] < -OQ_INFTY * MIN_SCALING) dy[i] = 0.0;
            else dy[i] = dy[i] < 0.0 ? dy[i] : 0.0;
        } else if (w->l[i] < -OQ_INFTY * MIN_SCALING) {
            dy[i] = dy[i] > 0.0 ? dy[i] : 0.0;
        }
    }
    double nrm = scaled_norm_inf(w->E, dy, m);
    if (nrm > DIV_TOL) {
        double lhs = 0;
        for (int i = 0; i < m; i++) {
            double pos = dy[i] > 0 ? dy[i] : 0, neg = dy[i] < 0 ? dy[i] : 0;
            if (w->s->nan_faithful) lhs += w->u[i] * pos + w->l[i] * neg;   /* inf*0 = NaN, as in C */
            else lhs += (pos != 0 ? w->u[i] * pos : 0) + (neg != 0 ? w->l[i] * neg : 0);
        }
        if (lhs < -eps * nrm) {
            mat_tvec(w->A, dy, w->tn);
            (void)n;
            return scaled_norm_inf(w->Dinv, w->tn, w->n) < eps * nrm;
        }
    }
    return 0;
}

static int is_dual_infeasible(work_t *w, double eps)
{
    int n = w->n, m = w->m;
    double nrm = scaled_norm_inf(w->D, w->delta_x, n);
    double cs = w->c;
    if (nrm > DIV_TOL) {
        double qdx = 0; for (int j = 0; j < n; j++) qdx += w->q[j] * w->delta_x[j];
        if (qdx < -cs * eps * nrm) {
            sym_triu_matvec(w->P, w->delta_x, w->tn);
            if (scaled_norm_inf(w->Dinv, w->tn, n) < cs * eps * nrm) {
                mat_vec(w->A, w->delta_x, w->tm);
                for (int i = 0; i < m; i++) {
                    double a = w->Einv[i] * w->tm[i];
                    if ((w->u[i] < OQ_INFTY * MIN_SCALING && a > eps * nrm) ||
                        (w->l[i] > -OQ_INFTY * MIN_SCALING && a < -eps * nrm)) return 0;
                }
                return 1;
            }
        }
    }
    return 0;
}

static int check_termination(work_t *w, int approx, int *status, resid_t *r)
{
    const oq_settings *s = w->s;
    compute_residuals(w, w->x, w->z, w->y, r);
    double ea = s->eps_abs, er = s->eps_rel, epi = s->eps_prim_inf, edi = s->eps_dual_inf;
    if (r->pri > OQ_INFTY || r->dua > OQ_INFTY) { *status = OQ_NON_CVX; return 1; }
    if (approx) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
    int pri_ok = 0, dua_ok = 0, pinf = 0, dinf = 0;
    if (w->m == 0) pri_ok = 1;
    else {
        double eps_pri = ea + er * r->eps_pri_norm;
        if (r->pri < eps_pri) pri_ok = 1; else pinf = is_primal_infeasible(w, epi);
    }
    double eps_dua = ea + er * r->eps_dua_norm;
    if (r->dua < eps_dua) dua_ok = 1; else dinf = is_dual_infeasible(w, edi);
    if (pri_ok && dua_ok) { *status = approx ? OQ_SOLVED_INACCURATE : OQ_SOLVED; return 1; }
    if (pinf) { *status = approx ? OQ_PRIMAL_INFEASIBLE_INACCURATE : OQ_PRIMAL_INFEASIBLE; return 1; }
    if (dinf) { *status = approx ? OQ_DUAL_INFEASIBLE_INACCURATE : OQ_DUAL_INFEASIBLE; return 1; }
    return 0;
}

static double rho_estimate(work_t *w)
{
    int n = w->n, m = w->m;
    mat_vec(w->A, w->x, w->Ax);
    sym_triu_matvec(w->P, w->x, w->Px);
    mat_tvec(w->A, w->y, w->Aty);
    double pr = 0, pn = 0;
    for (int i = 0; i < m; i++) {
        double a = fabs(w->Ax[i] - w->z[i]); if (a > pr) pr = a;
        a = fabs(w->z[i]); if (a > pn) pn = a;
        a = fabs(w->Ax[i]); if (a > pn) pn = a;
    }
    double dr = 0, dn = 0;
    for (int j = 0; j < n; j++) {
        double a = fabs(w->Px[j] + w->q[j] + w->Aty[j]); if (a > dr) dr = a;
        a = fabs(w->q[j]); if (a > dn) dn = a;
        a = fabs(w->Aty[j]); if (a > dn) dn = a;
        a = fabs(w->Px[j]); if (a > dn) dn = a;
    }
    pr /= (pn + DIV_TOL);
    dr /= (dn + DIV_TOL);
    double e = w->rho * sqrt(pr / dr);
    if (!(e > RHO_MIN)) e = RHO_MIN;       /* also catches NaN */
    if (e > RHO_MAX) e = RHO_MAX;
    return e;
}

/* ---- polish --------------------------------------------------------------- */
static int polish(work_t *w, oq_info *info, resid_t *r_admm, unsigned char *act_lo, unsigned char *act_up)
{
    int n = w->n, m = w->m;
    const oq_settings *s = w->s;
    int *rows = (int *)malloc((size_t)(2 * m + 1) * sizeof(int));
    int nlow = 0, nupp = 0;
    for (int i = 0; i < m; i++) { if (act_lo) act_lo[i] = 0; if (act_up) act_up[i] = 0; }
    for (int i = 0; i < m; i++)
        if (w->z[i] - w->l[i] < -w->y[i]) { rows[nlow++] = i; if (act_lo) act_lo[i] = 1; }
    for (int i = 0; i < m; i++)
        if (w->u[i] - w->z[i] < w->y[i]) { rows[nlow + nupp++] = i; if (act_up) act_up[i] = 1; }
    int na = nlow + nupp, N = n + na;
    /* reduced KKT upper triangle */
    int nz = w->P->p[n] + n + na;
    for (int k = 0; k < na; k++) nz += w->At->p[rows[k] + 1] - w->At->p[rows[k]];
    csc_t *K = csc_alloc(N, N, nz);
    int q = 0;
    for (int j = 0; j < n; j++) {
        K->p[j] = q;
        double dj = s->delta;
        for (int p = w->P->p[j]; p < w->P->p[j + 1]; p++) {
            int i = w->P->i[p];
            if (i == j) dj += w->P->x[p];
            else if (i < j) { K->i[q] = i; K->x[q] = w->P->x[p]; q++; }
        }
        K->i[q] = j; K->x[q] = dj; q++;
    }
    for (int k = 0; k < na; k++) {
        int rr = rows[k];
        K->p[n + k] = q;
        for (int p = w->At->p[rr]; p < w->At->p[rr + 1]; p++) { K->i[q] = w->At->i[p]; K->x[q] = w->At->x[p]; q++; }
        K->i[q] = n + k; K->x[q] = -s->delta; q++;
    }
    K->p[N] = q;
    /* ordering: the main KKT ordering restricted to the kept nodes */
    int *perm = (int *)malloc((size_t)N * sizeof(int));
    {
        int *map = (int *)malloc((size_t)(n + m) * sizeof(int));
        for (int i = 0; i < n + m; i++) map[i] = -1;
        for (int j = 0; j < n; j++) map[j] = j;
        /* a row may appear twice (lower and upper); the second copy is appended at the end */
        int *extra = (int *)malloc((size_t)(na + 1) * sizeof(int)); int nextra = 0;
        for (int k = 0; k < na; k++) {
            if (map[n + rows[k]] == -1) map[n + rows[k]] = n + k; else extra[nextra++] = n + k;
        }
        int c = 0;
        for (int k = 0; k < n + m; k++) { int o = w->F->perm[k]; if (map[o] >= 0) perm[c++] = map[o]; }
        for (int k = 0; k < nextra; k++) perm[c++] = extra[k];
        free(map); free(extra);
    }
    ldl_t *F = ldl_analyze(K, perm);
    free(perm);
    int rc = ldl_factor(F, K);
    double *b = (double *)malloc((size_t)N * sizeof(double));
    double *sol = (double *)malloc((size_t)N * sizeof(double));
    double *rhs = (double *)malloc((size_t)N * sizeof(double));
    double *xp = (double *)malloc((size_t)n * sizeof(double));
    double *zp = (double *)malloc((size_t)(m + 1) * sizeof(double));
    double *yp = (double *)calloc((size_t)(m + 1), sizeof(double));
    int ok = 0;
    if (rc == 0) {
        for (int j = 0; j < n; j++) b[j] = -w->q[j];
        for (int k = 0; k < nlow; k++) b[n + k] = w->l[rows[k]];
        for (int k = 0; k < nupp; k++) b[n + nlow + k] = w->u[rows[nlow + k]];
        memcpy(sol, b, (size_t)N * sizeof(double));
        ldl_solve(F, sol);
        for (int it = 0; it < s->polish_refine_iter; it++) {
            /* rhs = b - [P Ared'; Ared 0] sol */
            sym_triu_matvec(w->P, sol, rhs);
            for (int j = 0; j < n; j++) rhs[j] = b[j] - rhs[j];
            for (int k = 0; k < na; k++) {
                int rr = rows[k]; double acc = 0, yk = sol[n + k];
                for (int p = w->At->p[rr]; p < w->At->p[rr + 1]; p++) {
                    acc += w->At->x[p] * sol[w->At->i[p]];
                    rhs[w->At->i[p]] -= w->At->x[p] * yk;
                }
                rhs[n + k] = b[n + k] - acc;
            }
            ldl_solve(F, rhs);
            for (int j = 0; j < N; j++) sol[j] += rhs[j];
        }
        memcpy(xp, sol, (size_t)n * sizeof(double));
        mat_vec(w->A, xp, zp);
        for (int k = 0; k < nupp; k++) yp[rows[nlow + k]] = sol[n + nlow + k];
        for (int k = 0; k < nlow; k++) yp[rows[k]] = sol[n + k];    /* lower wins (as OSQP's if/else) */
        /* project (z, y) on the normal cone */
        for (int i = 0; i < m; i++) {
            double t = zp[i] + yp[i];
            double zc = t < w->l[i] ? w->l[i] : (t > w->u[i] ? w->u[i] : t);
            zp[i] = zc; yp[i] = t - zc;
        }
        resid_t rp;
        compute_residuals(w, xp, zp, yp, &rp);
        ok = (rp.pri < r_admm->pri && rp.dua < r_admm->dua) ||
             (rp.pri < r_admm->pri && r_admm->dua < 1e-10) ||
             (rp.dua < r_admm->dua && r_admm->pri < 1e-10);
        if (ok) {
            memcpy(w->x, xp, (size_t)n * sizeof(double));
            memcpy(w->z, zp, (size_t)m * sizeof(double));
            memcpy(w->y, yp, (size_t)m * sizeof(double));
            info->pri_res = rp.pri; info->dua_res = rp.dua;
        }
    }
    free(b); free(sol); free(rhs); free(xp); free(zp); free(yp); free(rows);
    ldl_free(F); csc_free(K);
    return ok ? 1 : -1;
}

int oq_solve(const csc_t *P, const double *q, const csc_t *A, const double *l, const double *u,
             const oq_settings *s, oq_cache *cache, const double *warm_x, const double *warm_y,
             double *x_out, double *y_out, oq_info *info, unsigned char *act_lo, unsigned char *act_up)
{
    int n = P->n, m = A->m;
    work_t W; memset(&W, 0, sizeof(W));
    work_t *w = &W;
    w->n = n; w->m = m; w->s = s;
    w->P = csc_copy(P); w->A = csc_copy(A);
#define DV(len) ((double *)calloc((size_t)((len) > 0 ? (len) : 1), sizeof(double)))
    w->q = DV(n); w->l = DV(m); w->u = DV(m);
    memcpy(w->q, q, (size_t)n * sizeof(double));
    memcpy(w->l, l, (size_t)m * sizeof(double));
    memcpy(w->u, u, (size_t)m * sizeof(double));
    w->D = DV(n); w->Dinv = DV(n); w->E = DV(m); w->Einv = DV(m);
    for (int j = 0; j < n; j++) w->D[j] = w->Dinv[j] = 1.0;
    for (int i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1.0;
    w->c = w->cinv = 1.0;
    w->ctype = (int *)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
    w->rho_vec = DV(m); w->rho_inv = DV(m);
    w->x = DV(n); w->z = DV(m); w->y = DV(m); w->x_prev = DV(n); w->z_prev = DV(m);
    w->xz = DV(n + m); w->delta_x = DV(n); w->delta_y = DV(m);
    w->Ax = DV(m); w->Px = DV(n); w->Aty = DV(n); w->tn = DV(n); w->tm = DV(m);

    if (s->scaling) ruiz_scale(w);
    w->At = csc_transpose(w->A);
    w->rho = s->rho < RHO_MIN ? RHO_MIN : (s->rho > RHO_MAX ? RHO_MAX : s->rho);
    set_rho_vec(w, 1);
    build_kkt(w);
    {
        uint64_t key = 0xcbf29ce484222325ull;
        key = hash_ints(key, w->K->p, n + m + 1);
        key = hash_ints(key, w->K->i, w->K->p[n + m]);
        if (cache && cache->perm && cache->key == key && cache->n == n + m) {
            w->F = ldl_analyze(w->K, cache->perm);
        } else {
            w->F = ldl_analyze(w->K, NULL);
            if (cache) {
                free(cache->perm);
                cache->perm = (int *)malloc((size_t)(n + m) * sizeof(int));
                memcpy(cache->perm, w->F->perm, (size_t)(n + m) * sizeof(int));
                cache->key = key; cache->n = n + m;
            }
        }
    }
    int rc = ldl_factor(w->F, w->K);
    memset(info, 0, sizeof(*info));
    int status = OQ_UNSOLVED;
    if (rc != 0) { status = OQ_NON_CVX; goto done; }

    if (s->warm_start && warm_x && warm_y) {
        for (int j = 0; j < n; j++) w->x[j] = w->Dinv[j] * warm_x[j];
        mat_vec(w->A, w->x, w->z);
        for (int i = 0; i < m; i++) w->y[i] = w->c * w->Einv[i] * warm_y[i];
    }

    resid_t r; memset(&r, 0, sizeof(r));
    int iter = 0, terminated = 0, can_check = 0;
    for (iter = 1; iter <= s->max_iter; iter++) {
        memcpy(w->x_prev, w->x, (size_t)n * sizeof(double));
        memcpy(w->z_prev, w->z, (size_t)m * sizeof(double));
        for (int j = 0; j < n; j++) w->xz[j] = s->sigma * w->x_prev[j] - w->q[j];
        for (int i = 0; i < m; i++) w->xz[n + i] = w->z_prev[i] - w->rho_inv[i] * w->y[i];
        ldl_solve(w->F, w->xz);
        for (int i = 0; i < m; i++) w->xz[n + i] = w->z_prev[i] + w->rho_inv[i] * (w->xz[n + i] - w->y[i]);
        for (int j = 0; j < n; j++) {
            w->x[j] = s->alpha * w->xz[j] + (1.0 - s->alpha) * w->x_prev[j];
            w->delta_x[j] = w->x[j] - w->x_prev[j];
        }
        for (int i = 0; i < m; i++) {
            double zr = s->alpha * w->xz[n + i] + (1.0 - s->alpha) * w->z_prev[i];
            double t = zr + w->rho_inv[i] * w->y[i];
            double zc = t < w->l[i] ? w->l[i] : (t > w->u[i] ? w->u[i] : t);
            w->z[i] = zc;
            w->delta_y[i] = w->rho_vec[i] * (zr - zc);
            w->y[i] += w->delta_y[i];
        }
        can_check = s->check_termination && (iter % s->check_termination == 0);
        if (can_check && check_termination(w, 0, &status, &r)) { terminated = 1; break; }
        if (s->adaptive_rho && s->adaptive_rho_interval && (iter % s->adaptive_rho_interval == 0)) {
            double rn = rho_estimate(w);
            if (rn > w->rho * s->adaptive_rho_tolerance || rn < w->rho / s->adaptive_rho_tolerance) {
                w->rho = rn;
                set_rho_vec(w, 0);
                update_kkt_rho(w);
                if (ldl_factor(w->F, w->K) != 0) { status = OQ_NON_CVX; terminated = 1; break; }
                info->rho_updates++;
            }
        }
    }
    if (!terminated) {
        iter = s->max_iter;
        if (!can_check) check_termination(w, 0, &status, &r);
        if (status == OQ_UNSOLVED) {
            if (!check_termination(w, 1, &status, &r)) status = OQ_MAX_ITER_REACHED;
        }
    }
    info->iters = iter;
    info->pri_res = r.pri; info->dua_res = r.dua;
    if (s->polish && status == OQ_SOLVED) info->polished = polish(w, info, &r, act_lo, act_up);
done:
    info->status = status;
    info->rho = w->rho;
    {
        int has_sol = !(status == OQ_PRIMAL_INFEASIBLE || status == OQ_PRIMAL_INFEASIBLE_INACCURATE ||
                        status == OQ_DUAL_INFEASIBLE || status == OQ_DUAL_INFEASIBLE_INACCURATE ||
                        status == OQ_NON_CVX);
        if (has_sol) {
            for (int j = 0; j < n; j++) x_out[j] = w->D[j] * w->x[j];
            for (int i = 0; i < m; i++) y_out[i] = w->cinv * w->E[i] * w->y[i];
            sym_triu_matvec(P, x_out, w->tn);
            double o = 0; for (int j = 0; j < n; j++) o += 0.5 * x_out[j] * w->tn[j] + q[j] * x_out[j];
            info->obj = o;
        } else {
            for (int j = 0; j < n; j++) x_out[j] = NAN;
            for (int i = 0; i < m; i++) y_out[i] = NAN;
            info->obj = (status == OQ_DUAL_INFEASIBLE || status == OQ_DUAL_INFEASIBLE_INACCURATE) ? -OQ_INFTY : OQ_INFTY;
        }
    }
    csc_free(w->P); csc_free(w->A); csc_free(w->At); csc_free(w->K); free(w->Kdiag); ldl_free(w->F);
    free(w->q); free(w->l); free(w->u); free(w->D); free(w->Dinv); free(w->E); free(w->Einv);
    free(w->ctype); free(w->rho_vec); free(w->rho_inv);
    free(w->x); free(w->z); free(w->y); free(w->x_prev); free(w->z_prev); free(w->xz);
    free(w->delta_x); free(w->delta_y); free(w->Ax); free(w->Px); free(w->Aty); free(w->tn); free(w->tm);
    return 0;
}
