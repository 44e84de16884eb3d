/* TEST / BENCHMARK INFRASTRUCTURE ONLY -- plain-C restatement of what libmpc++ hands NLopt on every SLSQP evaluation, for the three
 * example systems (examples/vanderpol_ex.cpp, ugv_ex.cpp, networked_oscillators_ex.cpp):
 *   Mapping::unwrapVector        include/mpc/NLMPC/Mapping.hpp:174-211
 *   Objective::evaluate / computeGradient (forward differences, step from Xa.array()(j))        Objective.hpp:91-265
 *   Constraints::getStateEqConstraints + computeStateEqJacobian (central differences)           Constraints.hpp:490-628, 844-905
 *   Constraints::evaluateIneq + computeIneqJacobian (central differences)                       Constraints.hpp:211-316, 641-721
 * Same formulas as oracle/nlmpc_numpy.py (tests/test_nlmpc_oracle.py pins the two against each other and against the reference's
 * component known answers); compiled, because bench.py's NLMPC `cpu_baseline` should cost what NLOptimizer::run costs in C++ --
 * the callbacks are where that time goes -- not what numpy loops cost.  Identity mapping scalings.  Never linked into the product. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int model;             /* 0 Van der Pol, 1 UGV (double integrator + two obstacles), 2 coupled oscillators */
    int nx, nu, ph, ch, nineq, N;
    double Ts, mu, k, vpx, vpy;
    double obs[6];         /* (x, y, r) x 2 */
} nl_model;

static const double DV = 1.4901161193847656e-08;      /* sqrt(DBL_EPSILON), Objective.hpp:283 */

static void nl_f(const nl_model *m, const double *x, const double *u, double *o)
{
    if (m->model == 0) { o[0] = (1.0 - x[1] * x[1]) * x[0] - x[1] + u[0]; o[1] = x[0]; }
    else if (m->model == 1) {
        const double T = 0.1;
        o[0] = x[0] + T * x[2] + 0.5 * T * T * u[0]; o[1] = x[1] + T * x[3] + 0.5 * T * T * u[1];
        o[2] = x[2] + T * u[0]; o[3] = x[3] + T * u[1];
    } else {
        const int N = m->N;
        for (int i = 0; i < N; i++) {
            double a = m->mu * (1 - x[2 * i] * x[2 * i]) * x[2 * i + 1] - x[2 * i] + u[i];
            for (int j = 0; j < N; j++) if (j != i) a += m->k * (x[2 * j] - x[2 * i]);
            o[2 * i] = x[2 * i + 1]; o[2 * i + 1] = a;
        }
    }
}

/* X [(ph+1) x nx], U [(ph+1) x nu], row-major */
static double nl_cost(const nl_model *m, const double *X, const double *U, double e)
{
    const int n1 = m->ph + 1;
    double s = 0;
    if (m->model == 1) {
        double a = 0, b = 0;
        for (int i = 0; i < n1; i++) {
            const double dx = X[i * 4 + 2] - m->vpx, dy = X[i * 4 + 3] - m->vpy;
            a += dx * dx + dy * dy;
            b += U[i * 2] * U[i * 2] + U[i * 2 + 1] * U[i * 2 + 1];
        }
        return 1e3 * a + 1e-2 * b + 1e-5 * e * e;
    }
    for (int k = 0; k < n1 * m->nx; k++) s += X[k] * X[k];
    for (int k = 0; k < n1 * m->nu; k++) s += U[k] * U[k];
    return s;
}

static void nl_ineq(const nl_model *m, const double *X, const double *U, double e, double *g)
{
    const int n1 = m->ph + 1;
    (void)e;
    if (m->model == 1) {
        for (int i = 0; i < n1; i++)
            for (int k = 0; k < 2; k++) {
                const double dx = X[i * 4] - m->obs[3 * k], dy = X[i * 4 + 1] - m->obs[3 * k + 1];
                g[i * 2 + k] = m->obs[3 * k + 2] - sqrt(dx * dx + dy * dy);
            }
    } else if (m->model == 0) {
        for (int i = 0; i < n1; i++) g[i] = U[i] - 0.5;
    } else {
        for (int k = 0; k < n1 * m->nu; k++) g[k] = U[k] - 0.5;
    }
}

static void unwrap(const nl_model *m, const double *z, const double *x0, double *X, double *U, double *e)
{
    const int nx = m->nx, nu = m->nu, ph = m->ph, ch = m->ch;
    memcpy(X, x0, sizeof(double) * nx);
    memcpy(X + nx, z, sizeof(double) * ph * nx);
    for (int i = 0; i <= ph; i++) {
        int b = i < ph - 1 ? i : ph - 1;
        if (b > ch - 1) b = ch - 1;
        memcpy(U + i * nu, z + ph * nx + b * nu, sizeof(double) * nu);
    }
    *e = z[ph * nx + ch * nu];
}

/* Eigen's .array()(j) on a (ph+1) x n column-major matrix whose entry (r, c) we hold at M[r * n + c] */
static double lin_abs1(const double *M, int n, int n1, int j)
{
    const double v = fabs(M[(j % n1) * n + j / n1]);
    return v > 1.0 ? v : 1.0;
}

int nlc_nz(const nl_model *m) { return m->ph * m->nx + m->ch * m->nu + 1; }

double nlc_objective(const nl_model *m, const double *z, const double *x0, double *grad)
{
    const int nx = m->nx, nu = m->nu, ph = m->ph, ch = m->ch, n1 = ph + 1, nz = nlc_nz(m);
    double *X = malloc(sizeof(double) * n1 * (nx + nu)), *U = X + n1 * nx, e;
    unwrap(m, z, x0, X, U, &e);
    const double f0 = nl_cost(m, X, U, e);
    if (grad) {
        memset(grad, 0, sizeof(double) * nz);
        for (int i = 0; i < ph; i++)
            for (int j = 0; j < nx; j++) {
                const double dx = DV * lin_abs1(X, nx, n1, j), keep = X[(i + 1) * nx + j];
                X[(i + 1) * nx + j] = keep + dx;
                grad[i * nx + j] = (nl_cost(m, X, U, e) - f0) / dx;
                X[(i + 1) * nx + j] = keep;
            }
        for (int i = 0; i < ph; i++)
            for (int j = 0; j < nu; j++) {
                const double du = DV * lin_abs1(U, nu, n1, j), k0 = U[i * nu + j], k1 = U[ph * nu + j];
                U[i * nu + j] = k0 + du;
                if (i == ph - 1) U[ph * nu + j] = k1 + du;          /* the last row moves with its copy */
                const double d = (nl_cost(m, X, U, e) - f0) / du;
                U[i * nu + j] = k0; U[ph * nu + j] = k1;
                int b = i < ch - 1 ? i : ch - 1;
                grad[ph * nx + b * nu + j] += d;                     /* Iz2u' vec(Jmv) */
            }
        const double de = fmax(DV, fabs(e)) * DV;
        grad[nz - 1] = (nl_cost(m, X, U, e + de) - nl_cost(m, X, U, e - de)) / (2 * de);
    }
    free(X);
    return f0;
}

static void state_jac(const nl_model *m, const double *x, const double *u, double *A, double *B)
{
    const int nx = m->nx, nu = m->nu;
    double xp[64], xm[64], up[32], um[32], fp[64], fm[64];
    for (int i = 0; i < nx; i++) {
        const double dx = DV * fmax(fabs(x[i]), 1.0);
        memcpy(xp, x, sizeof(double) * nx); memcpy(xm, x, sizeof(double) * nx);
        xp[i] += dx; xm[i] -= dx;
        nl_f(m, xp, u, fp); nl_f(m, xm, u, fm);
        for (int a = 0; a < nx; a++) A[a * nx + i] = (fp[a] - fm[a]) / (2 * dx);
    }
    for (int i = 0; i < nu; i++) {
        const double du = DV * fmax(fabs(u[i]), 1.0);
        memcpy(up, u, sizeof(double) * nu); memcpy(um, u, sizeof(double) * nu);
        up[i] += du; um[i] -= du;
        nl_f(m, x, up, fp); nl_f(m, x, um, fm);
        for (int a = 0; a < nx; a++) B[a * nu + i] = (fp[a] - fm[a]) / (2 * du);
    }
}

/* c [ph nx]; J [(ph nx) x nz] row-major (dense, as the reference's Jacobian matrix), or NULL */
void nlc_state_eq(const nl_model *m, const double *z, const double *x0, double *c, double *J)
{
    const int nx = m->nx, nu = m->nu, ph = m->ph, ch = m->ch, n1 = ph + 1, nz = nlc_nz(m);
    const int continuous = m->model != 1;
    double *X = malloc(sizeof(double) * (n1 * (nx + nu) + 2 * nx * nx + 2 * nx * nu + 2 * nx)), *U = X + n1 * nx, e;
    double *Ak = U + n1 * nu, *Bk = Ak + nx * nx, *Ak1 = Bk + nx * nu, *Bk1 = Ak1 + nx * nx, *f0 = Bk1 + nx * nu, *f1 = f0 + nx;
    unwrap(m, z, x0, X, U, &e);
    if (J) memset(J, 0, sizeof(double) * (size_t)ph * nx * nz);
    for (int i = 0; i < ph; i++) {
        const double *xk = X + i * nx, *uk = U + i * nu, *xk1 = X + (i + 1) * nx;
        int b = i < ch - 1 ? i : ch - 1;
        if (continuous) {
            const double h = m->Ts / 2.0;
            nl_f(m, xk, uk, f0); nl_f(m, xk1, uk, f1);
            for (int a = 0; a < nx; a++) c[i * nx + a] = xk[a] + h * (f0[a] + f1[a]) - xk1[a];
            if (J) {
                state_jac(m, xk, uk, Ak, Bk); state_jac(m, xk1, uk, Ak1, Bk1);
                for (int a = 0; a < nx; a++) {
                    double *row = J + (size_t)(i * nx + a) * nz;
                    for (int q = 0; q < nx; q++) {
                        if (i > 0) row[(i - 1) * nx + q] = (a == q ? 1.0 : 0.0) + h * Ak[a * nx + q];
                        row[i * nx + q] = (a == q ? -1.0 : 0.0) + h * Ak1[a * nx + q];
                    }
                    for (int q = 0; q < nu; q++) row[ph * nx + b * nu + q] += h * (Bk[a * nu + q] + Bk1[a * nu + q]);
                }
            }
        } else {
            nl_f(m, xk, uk, f0);
            for (int a = 0; a < nx; a++) c[i * nx + a] = xk1[a] - f0[a];
            if (J) {
                state_jac(m, xk, uk, Ak, Bk);
                for (int a = 0; a < nx; a++) {
                    double *row = J + (size_t)(i * nx + a) * nz;
                    row[i * nx + a] = 1.0;
                    if (i > 0) for (int q = 0; q < nx; q++) row[(i - 1) * nx + q] = -Ak[a * nx + q];
                    for (int q = 0; q < nu; q++) row[ph * nx + b * nu + q] += -Bk[a * nu + q];
                }
            }
        }
    }
    free(X);
}

/* g [nineq]; J [nineq x nz] row-major, or NULL */
void nlc_user_ineq(const nl_model *m, const double *z, const double *x0, double *g, double *J)
{
    const int nx = m->nx, nu = m->nu, ph = m->ph, ch = m->ch, n1 = ph + 1, nz = nlc_nz(m), ni = m->nineq;
    double *X = malloc(sizeof(double) * (n1 * (nx + nu) + 2 * ni)), *U = X + n1 * nx, *fp = U + n1 * nu, *fm = fp + ni, e;
    unwrap(m, z, x0, X, U, &e);
    nl_ineq(m, X, U, e, g);
    if (J) {
        memset(J, 0, sizeof(double) * (size_t)ni * nz);
        for (int i = 0; i < ph; i++)
            for (int j = 0; j < nx; j++) {
                const double dx = DV * lin_abs1(X, nx, n1, j), keep = X[(i + 1) * nx + j];
                X[(i + 1) * nx + j] = keep + dx; nl_ineq(m, X, U, e, fp);
                X[(i + 1) * nx + j] = keep - dx; nl_ineq(m, X, U, e, fm);
                X[(i + 1) * nx + j] = keep;
                for (int r = 0; r < ni; r++) J[(size_t)r * nz + i * nx + j] = (fp[r] - fm[r]) / (2 * dx);
            }
        for (int i = 0; i < ph; i++)                     /* every input row on its own, no pairing of the last one (:684-706) */
            for (int j = 0; j < nu; j++) {
                const double du = DV * lin_abs1(U, nu, n1, j), keep = U[i * nu + j];
                U[i * nu + j] = keep + du; nl_ineq(m, X, U, e, fp);
                U[i * nu + j] = keep - du; nl_ineq(m, X, U, e, fm);
                U[i * nu + j] = keep;
                int b = i < ch - 1 ? i : ch - 1;
                for (int r = 0; r < ni; r++) J[(size_t)r * nz + ph * nx + b * nu + j] += (fp[r] - fm[r]) / (2 * du);
            }
        const double de = fmax(DV, fabs(e)) * DV;
        nl_ineq(m, X, U, e + de, fp); nl_ineq(m, X, U, e - de, fm);
        for (int r = 0; r < ni; r++) J[(size_t)r * nz + nz - 1] = (fp[r] - fm[r]) / (2 * de);
    }
    free(X);
}
