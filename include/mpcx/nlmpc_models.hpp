// Device functors for the NLMPC kernels: the model zoo.
//
// libmpc++ takes the system, cost and constraint hooks as host closures (reference include/mpc/IDimensionable.hpp:94-149,
// set through NLMPC::setStateSpaceFunction / setObjectiveFunction / setIneqConFunction, NLMPC.hpp:139-280).  A kernel
// cannot call those, so a model here is a struct of static device functions with the same argument meaning:
//
//   NX, NU                      state / input dimensions                          (template arguments of mpc::NLMPC<>)
//   CONTINUOUS                  true: f is dx/dt and the transcription is trapezoidal collocation with step Ts
//                               (setDiscretizationSamplingTime, NLMPC.hpp:80-90); false: f is x(k+1)
//   nineq(ph)                   number of user inequalities                       (Tineq)
//   f(out, x, u, params)        StateFunHandle: vector field                      (IDimensionable.hpp:130)
//   cost(X, U, e, ph, params)   ObjFunHandle: X(i, j), U(i, j) are rows i = 0..ph of the (ph+1) x n matrices the
//                               reference passes (row 0 = x0, U row ph = copy of row ph-1), e the slack
//   ineq(k, X, U, e, ph, p)     component k of IConFunHandle's output vector, g_k <= 0
//   COST_STAGEWISE, stage(i, X, U, ph, p)
//                               optional: the cost is a sum over the rows i = 0..ph of X and U (plus a term in the slack alone) and stage(i)
//                               is row i's share.  A finite difference of the cost in an entry of row i is then the difference of that
//                               row's share -- the same quotient as (f(x + d e) - f(x)) / d of Objective.hpp:198-265 with the rows that
//                               cancel left out (ph (nx + nu) evaluations of one row instead of the whole horizon, and without the
//                               cancellation of two sums of ph rows); slack_cost(e, p) is the term in the slack alone.  The workgroup form
//                               also sums the rows' shares over its lanes for the cost itself (another order of the same additions)
//   neq_user(ph), eq(k, X, U, ph, p)
//                               EConFunHandle: component k of the user equalities h_k = 0 (defaults: none, NoUserEq)
//   ineq_reads_x/u(k, i), ineq_rows_of_x/u(i, first, count), INEQ_USES_SLACK, INEQ_U_ROWS_DISJOINT
//                               structure of that vector, both ways round: which rows of X / U constraint k reads, and
//                               which (contiguous) constraints read row i.  Everything else differentiates to an exact
//                               zero (also in the reference's finite differences) and is skipped; the second form lets
//                               every lane walk its own few rows instead of the wavefront walking the union of them.
// X and U are accessor objects (a perturbed or a shifted view of the trajectory in LDS), hence the templates.  This is the
// fast form: the declared structure lets the engine skip every structural zero.  Hooks with the reference's own
// signatures (whole constraint vectors, Eigen-style matrices) go through mpcx/nlmpc_hooks.hpp instead -- compiled in the
// user's translation unit or at run time from source -- and need no change to this file.
#pragma once

#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif

namespace mpcx {
namespace models {

// (ph+1) x n matrix in LDS, row-major, with up to two perturbed elements of one column
struct Pert {
    const double *M;
    int n, r1, r2, c;
    double d;
    __device__ __forceinline__ double operator()(int i, int j) const
    {
        const double v = M[i * n + j];
        return (j == c && (i == r1 || i == r2)) ? v + d : v;
    }
};
// M + a * D: a trial point of the line search
struct Lin {
    const double *M, *D;
    int n;
    double a;
    __device__ __forceinline__ double operator()(int i, int j) const { return M[i * n + j] + a * D[i * n + j]; }
};

// defaults for systems without user equalities (EConFunHandle, IDimensionable.hpp:118-122 / NLMPC::setEqConFunction)
// OutFunHandle (IDimensionable.hpp:138-143, NLMPC::setOutputFunction): y = out(x, u) feeds OptSequence::output.  Without an
// output function the reference's outputs are zeros (Model.hpp:72-96); the zoo's cost and constraint functors read X
// and U directly, as the reference examples' do.
struct NoOutput {
    static constexpr bool HAS_OUTPUT = false;
    __device__ static void out(double *, const double *, const double *, const double *) {}
};

struct NoUserEq {
    static constexpr bool VECTOR_HOOKS = false;        // component-wise constraint functors with declared structure (this file)
    __host__ __device__ static int neq_user(int) { return 0; }
    template <class XA, class UA>
    __device__ static double eq(int, const XA &, const UA &, int, const double *) { return 0.0; }
    __host__ __device__ static bool eq_reads_x(int, int) { return true; }       // dense unless the model says otherwise
    __host__ __device__ static bool eq_reads_u(int, int) { return true; }
    static constexpr bool INEQ_U_ROWS_DISJOINT = false;                 // true: ineq_rows_of_u(i) and ineq_rows_of_u(i') share no row for i != i'
    static constexpr bool COST_STAGEWISE = false;                       // true: stage(i, X, U, ph, p) is row i's share of the cost (see the header)
    template <class XA, class UA>
    __device__ static double stage(int, const XA &, const UA &, int, const double *) { return 0.0; }
    __device__ static double slack_cost(double, const double *) { return 0.0; }      // the cost's term in the slack variable alone
    static constexpr bool XFREE_ROWS_SPARSE = false;                    // true: a user row that reads no state has at most kNlSparse non-zero entries in the
                                                                        // move-blocked inputs (+ slack); the workgroup form then keeps it as an (index, value) list
    static constexpr bool SPARSE_ROWS_ONE_ENTRY = false;                // true: every short-list row (XFREE_ROWS_SPARSE) has ONE entry.  Two such rows on one variable are
                                                                        // parallel and the dual method never holds two parallel rows, so at most one working row touches a
                                                                        // variable: the workgroup form then scatters N_W' r by LDS atomic adds that cannot meet; without the
                                                                        // promise it sums every variable's contributions in the working set's order (the same bits every run)
    static constexpr int CURV0_AFTER = 0;                               // the workgroup form sets its curvature estimate to the condensed Gauss-Newton Hessian of the cost before
                                                                        // iteration CURV0_AFTER (0: the solve starts from it).  A model whose constraints make the problem non-convex
                                                                        // in a way that matters -- which side an obstacle is passed on -- lets the first iterations run from the
                                                                        // identity, as NLopt's SLSQP does: they decide the local optimum the solve ends at
    static constexpr bool XFREE_ROWS_AFFINE = false;                    // true: those rows are affine in the inputs -- the same Jacobian at every iterate (the workgroup
                                                                        // form then carries the inverse of the working set's Schur complement from one sub-problem to the next)
};

// ---- model zoo ----------------------------------------------------------------------------------
struct VanDerPol : NoUserEq, NoOutput {      // reference examples/vanderpol_ex.cpp:33-65
    static constexpr int NX = 2, NU = 1, NY = 2;
    static constexpr int NPARAMS = 1;                              // (none used; the library keeps one)
    static constexpr bool CONTINUOUS = true;
    __host__ __device__ static int nineq(int ph) { return ph + 1; }
    __device__ static void f(double *dx, const double *x, const double *u, const double *)
    {
        dx[0] = ((1.0 - (x[1] * x[1])) * x[0]) - x[1] + u[0];
        dx[1] = x[0];
    }
    template <class XA, class UA>
    __device__ static double cost(const XA &X, const UA &U, double, int ph, const double *)
    {
        // x.array().square().sum() + u.array().square().sum() with the coefficients in storage (column-major) order: the
        // same arithmetic, term for term, as the reference example's lambda evaluated through mpcx/matrix.hpp
        double sx = 0, su = 0;
        for (int j = 0; j < NX; ++j)
#pragma unroll 8
            for (int i = 0; i <= ph; ++i) sx += X(i, j) * X(i, j);
#pragma unroll 8
        for (int i = 0; i <= ph; ++i) su += U(i, 0) * U(i, 0);
        return sx + su;
    }
    static constexpr bool COST_STAGEWISE = true;
    template <class XA, class UA>
    __device__ static double stage(int i, const XA &X, const UA &U, int, const double *)
    {
        return (X(i, 0) * X(i, 0) + X(i, 1) * X(i, 1)) + U(i, 0) * U(i, 0);
    }
    template <class XA, class UA>
    __device__ static double ineq(int k, const XA &, const UA &U, double, int, const double *) { return U(k, 0) - 0.5; }
    // structure of the inequality Jacobian: which rows of X / U constraint k reads (anything else differentiates to an exact 0)
    static constexpr bool INEQ_USES_SLACK = false;
    static constexpr bool INEQ_U_ROWS_DISJOINT = true;
    static constexpr bool XFREE_ROWS_SPARSE = true;
    static constexpr bool XFREE_ROWS_AFFINE = true;                     // u <= 0.5
    static constexpr bool SPARSE_ROWS_ONE_ENTRY = true;
    __host__ __device__ static bool ineq_reads_x(int, int) { return false; }
    __host__ __device__ static bool ineq_reads_u(int k, int i) { return k == i; }
    __host__ __device__ static void ineq_rows_of_x(int, int &first, int &count) { first = 0; count = 0; }
    __host__ __device__ static void ineq_rows_of_u(int i, int &first, int &count) { first = i; count = 1; }
};

// The same system with the terminal equality x(ph) = 0: the textbook use of NLMPC::setEqConFunction (NLMPC.hpp:246-262).
// No reference example sets user equalities; this model exists so that the path is exercised.
struct VanDerPolTerminal : VanDerPol {
    __host__ __device__ static int neq_user(int) { return 2; }
    template <class XA, class UA>
    __device__ static double eq(int k, const XA &X, const UA &, int ph, const double *) { return X(ph, k); }
};

// The same system with a rate limit on the input, |u_i - u_{i-1}| <= params[0], next to u_i <= 0.5: short-list rows with TWO entries, and up to five
// rows that touch one input (its own bound, the two rate rows of its step and of the next) -- what the reference's examples do not have and the
// sub-problem's (index, value) lists are built for.  Rows are grouped by step, [u_i <= 0.5 | u_i - u_{i-1} <= r | u_{i-1} - u_i <= r] (the rate
// rows of step 0 compare u_0 with itself: constant -r), so that the rows reading U row i are the contiguous range ineq_rows_of_u wants.
// (A limit on the SECOND difference -- three entries, three working rows on one input at most optima -- was tried as the test vehicle and dropped:
// with a = 0.02 the problem is degenerate enough that scipy's SLSQP gives up on two starts of three and the kernel's line search stalls on some.)
struct VanDerPolRate : VanDerPol {
    __host__ __device__ static int nineq(int ph) { return 3 * (ph + 1); }
    template <class XA, class UA>
    __device__ static double ineq(int k, const XA &, const UA &U, double, int, const double *p)
    {
        const int i = k / 3, t = k - 3 * i, im = i > 0 ? i - 1 : 0;
        if (t == 0) return U(i, 0) - 0.5;
        const double du = U(i, 0) - U(im, 0);
        return (t == 1 ? du : -du) - p[0];
    }
    static constexpr bool INEQ_U_ROWS_DISJOINT = false;
    static constexpr bool SPARSE_ROWS_ONE_ENTRY = false;
    __host__ __device__ static bool ineq_reads_u(int k, int i) { const int s = k / 3, t = k - 3 * s; return i == s || (t > 0 && s > 0 && i == s - 1); }
    __host__ __device__ static void ineq_rows_of_u(int i, int &first, int &count) { first = 3 * i; count = 6; }      // (rows 3 i .. 3 i + 5; asked for i < ph only: within nineq)
};

struct Ugv : NoUserEq {            // reference examples/ugv_ex.cpp:32-124 (zero-order hold of a planar double integrator)
    static constexpr int NX = 4, NU = 2, NY = 4;
    static constexpr int NPARAMS = 9;
    static constexpr bool CONTINUOUS = false;
    static constexpr bool HAS_OUTPUT = true;                       // y = Cd x + Dd u with C = I, D = 0 (ugv_ex.cpp:34-77)
    __device__ static void out(double *y, const double *x, const double *, const double *) { for (int a = 0; a < 4; ++a) y[a] = x[a]; }
    // params: [0..1] v_pref, [2..4] obstacle 0 (x, y, r), [5..7] obstacle 1, [8] Ts
    __host__ __device__ static int nineq(int ph) { return 2 * (ph + 1); }
    __device__ static void f(double *xn, const double *x, const double *u, const double *p)
    {
        const double Ts = p[8];
        xn[0] = x[0] + Ts * x[2] + 0.5 * Ts * Ts * u[0];
        xn[1] = x[1] + Ts * x[3] + 0.5 * Ts * Ts * u[1];
        xn[2] = x[2] + Ts * u[0];
        xn[3] = x[3] + Ts * u[1];
    }
    template <class XA, class UA>
    __device__ static double cost(const XA &X, const UA &U, double e, int ph, const double *p)
    {
        double s = 0;
        const double vx = p[0], vy = p[1];
#pragma unroll 8
        for (int i = 0; i <= ph; ++i) {
            const double a = X(i, 2) - vx, b = X(i, 3) - vy;
            s += 1e3 * (a * a + b * b);
            s += 1e-2 * (U(i, 0) * U(i, 0) + U(i, 1) * U(i, 1));
        }
        return s + 1e-5 * e * e;
    }
    static constexpr bool COST_STAGEWISE = true;
    template <class XA, class UA>
    __device__ static double stage(int i, const XA &X, const UA &U, int, const double *p)
    {
        const double a = X(i, 2) - p[0], b = X(i, 3) - p[1];
        return 1e3 * (a * a + b * b) + 1e-2 * (U(i, 0) * U(i, 0) + U(i, 1) * U(i, 1));
    }
    __device__ static double slack_cost(double e, const double *) { return 1e-5 * e * e; }
    template <class XA, class UA>
    __device__ static double ineq(int k, const XA &X, const UA &, double, int, const double *p)
    {
        const int i = k >> 1, o = k & 1;
        const double dx = X(i, 0) - p[2 + 3 * o], dy = X(i, 1) - p[3 + 3 * o];
        return p[4 + 3 * o] - sqrt(dx * dx + dy * dy);
    }
    static constexpr bool INEQ_USES_SLACK = false;
    static constexpr bool INEQ_U_ROWS_DISJOINT = true;
    // (measured on the golden set and on config 3's batch, tools/ugv_curv_sweep.py -> profiles/r06_ugv_curv_sweep.txt: installed at the start the
    // Gauss-Newton matrix sends 29 of 231 golden instances to a worse local optimum than the oracle's; after 10 iterations from the identity 228 of 231
    // agree with the oracle -- 224 when it is never installed -- and 97.9 % of the 4096 end where the identity's route ends, 81 of the others lower, 6 higher)
    static constexpr int CURV0_AFTER = 10;
    __host__ __device__ static bool ineq_reads_x(int k, int i) { return (k >> 1) == i; }
    __host__ __device__ static bool ineq_reads_u(int, int) { return false; }
    __host__ __device__ static void ineq_rows_of_x(int i, int &first, int &count) { first = 2 * i; count = 2; }
    __host__ __device__ static void ineq_rows_of_u(int, int &first, int &count) { first = 0; count = 0; }
};

template <int N>
struct Oscillators : NoUserEq, NoOutput {    // reference examples/networked_oscillators_ex.cpp:17-76; params: [mu, k]
    static constexpr int NX = 2 * N, NU = N, NY = 2 * N;
    static constexpr int NPARAMS = 2;
    static constexpr bool CONTINUOUS = true;
    __host__ __device__ static int nineq(int ph) { return (ph + 1) * N; }
    __device__ static void f(double *dx, const double *x, const double *u, const double *p)
    {
        const double mu = p[0], k = p[1];
        // (unrolled: with run-time subscripts the callers' copies of x and dx would live in scratch memory, a trip to HBM per element)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            dx[2 * i] = x[2 * i + 1];
            double a = mu * (1 - x[2 * i] * x[2 * i]) * x[2 * i + 1] - x[2 * i] + u[i];
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (i != j) a += k * (x[2 * j] - x[2 * i]);
            dx[2 * i + 1] = a;
        }
    }
    template <class XA, class UA>
    __device__ static double cost(const XA &X, const UA &U, double, int ph, const double *)
    {
        double sx = 0, su = 0;                       // column-major order, as for VanDerPol above
        for (int j = 0; j < NX; ++j)
#pragma unroll 8
            for (int i = 0; i <= ph; ++i) sx += X(i, j) * X(i, j);
        for (int j = 0; j < NU; ++j)
#pragma unroll 8
            for (int i = 0; i <= ph; ++i) su += U(i, j) * U(i, j);
        return sx + su;
    }
    static constexpr bool COST_STAGEWISE = true;
    template <class XA, class UA>
    __device__ static double stage(int i, const XA &X, const UA &U, int, const double *)
    {
        double sx = 0, su = 0;
#pragma unroll
        for (int j = 0; j < NX; ++j) sx += X(i, j) * X(i, j);
#pragma unroll
        for (int j = 0; j < NU; ++j) su += U(i, j) * U(i, j);
        return sx + su;
    }
    template <class XA, class UA>
    __device__ static double ineq(int k, const XA &, const UA &U, double, int, const double *) { return U(k / N, k % N) - 0.5; }
    static constexpr bool INEQ_USES_SLACK = false;
    static constexpr bool INEQ_U_ROWS_DISJOINT = true;
    static constexpr bool XFREE_ROWS_SPARSE = true;
    static constexpr bool XFREE_ROWS_AFFINE = true;                     // u_j <= 0.5
    static constexpr bool SPARSE_ROWS_ONE_ENTRY = true;
    __host__ __device__ static bool ineq_reads_x(int, int) { return false; }
    __host__ __device__ static bool ineq_reads_u(int k, int i) { return k / N == i; }
    __host__ __device__ static void ineq_rows_of_x(int, int &first, int &count) { first = 0; count = 0; }
    __host__ __device__ static void ineq_rows_of_u(int i, int &first, int &count) { first = i * N; count = N; }
};

}  // namespace models
}  // namespace mpcx
