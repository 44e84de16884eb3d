// NLMPC transcription kernels for gfx950: what libmpc++ evaluates inside every NLopt SLSQP callback
// (reference include/mpc/NLMPC/NLOptimizer.hpp:760-997 -> Objective.hpp:91-265, Constraints.hpp:211-316,
// 490-905, Mapping.hpp:174-211), for a batch of decision vectors, one instance per wavefront:
//   - unwrap z into (X, U, slack) with move blocking                       (Mapping::unwrapVector)
//   - cost + forward-difference gradient, with the reference's step quirk  (Objective::computeGradient)
//   - dynamics equalities (trapezoidal collocation or one-step) + central-difference blocks A_i, B_i
//                                                                          (getStateEqConstraints)
//   - user inequalities + central-difference Jacobian                      (computeIneqJacobian)
// The user hooks of the reference are host std::function objects (IDimensionable.hpp:94-149) which a
// kernel cannot call; here they are device functors compiled into the library (a small model zoo with
// the reference's example systems), selected by id.  Every finite-difference column is an independent
// re-evaluation of a whole-horizon function: lanes own columns, the unwrapped trajectory sits in the
// wave's LDS slice and perturbations are applied on the fly by the accessor.
#include <hip/hip_runtime.h>

#include <cmath>

#include "nlmpc_device.hpp"

namespace mpcx {
namespace {

__device__ __forceinline__ void nl_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

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

// ---- model zoo ----------------------------------------------------------------------------------
struct VanDerPol {      // reference examples/vanderpol_ex.cpp:33-65
    static constexpr int NX = 2, NU = 1;
    static constexpr bool CONTINUOUS = true;
    __device__ static int nineq(int ph) { return ph + 1; }
    __device__ static void f(double *dx, const double *x, const double *u, const double *)
    {
        dx[0] = ((1.0 - (x[1] * x[1])) * x[0]) - x[1] + u[0];
        dx[1] = x[0];
    }
    __device__ static double cost(const Pert &X, const Pert &U, double, int ph, const double *)
    {
        double s = 0;
        for (int i = 0; i <= ph; ++i) { s += X(i, 0) * X(i, 0) + X(i, 1) * X(i, 1); s += U(i, 0) * U(i, 0); }
        return s;
    }
    __device__ static double ineq(int k, const Pert &, const Pert &U, double, int, const double *) { return U(k, 0) - 0.5; }
};

struct Ugv {            // reference examples/ugv_ex.cpp:32-124 (zero-order hold of a planar double integrator)
    static constexpr int NX = 4, NU = 2;
    static constexpr bool CONTINUOUS = false;
    // params: [0..1] v_pref, [2..4] obstacle 0 (x, y, r), [5..7] obstacle 1, [8] Ts
    __device__ static int nineq(int ph) { return 2 * (ph + 1); }
    __device__ static void f(double *xn, const double *x, const double *u, const double *p)
    {
        const double Ts = p[8];
        xn[0] = x[0] + Ts * x[2] + 0.5 * Ts * Ts * u[0];
        xn[1] = x[1] + Ts * x[3] + 0.5 * Ts * Ts * u[1];
        xn[2] = x[2] + Ts * u[0];
        xn[3] = x[3] + Ts * u[1];
    }
    __device__ static double cost(const Pert &X, const Pert &U, double e, int ph, const double *p)
    {
        double s = 0;
        for (int i = 0; i <= ph; ++i) {
            const double a = X(i, 2) - p[0], b = X(i, 3) - p[1];
            s += 1e3 * (a * a + b * b);
            s += 1e-2 * (U(i, 0) * U(i, 0) + U(i, 1) * U(i, 1));
        }
        return s + 1e-5 * e * e;
    }
    __device__ static double ineq(int k, const Pert &X, const Pert &, double, int, const double *p)
    {
        const int i = k >> 1, o = k & 1;
        const double dx = X(i, 0) - p[2 + 3 * o], dy = X(i, 1) - p[3 + 3 * o];
        return p[4 + 3 * o] - sqrt(dx * dx + dy * dy);
    }
};

template <class Mdl>
__global__ __launch_bounds__(256) void nlmpc_evaluate(const NlmpcDev M, const NlmpcBatchDev Bt)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int ph = M.ph, ch = M.ch, nz = M.nz, nineq = M.nineq;
    const double dv = 1.4901161193847656e-08;          // sqrt(DBL_EPSILON), Objective.hpp:283
    double *Xs = smem + (size_t)wave * M.lds_per_wave;  // (ph+1) x NX
    double *Us = Xs + (ph + 1) * NX;                    // (ph+1) x NU
    double *Jm = Us + (ph + 1) * NU;                    // ph x NU scratch (gradient wrt the input rows)
    const double *prm = M.params;

    for (int b = blockIdx.x * wpb + wave; b < Bt.batch; b += gridDim.x * wpb) {
        const double *z = Bt.z + (size_t)b * nz;
        // ---- Mapping::unwrapVector (scalings are 1 for the zoo models)
        for (int k = lane; k < (ph + 1) * NX; k += 64) {
            const int i = k / NX, j = k - i * NX;
            Xs[k] = i == 0 ? Bt.x0[(size_t)b * NX + j] : z[(i - 1) * NX + j];
        }
        for (int k = lane; k < (ph + 1) * NU; k += 64) {
            const int i = k / NU, j = k - i * NU;
            const int blk = min(min(i, ph - 1), ch - 1);          // first ch-1 moves one step each, the last one held
            Us[k] = z[ph * NX + blk * NU + j];
        }
        nl_wave_sync();
        const double e = z[nz - 1];
        auto Xa = [&](int j) { const double v = fabs(Xs[(j % (ph + 1)) * NX + j / (ph + 1)]); return v > 1.0 ? v : 1.0; };
        auto Ua = [&](int j) { const double v = fabs(Us[(j % (ph + 1)) * NU + j / (ph + 1)]); return v > 1.0 ? v : 1.0; };
        const Pert X0{Xs, NX, -1, -1, -1, 0.0}, U0{Us, NU, -1, -1, -1, 0.0};

        // ---- Objective::evaluate + computeGradient
        if (Bt.cost || Bt.grad) {
            const double f0 = Mdl::cost(X0, U0, e, ph, prm);
            if (lane == 0 && Bt.cost) Bt.cost[b] = f0;
            if (Bt.grad) {
                double *g = Bt.grad + (size_t)b * nz;
                for (int k = lane; k < ph * NX; k += 64) {
                    const int i = k / NX, j = k - i * NX;
                    const double dx = dv * Xa(j);
                    const Pert Xp{Xs, NX, i + 1, -1, j, dx};
                    g[k] = (Mdl::cost(Xp, U0, e, ph, prm) - f0) / dx;
                }
                for (int k = lane; k < ph * NU; k += 64) {
                    const int i = k / NU, j = k - i * NU;
                    const double du = dv * Ua(j);
                    const Pert Up{Us, NU, i, i == ph - 1 ? ph : -1, j, du};     // the last row moves with its copy
                    Jm[k] = (Mdl::cost(X0, Up, e, ph, prm) - f0) / du;
                }
                nl_wave_sync();
                for (int k = lane; k < ch * NU; k += 64) {
                    const int bl = k / NU, j = k - bl * NU;
                    double s = 0;
                    for (int i = 0; i < ph; ++i) if (min(i, ch - 1) == bl) s += Jm[i * NU + j];
                    g[ph * NX + k] = s;                                        // Iz2u' * vec(Jmv)
                }
                if (lane == 0) {
                    const double de = fmax(dv, fabs(e)) * dv;
                    g[nz - 1] = (Mdl::cost(X0, U0, e + de, ph, prm) - Mdl::cost(X0, U0, e - de, ph, prm)) / (2 * de);
                }
                nl_wave_sync();
            }
        }

        // ---- Constraints::getStateEqConstraints: value and the blocks [dc/dx_i | dc/dx_{i+1} | dc/du_i]
        if (Bt.ceq || Bt.jeq) {
            const double h = 0.5 * M.Ts;
            const int W = 2 * NX + NU;
            for (int k = lane; k < ph * (W + 1); k += 64) {
                const int i = k / (W + 1), c = k - i * (W + 1);     // c = 0: value; 1..: one Jacobian column
                double xk[NX], xk1[NX], uk[NU], fa[NX], fb[NX];
                for (int a = 0; a < NX; ++a) { xk[a] = Xs[i * NX + a]; xk1[a] = Xs[(i + 1) * NX + a]; }
                for (int a = 0; a < NU; ++a) uk[a] = Us[i * NU + a];
                if (c == 0) {
                    if (!Bt.ceq) continue;
                    double *cv = Bt.ceq + (size_t)b * ph * NX + i * NX;
                    Mdl::f(fa, xk, uk, prm);
                    if (Mdl::CONTINUOUS) {
                        Mdl::f(fb, xk1, uk, prm);
                        for (int a = 0; a < NX; ++a) cv[a] = xk[a] + (h * (fa[a] + fb[a])) - xk1[a];
                    } else {
                        for (int a = 0; a < NX; ++a) cv[a] = xk1[a] - fa[a];
                    }
                    continue;
                }
                if (!Bt.jeq) continue;
                double *J = Bt.jeq + ((size_t)b * ph + i) * NX * W;          // [NX x W] row-major block of step i
                const int col = c - 1;
                auto cdiff = [&](const double *xx, const double *uu, int v, bool isu, double *out) {
                    double xp[NX], up[NU], f1[NX], f2[NX];
                    for (int a = 0; a < NX; ++a) xp[a] = xx[a];
                    for (int a = 0; a < NU; ++a) up[a] = uu[a];
                    const double base = isu ? uu[v] : xx[v];
                    const double d = dv * fmax(fabs(base), 1.0);
                    if (isu) up[v] = base + d; else xp[v] = base + d;
                    Mdl::f(f1, xp, up, prm);
                    if (isu) up[v] = base - d; else xp[v] = base - d;
                    Mdl::f(f2, xp, up, prm);
                    for (int a = 0; a < NX; ++a) out[a] = (f1[a] - f2[a]) / (2 * d);
                };
                double dcol[NX];
                if (col < NX) {                    // d c_i / d x_i  (not a decision variable for i = 0: kept for the caller to drop)
                    cdiff(xk, uk, col, false, dcol);
                    for (int a = 0; a < NX; ++a)
                        J[a * W + col] = Mdl::CONTINUOUS ? ((a == col ? 1.0 : 0.0) + h * dcol[a]) : -dcol[a];
                } else if (col < 2 * NX) {         // d c_i / d x_{i+1}
                    const int v = col - NX;
                    if (Mdl::CONTINUOUS) {
                        cdiff(xk1, uk, v, false, dcol);
                        for (int a = 0; a < NX; ++a) J[a * W + col] = (a == v ? -1.0 : 0.0) + h * dcol[a];
                    } else {
                        for (int a = 0; a < NX; ++a) J[a * W + col] = (a == v ? 1.0 : 0.0);
                    }
                } else {                           // d c_i / d u_i
                    const int v = col - 2 * NX;
                    cdiff(xk, uk, v, true, dcol);
                    if (Mdl::CONTINUOUS) {
                        double d2[NX];
                        cdiff(xk1, uk, v, true, d2);
                        for (int a = 0; a < NX; ++a) J[a * W + col] = h * (dcol[a] + d2[a]);
                    } else {
                        for (int a = 0; a < NX; ++a) J[a * W + col] = -dcol[a];
                    }
                }
            }
        }

        // ---- Constraints::evaluateIneq + computeIneqJacobian (dense [nineq x nz], row-major)
        if (Bt.cineq)
            for (int k = lane; k < nineq; k += 64) Bt.cineq[(size_t)b * nineq + k] = Mdl::ineq(k, X0, U0, e, ph, prm);
        if (Bt.jineq) {
            double *J = Bt.jineq + (size_t)b * nineq * nz;
            for (int k = lane; k < nz; k += 64) {
                if (k < ph * NX) {
                    const int i = k / NX, j = k - i * NX;
                    const double dx = dv * Xa(j);
                    const Pert Xp{Xs, NX, i + 1, -1, j, dx}, Xm{Xs, NX, i + 1, -1, j, -dx};
                    for (int r = 0; r < nineq; ++r)
                        J[(size_t)r * nz + k] = (Mdl::ineq(r, Xp, U0, e, ph, prm) - Mdl::ineq(r, Xm, U0, e, ph, prm)) / (2 * dx);
                } else if (k < nz - 1) {
                    const int q = k - ph * NX, bl = q / NU, j = q - bl * NU;
                    const double du = dv * Ua(j);
                    for (int r = 0; r < nineq; ++r) {
                        double s = 0;
                        for (int i = 0; i < ph; ++i) {          // every input row of the block on its own (no pairing here)
                            if (min(i, ch - 1) != bl) continue;
                            const Pert Up{Us, NU, i, -1, j, du}, Um{Us, NU, i, -1, j, -du};
                            s += (Mdl::ineq(r, X0, Up, e, ph, prm) - Mdl::ineq(r, X0, Um, e, ph, prm)) / (2 * du);
                        }
                        J[(size_t)r * nz + k] = s;
                    }
                } else {
                    const double de = fmax(dv, fabs(e)) * dv;
                    for (int r = 0; r < nineq; ++r)
                        J[(size_t)r * nz + k] = (Mdl::ineq(r, X0, U0, e + de, ph, prm) - Mdl::ineq(r, X0, U0, e - de, ph, prm)) / (2 * de);
                }
            }
        }
        nl_wave_sync();
    }
}

}  // namespace

int nlmpc_model_dims(int model_id, int *nx, int *nu)
{
    switch (model_id) {
    case 1: *nx = VanDerPol::NX; *nu = VanDerPol::NU; return 0;
    case 2: *nx = Ugv::NX; *nu = Ugv::NU; return 0;
    default: return -1;
    }
}

int nlmpc_launch(const NlmpcDev &m, const NlmpcBatchDev &b, void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int wpb = 4;
    int blocks = (b.batch + wpb - 1) / wpb;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    const size_t lds = (size_t)wpb * m.lds_per_wave * sizeof(double);
    if (lds > 64 * 1024) return -2;
    switch (m.model_id) {
    case 1: hipLaunchKernelGGL(nlmpc_evaluate<VanDerPol>, dim3(blocks), dim3(wpb * 64), lds, s, m, b); break;
    case 2: hipLaunchKernelGGL(nlmpc_evaluate<Ugv>, dim3(blocks), dim3(wpb * 64), lds, s, m, b); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace mpcx
