// The NLMPC engine: kernel templates over a model type, for gfx950, one instance per wavefront.
//
// (1) nlmpc_evaluate: what libmpc++ evaluates inside every NLopt SLSQP callback
//     (reference include/mpc/NLMPC/NLOptimizer.hpp:760-997 -> Objective.hpp:91-265, Constraints.hpp:211-316,
//     490-905, Mapping.hpp:174-211) for a batch of decision vectors:
//       - unwrap z into (X, U, slack) with move blocking and the mapping's scalings  (Mapping::unwrapVector)
//       - cost + forward-difference gradient, with the reference's step quirk  (Objective::computeGradient)
//       - dynamics equalities (trapezoidal collocation or one-step) + central-difference blocks
//                                                                              (getStateEqConstraints)
//       - user inequalities + central-difference Jacobian                      (computeIneqJacobian)
//       - user equalities + central-difference Jacobian, with their own step rules (computeEqJacobian)
// (2) nlmpc_sqp: the optimisation NLOptimizer::run (NLOptimizer.hpp:412-638) hands to nlopt LD_SLSQP, as a
//     sequential quadratic programme on the same transcription: the dynamics equalities are eliminated by a
//     forward sweep over their block-bidiagonal Jacobian (condensing), the quadratic sub-problem lives in the
//     move-blocked inputs (+ slack), its Hessian is a damped BFGS estimate kept in inverse form, it is solved by
//     a dual active-set method (Goldfarb-Idnani, range-space form on that inverse), and the step is globalised by
//     a backtracking search on an l1 merit function whose trial points are evaluated by the lanes in parallel.
//
// The user hooks of the reference are host std::function objects (IDimensionable.hpp:94-149) which a kernel
// cannot call.  The engine is therefore a header: it is instantiated for a model type wherever that type is known --
//   * in libmpcx.so for the zoo of mpcx/nlmpc_models.hpp (component-wise constraint functors with declared structure);
//   * in the user's own translation unit, compiled by hipcc, for hooks written with the reference's signatures
//     (mpcx/nlmpc_hooks.hpp, mpc::NLMPC<>::setStateSpaceFunction & co.);
//   * at run time by hipRTC for hooks given as source text (mpcx_nlmpc_create_from_source, the Python front-end).
// Every finite-difference column is an independent re-evaluation of a whole-horizon function: lanes own columns, the
// unwrapped trajectory sits in the wave's LDS slice and perturbations are applied on the fly by the accessor.
#pragma once

#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>

#include <cmath>
#endif

#include "nlmpc_device.hpp"
#include "nlmpc_models.hpp"
#if !defined(__HIPCC_RTC__)
#include <cstdio>
#include <cstdlib>
#endif

namespace mpcx {
namespace engine {

using namespace models;

typedef double __attribute__((address_space(1))) *gwp;      // a pointer known to point into HBM
// The defects and the dynamics blocks: in HBM, or (BLK: small systems, nlmpc_plan) in the wavefront's LDS slice.  Either way the
// pointer is re-derived so that the compiler knows the address space: flat accesses wait on both memory counters, i.e. for every
// outstanding global load and store as well (measured: the LDS-resident blocks behind generic pointers bought nothing).
template <bool BLK> struct BlockPtr {
    typedef gwp type;
    static __device__ __forceinline__ type make(double *p) { return (gwp)p; }
};
template <> struct BlockPtr<true> {
    typedef double *type;
    static __device__ __forceinline__ type make(double *p)
    {
        extern __shared__ __attribute__((aligned(16))) double lds_base[];
        return lds_base + (p - lds_base);
    }
};

__device__ __forceinline__ void nl_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Wave-wide reductions, the result in every lane.  Within a row of sixteen lanes by DPP operations (VALU, no LDS round trip --
// the __shfl_xor butterfly compiles to six dependent ds_bpermute, about 700 cycles a reduction), across the four rows by
// v_readlane.  The order of the additions is fixed: the same bits on every run.
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
    return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double lane_d(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// 0xB1: quad_perm [1,0,3,2]; 0x4E: quad_perm [2,3,0,1]; 0x141: row_half_mirror; 0x140: row_mirror
__device__ __forceinline__ double wave_sum(double v)
{
    // the argument is often a product: without this the compiler contracts it into the first addition for one operand (fma(a, b,
    // moved copy of round(a b))) wherever inlining lets it see the product -- the same source then rounds differently from one
    // instantiation to the next
    asm volatile("" : "+v"(v));
    v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); v += dpp_d<0x141>(v); v += dpp_d<0x140>(v);
    return (lane_d(v, 0) + lane_d(v, 16)) + (lane_d(v, 32) + lane_d(v, 48));
}
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_d<0xB1>(v)); v = fmax(v, dpp_d<0x4E>(v)); v = fmax(v, dpp_d<0x141>(v)); v = fmax(v, dpp_d<0x140>(v));
    return fmax(fmax(lane_d(v, 0), lane_d(v, 16)), fmax(lane_d(v, 32), lane_d(v, 48)));
}

// sum_j a[j * stride] * x[j]: the loads of a batch are issued together and waited for once -- the compiler does not
// pipeline loads across the iterations of a plain loop, and every iteration would pay the full memory latency
// (U: loads in flight at a time; kernels that run one wavefront per SIMD have the registers for 32)
template <int U = 8>
__device__ __forceinline__ double gdot(const double *__restrict__ a, size_t stride, const double *x, int n)
{
    double s = 0;
    int j = 0;
    for (; j + U <= n; j += U) {
        double v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[(size_t)(j + u) * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = x[j + u];
#pragma unroll
        for (int u = 0; u < U; ++u) s = fma(v[u], w[u], s);
    }
    if constexpr (U > 8) {
        // the remainder in one more batch (clamped addresses, surplus products dropped): a short vector still has all its loads in flight
        if (j < n) {
            double v[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int jj = min(j + u, n - 1); v[u] = a[(size_t)jj * stride]; w[u] = x[jj]; }
#pragma unroll
            for (int u = 0; u < U; ++u) if (j + u < n) s = fma(v[u], w[u], s);
        }
    } else {
        for (; j < n; ++j) s = fma(a[(size_t)j * stride], x[j], s);
    }
    return s;
}
// the same with both operands strided in memory
__device__ __forceinline__ double gdot2(const double *__restrict__ a, size_t sa, const double *__restrict__ b, size_t sb, int n)
{
    constexpr int U = 8;
    double s = 0;
    int j = 0;
    for (; j + U <= n; j += U) {
        double v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = a[(size_t)(j + u) * sa]; w[u] = b[(size_t)(j + u) * sb]; }
#pragma unroll
        for (int u = 0; u < U; ++u) s = fma(v[u], w[u], s);
    }
    for (; j < n; ++j) s = fma(a[(size_t)j * sa], b[(size_t)j * sb], s);
    return s;
}
// largest value and the lowest lane-supplied index holding it
__device__ __forceinline__ void wave_argmax(double &v, int &idx)
{
    auto take = [&](double ov, int oi) { if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; } };
    take(dpp_d<0xB1>(v), dpp_i<0xB1>(idx));
    take(dpp_d<0x4E>(v), dpp_i<0x4E>(idx));
    take(dpp_d<0x141>(v), dpp_i<0x141>(idx));
    take(dpp_d<0x140>(v), dpp_i<0x140>(idx));
    const double v0 = lane_d(v, 0), v1 = lane_d(v, 16), v2 = lane_d(v, 32), v3 = lane_d(v, 48);
    const int i0 = __builtin_amdgcn_readlane(idx, 0), i1 = __builtin_amdgcn_readlane(idx, 16), i2 = __builtin_amdgcn_readlane(idx, 32),
              i3 = __builtin_amdgcn_readlane(idx, 48);
    v = v0; idx = i0;
    take(v1, i1); take(v2, i2); take(v3, i3);
}

constexpr double kDv = 1.4901161193847656e-08;          // sqrt(DBL_EPSILON), Objective.hpp:283

// Mapping scalings (Mapping.hpp:71-86).  Identity unless NLMPC::setInputScale / setStateScale were called: then nothing is
// loaded or multiplied (the branch is wave-uniform); otherwise divisions go through the reciprocals the host prepared.
struct Scale {
    const double *su, *ss, *iss;
    bool on;
    __device__ __forceinline__ explicit Scale(const NlmpcDev &M) : su(M.su), ss(M.ss), iss(M.iss), on(M.scaled != 0) {}
    __device__ __forceinline__ Scale(const double *su_, const double *ss_, const double *iss_, bool on_) : su(su_), ss(ss_), iss(iss_), on(on_) {}
    __device__ __forceinline__ double by_su(double v, int j) const { return on ? v * su[j] : v; }
    __device__ __forceinline__ double by_ss(double v, int j) const { return on ? v * ss[j] : v; }
    __device__ __forceinline__ double over_ss(double v, int j) const { return on ? v * iss[j] : v; }
};

// zoo models fix the kind of transcription at compile time; hook models may be either (setDiscretizationSamplingTime)
template <class Mdl> __device__ __forceinline__ bool is_ct(const NlmpcDev &M)
{
    if constexpr (Mdl::VECTOR_HOOKS) return M.continuous != 0; else return Mdl::CONTINUOUS;
}
// StateFunHandle (IDimensionable.hpp:130-134): the hook form also receives the horizon step
template <class Mdl> __device__ __forceinline__ void call_f(double *out, const double *x, const double *u, const double *prm, int step)
{
    if constexpr (Mdl::VECTOR_HOOKS) Mdl::f(out, x, u, prm, (unsigned)step); else Mdl::f(out, x, u, prm);
}
template <class Mdl> __device__ __forceinline__ void call_out(double *y, const double *x, const double *u, const double *prm, int step)
{
    if constexpr (Mdl::VECTOR_HOOKS) Mdl::out(y, x, u, prm, (unsigned)step); else Mdl::out(y, x, u, prm);
}

// Mapping::unwrapVector (Mapping.hpp:174-211): U = input_scale * (blocked z_u), X = [x0; z_x] / state_scale -- x0's row too
template <class Mdl>
__device__ __forceinline__ void unwrap(const NlmpcDev &M, const double *z, const double *x0, double *Xs, double *Us, int lane)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int ph = M.ph, ch = M.ch;
    const Scale sc(M);
    for (int k = lane; k < (ph + 1) * NX; k += 64) {
        const int i = k / NX, j = k - i * NX;
        Xs[k] = sc.over_ss(i == 0 ? x0[j] : z[(i - 1) * NX + j], j);
    }
    for (int k = lane; k < (ph + 1) * NU; k += 64) {
        const int i = k / NU, j = k - i * NU;
        const int blk = min(min(i, ph - 1), ch - 1);          // first ch-1 moves one step each, the last one held
        Us[k] = sc.by_su(z[ph * NX + blk * NU + j], j);
    }
    nl_wave_sync();
}

// ---------------------------------------------------------------------------------------------------
// hook models (mpcx/nlmpc_hooks.hpp): the reference's own hook signatures -- whole constraint vectors, matrix arguments
// ---------------------------------------------------------------------------------------------------
// OutFunHandle on one row of the trajectory, optionally with one perturbed state or input component (Model::getOutput,
// Model.hpp:72-96, is re-run by the reference inside every finite-difference perturbation)
template <class Mdl>
__device__ __forceinline__ void hook_out_row(double *yo, const double *Xs, const double *Us, int row, int jx, double dx, int ju,
                                             double du, const double *prm)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    double xr[NX], ur[NU];
    for (int a = 0; a < NX; ++a) xr[a] = Xs[row * NX + a] + (a == jx ? dx : 0.0);
    for (int a = 0; a < NU; ++a) ur[a] = Us[row * NU + a] + (a == ju ? du : 0.0);
    Mdl::out(yo, xr, ur, prm, (unsigned)row);
}

// the outputs along the (unperturbed) trajectory into Ys [(ph+1) x NY]
template <class Mdl>
__device__ __forceinline__ void hook_base_outputs(const NlmpcDev &M, const double *Xs, const double *Us, double *Ys, int lane)
{
    if constexpr (Mdl::VECTOR_HOOKS) {
        if (M.has_output) {
            // every lane makes the call (surplus lanes redo row ph): no divergence around a hook that may sit behind a pointer
            for (int i0 = 0; i0 <= M.ph; i0 += 64) {
                const int i = min(i0 + lane, M.ph);
                double yr[Mdl::NY > 0 ? Mdl::NY : 1];
                hook_out_row<Mdl>(yr, Xs, Us, i, -1, 0.0, -1, 0.0, M.params);
                if (i0 + lane <= M.ph) for (int a = 0; a < Mdl::NY; ++a) Ys[i * Mdl::NY + a] = yr[a];
            }
            nl_wave_sync();
        }
    }
}

// Objective::evaluate + computeGradient (Objective.hpp:91-265) through ObjFunHandle(X, Y, U, slack)
template <class Mdl>
__device__ __forceinline__ void hook_cost_grad(const NlmpcDev &M, const double *z, const double *Xs, const double *Us, double *Jm,
                                               const double *Ys, int lane, double *cost, double *grad)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NY = Mdl::NY, NYA = NY > 0 ? NY : 1;
    using MX = typename Mdl::MatX; using MU = typename Mdl::MatU; using MY = typename Mdl::MatY;
    const int ph = M.ph, ch = M.ch, nz = M.nz;
    const double dv = kDv, e = z[nz - 1];
    const double *prm = M.params;
    const bool ho = M.has_output != 0;
    auto Xa = [&](int j) { const double v = fabs(Xs[(j % (ph + 1)) * NX + j / (ph + 1)]); return v > 1.0 ? v : 1.0; };
    auto Ua = [&](int j) { const double v = fabs(Us[(j % (ph + 1)) * NU + j / (ph + 1)]); return v > 1.0 ? v : 1.0; };
    const MX X0 = MX::trajectory(Xs);
    const MU U0 = MU::trajectory(Us);
    const MY Y0 = ho ? MY::trajectory(Ys) : MY::zeros_view();
    const double f0 = Mdl::cost(X0, Y0, U0, e, prm);
    if (lane == 0 && cost) *cost = f0;
    if (!grad) return;
    double *g = grad;
    // Every lane runs the same number of passes and makes the same calls in each (a surplus lane repeats the last column and
    // keeps the result to itself): a hook may sit behind a function pointer, and call sites inside control flow that differs
    // between lanes are where that goes wrong.
    for (int k0 = 0; k0 < ph * NX; k0 += 64) {
        const bool live = k0 + lane < ph * NX;
        const int k = live ? k0 + lane : ph * NX - 1;
        const int i = k / NX, j = k - i * NX;
        const double dx = dv * Xa(j);
        MX Xp = MX::trajectory(Xs); Xp.perturb(i + 1, -1, j, dx);
        MY Yp = Y0;
        double yo1[NYA];
        if (ho) { hook_out_row<Mdl>(yo1, Xs, Us, i + 1, j, dx, -1, 0.0, prm); Yp.perturb(i + 1, -1, -1, 0.0).replace_rows(yo1, nullptr); }
        const double fp = Mdl::cost(Xp, Yp, U0, e, prm);
        if (live) g[k] = (fp - f0) / dx;
    }
    for (int k0 = 0; k0 < ph * NU; k0 += 64) {
        const bool live = k0 + lane < ph * NU;
        const int k = live ? k0 + lane : ph * NU - 1;
        const int i = k / NU, j = k - i * NU;
        const double du = dv * Ua(j);
        const int pair = i == ph - 1 ? ph : -1;                               // the last row moves with its copy
        MU Up = MU::trajectory(Us); Up.perturb(i, pair, j, du);
        MY Yp = Y0;
        double yo1[NYA], yo2[NYA];
        if (ho) {
            hook_out_row<Mdl>(yo1, Xs, Us, i, -1, 0.0, j, du, prm);
            hook_out_row<Mdl>(yo2, Xs, Us, ph, -1, 0.0, j, du, prm);           // used only by the paired last row
            Yp.perturb(i, pair, -1, 0.0).replace_rows(yo1, pair >= 0 ? yo2 : nullptr);
        }
        const double fp = Mdl::cost(X0, Yp, Up, e, prm);
        if (live) Jm[k] = (fp - f0) / du;
    }
    nl_wave_sync();
    for (int k = lane; k < ch * NU; k += 64) {
        const int bl = k / NU, j = k - bl * NU;
        double s = 0;
        for (int i = 0; i < ph; ++i) if (min(i, ch - 1) == bl) s += Jm[i * NU + j];
        g[ph * NX + k] = Scale(M).by_su(s, j);                                // Iz2u' * vec(Jmv)
    }
    {
        const double de = fmax(dv, fabs(e)) * dv;
        const double ge = (Mdl::cost(X0, Y0, U0, e + de, prm) - Mdl::cost(X0, Y0, U0, e - de, prm)) / (2 * de);
        if (lane == 0) g[nz - 1] = ge;
    }
    nl_wave_sync();
}

// Constraints::evaluateIneq / evaluateEq with their central-difference Jacobians (Constraints.hpp:211-442, 641-832)
// through IConFunHandle / EConFunHandle.  A lane owns a column; the hook fills the whole vector at the point moved up
// into one column buffer and at the point moved down into another (hk: [2][rows][64]); their difference is the column.
template <class Mdl>
__device__ __forceinline__ void hook_constraints(const NlmpcDev &M, const double *z, const double *Xs, const double *Us,
                                                 const double *Ys, double *hk, int lane, double *cineq, double *jineq)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NY = Mdl::NY, NYA = NY > 0 ? NY : 1, NI = Mdl::NI, NE = Mdl::NE;
    using MX = typename Mdl::MatX; using MU = typename Mdl::MatU; using MY = typename Mdl::MatY;
    using VI = typename Mdl::VecI; using VE = typename Mdl::VecE;
    const int ph = M.ph, ch = M.ch, nz = M.nz, m = NI + NE;
    const double dv = kDv, e = z[nz - 1];
    const double *prm = M.params;
    const bool ho = M.has_output != 0;
    const Scale sc(M);
    auto Xa = [&](int j) { const double v = fabs(Xs[(j % (ph + 1)) * NX + j / (ph + 1)]); return v > 1.0 ? v : 1.0; };
    auto Ua = [&](int j) { const double v = fabs(Us[(j % (ph + 1)) * NU + j / (ph + 1)]); return v > 1.0 ? v : 1.0; };
    const MX X0 = MX::trajectory(Xs);
    const MU U0 = MU::trajectory(Us);
    const MY Y0 = ho ? MY::trajectory(Ys) : MY::zeros_view();
    double *J = jineq;
    double *cA = hk + lane, *cB = hk + (size_t)64 * m + lane;                  // element r at [r * 64]
    // values: every lane evaluates the vectors into its own column buffer (same calls in every lane, see hook_cost_grad),
    // lane 0 files them
    if (cineq) {
        if constexpr (NI > 0) { VI o = VI::output(cA, 64); Mdl::ineq_all(o, X0, Y0, U0, e, prm); }
        if constexpr (NE > 0) { VE o = VE::output(cA + (size_t)64 * NI, 64); Mdl::eq_all(o, X0, U0, prm); }
        if (lane == 0) for (int r = 0; r < m; ++r) cineq[r] = cA[r * 64];
    }
    if (!jineq) { nl_wave_sync(); return; }
    // state columns
    for (int k0 = 0; k0 < ph * NX; k0 += 64) {
        const bool live = k0 + lane < ph * NX;
        const int k = live ? k0 + lane : ph * NX - 1;
        const int i = k / NX, j = k - i * NX;
        if constexpr (NI > 0) {
            const double dx = dv * Xa(j);
            for (int sgn = 0; sgn < 2; ++sgn) {
                const double d = sgn ? -dx : dx;
                MX Xp = MX::trajectory(Xs); Xp.perturb(i + 1, -1, j, d);
                MY Yp = Y0;
                double yo1[NYA];
                if (ho) { hook_out_row<Mdl>(yo1, Xs, Us, i + 1, j, d, -1, 0.0, prm); Yp.perturb(i + 1, -1, -1, 0.0).replace_rows(yo1, nullptr); }
                VI o = VI::output(sgn ? cB : cA, 64);
                Mdl::ineq_all(o, Xp, Yp, U0, e, prm);
            }
            // computeIneqJacobian multiplies the state columns by the state scaling (Constraints.hpp:269-284)
            if (live) for (int r = 0; r < NI; ++r) J[(size_t)r * nz + k] = sc.by_ss((cA[r * 64] - cB[r * 64]) / (2 * dx), j);
        }
        if constexpr (NE > 0) {
            const double dx = dv * fmax(fabs(Xs[(i + 1) * NX + j]), 1.0);
            for (int sgn = 0; sgn < 2; ++sgn) {
                MX Xp = MX::trajectory(Xs); Xp.perturb(i + 1, -1, j, sgn ? -dx : dx);
                VE o = VE::output((sgn ? cB : cA) + (size_t)64 * NI, 64);
                Mdl::eq_all(o, Xp, U0, prm);
            }
            if (live) for (int r = NI; r < m; ++r) J[(size_t)r * nz + k] = sc.by_ss((cA[r * 64] - cB[r * 64]) / (2 * dx), j);
        }
    }
    // input columns: block bl drives the steps i_first..i_last; the passes over the steps are the same for every lane (the
    // longest block), a lane whose block is shorter repeats its last step and drops the result
    const int span_max = ph - ch + 1;
    for (int q0 = 0; q0 < ch * NU; q0 += 64) {
        const bool live = q0 + lane < ch * NU;
        const int q = live ? q0 + lane : ch * NU - 1;
        const int bl = q / NU, j = q - bl * NU, k = ph * NX + q;
        const int i_first = bl, i_last = bl == ch - 1 ? ph - 1 : bl;           // the steps this block drives
        if (live) for (int r = 0; r < m; ++r) J[(size_t)r * nz + k] = 0.0;
        for (int t = 0; t < span_max; ++t) {
            const bool step_live = live && i_first + t <= i_last;
            const int i = min(i_first + t, i_last);
            if constexpr (NI > 0) {
                const double du = dv * Ua(j);
                for (int sgn = 0; sgn < 2; ++sgn) {                            // every input row of the block on its own (no pairing here)
                    const double d = sgn ? -du : du;
                    MU Up = MU::trajectory(Us); Up.perturb(i, -1, j, d);
                    MY Yp = Y0;
                    double yo1[NYA];
                    if (ho) { hook_out_row<Mdl>(yo1, Xs, Us, i, -1, 0.0, j, d, prm); Yp.perturb(i, -1, -1, 0.0).replace_rows(yo1, nullptr); }
                    VI o = VI::output(sgn ? cB : cA, 64);
                    Mdl::ineq_all(o, X0, Yp, Up, e, prm);
                }
                if (step_live) for (int r = 0; r < NI; ++r) J[(size_t)r * nz + k] += (cA[r * 64] - cB[r * 64]) / (2 * du);
            }
            if constexpr (NE > 0) {
                const double du = dv * fmax(fabs(Us[(ph - 1) * NU + j]), 1.0);  // row ph-1's magnitude for every step (Constraints.hpp:780,806)
                for (int sgn = 0; sgn < 2; ++sgn) {
                    MU Up = MU::trajectory(Us); Up.perturb(i, i == ph - 1 ? ph : -1, j, sgn ? -du : du);
                    VE o = VE::output((sgn ? cB : cA) + (size_t)64 * NI, 64);
                    Mdl::eq_all(o, X0, Up, prm);
                }
                if (step_live) for (int r = NI; r < m; ++r) J[(size_t)r * nz + k] += (cA[r * 64] - cB[r * 64]) / (2 * du);
            }
        }
        if (live && sc.on) for (int r = 0; r < m; ++r) J[(size_t)r * nz + k] *= sc.su[j];      // glueJacobian: Jmanvar * Iz2u
    }
    // slack column (every lane computes it, lane 0 files it)
    {
        const int k = nz - 1;
        const double de = fmax(dv, fabs(e)) * dv;
        if constexpr (NI > 0) {
            for (int sgn = 0; sgn < 2; ++sgn) {
                VI o = VI::output(sgn ? cB : cA, 64);
                Mdl::ineq_all(o, X0, Y0, U0, sgn ? e - de : e + de, prm);
            }
            if (lane == 0) for (int r = 0; r < NI; ++r) J[(size_t)r * nz + k] = (cA[r * 64] - cB[r * 64]) / (2 * de);
        }
        if (lane == 0) for (int r = NI; r < m; ++r) J[(size_t)r * nz + k] = 0.0;
    }
    nl_wave_sync();
}

// the transcription of one instance; any output may be null.  Xs/Us/Jm/Ys: this wave's LDS; hk: hook scratch (HBM).
template <class Mdl, bool BLK = false>
__device__ void eval_instance(const NlmpcDev &M, const double *z, const double *x0, double *Xs, double *Us, double *Jm, double *Ys,
                              double *hk, int lane, double *cost, double *grad, double *ceq, double *jeq, double *cineq,
                              double *jineq, bool jin_fill = true, const double *prm_instance = nullptr)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    if constexpr (!Mdl::VECTOR_HOOKS) {
        // Inside the SQP kernels this function is a real call, and its pointer arguments arrive as generic pointers: every
        // access to the trajectories would be a flat load (278 of them in the 8-oscillator instance, none to LDS as such).
        // Re-deriving them from the block's dynamic shared memory tells the compiler where they point.  Not for hook models:
        // their hooks read the trajectories through views that hold generic pointers, and LDS accesses of both kinds to the
        // same data are not ordered with respect to each other (seen: an output trajectory written one way and read the other).
        extern __shared__ __attribute__((aligned(16))) double lds_base[];
        Xs = lds_base + (Xs - lds_base); Us = lds_base + (Us - lds_base); Jm = lds_base + (Jm - lds_base);
    }
    const int ph = M.ph, ch = M.ch, nz = M.nz, nineq = M.nineq;
    const double dv = kDv;
    const double *prm = prm_instance ? prm_instance : M.params;     // the instance's own model parameters, if the batch brings them
    const Scale sc(M);
    const bool CT = is_ct<Mdl>(M);
    unwrap<Mdl>(M, z, x0, Xs, Us, lane);
    hook_base_outputs<Mdl>(M, Xs, Us, Ys, lane);
    const double e = z[nz - 1];
    auto Xa = [&](int j) { const double v = fabs(Xs[(j % (ph + 1)) * NX + j / (ph + 1)]); return v > 1.0 ? v : 1.0; };
    auto Ua = [&](int j) { const double v = fabs(Us[(j % (ph + 1)) * NU + j / (ph + 1)]); return v > 1.0 ? v : 1.0; };
    const Pert X0{Xs, NX, -1, -1, -1, 0.0}, U0{Us, NU, -1, -1, -1, 0.0};

    // ---- Objective::evaluate + computeGradient
    if constexpr (Mdl::VECTOR_HOOKS) {
        if (cost || grad) hook_cost_grad<Mdl>(M, z, Xs, Us, Jm, Ys, lane, cost, grad);
    } else if (cost || grad) {
        const double f0 = Mdl::cost(X0, U0, e, ph, prm);
        if (lane == 0 && cost) *cost = f0;
        if (grad) {
            gwp g = (gwp)grad;                               // (HBM: global stores, not flat ones that also tie up the LDS counter)
            // one loop over every perturbed copy of the point -- ph nx state entries, ph nu input entries, the slack both ways --
            // so that the lanes share them evenly (each is a whole evaluation of the cost); one call site for all kinds
            const int nxv = ph * NX, nuv = ph * NU, nall = nxv + nuv + 2;
            const double de = fmax(dv, fabs(e)) * dv;
            double fslack = 0.0;
            for (int idx = lane; idx < nall; idx += 64) {
                const bool isx = idx < nxv, isu = !isx && idx < nxv + nuv;
                const int kk = isx ? idx : idx - nxv;
                const int i = isx ? kk / NX : (isu ? kk / NU : 0), j = isx ? kk - i * NX : (isu ? kk - i * NU : 0);
                const double dx = dv * Xa(j), du = dv * Ua(j);
                const Pert Xp{Xs, NX, isx ? i + 1 : -1, -1, isx ? j : -1, isx ? dx : 0.0};       // no chain rule for the state scaling (Objective.hpp:107-144)
                const Pert Up{Us, NU, isu ? i : -1, (isu && i == ph - 1) ? ph : -1, isu ? j : -1, isu ? du : 0.0};   // the last row moves with its copy
                const double ee = idx == nall - 2 ? e + de : (idx == nall - 1 ? e - de : e);
                const double fp = Mdl::cost(Xp, Up, ee, ph, prm);
                if (isx) g[kk] = (fp - f0) / dx;
                else if (isu) Jm[kk] = (fp - f0) / du;
                else fslack = fp;
            }
            const double fplus = __shfl(fslack, (nall - 2) & 63), fminus = __shfl(fslack, (nall - 1) & 63);
            nl_wave_sync();
            for (int k = lane; k < ch * NU; k += 64) {
                const int bl = k / NU, j = k - bl * NU;
                double s = 0;
                for (int i = 0; i < ph; ++i) if (min(i, ch - 1) == bl) s += Jm[i * NU + j];
                g[ph * NX + k] = sc.by_su(s, j);                            // Iz2u' * vec(Jmv)
            }
            if (lane == 0) g[nz - 1] = (fplus - fminus) / (2 * de);
            nl_wave_sync();
        }
    }

    // ---- Constraints::getStateEqConstraints: value and the blocks [dc/dx_i | dc/dx_{i+1} | dc/du_i], scaled as
    // Constraints.hpp:515-517,545,565-571,595,607-608 (value / state_scale; Sx A Tx, Sx B; the input block times the input scale)
    if (ceq || jeq) {
        const double h = 0.5 * M.Ts;
        const int W = 2 * NX + NU;
        for (int k = lane; k < ph * (W + 1); k += 64) {
            const int i = k / (W + 1), c = k - i * (W + 1);     // c = 0: value; 1..: one Jacobian column
            double xk[NX], xk1[NX], uk[NU], fa[NX], fb[NX];
            for (int a = 0; a < NX; ++a) { xk[a] = Xs[i * NX + a]; xk1[a] = Xs[(i + 1) * NX + a]; }
            for (int a = 0; a < NU; ++a) uk[a] = Us[i * NU + a];
            if (c == 0) {
                if (!ceq) continue;
                typename BlockPtr<BLK>::type cv = BlockPtr<BLK>::make(ceq) + i * NX;
                call_f<Mdl>(fa, xk, uk, prm, i);
                if (CT) {
                    call_f<Mdl>(fb, xk1, uk, prm, i);
                    for (int a = 0; a < NX; ++a) cv[a] = sc.over_ss(xk[a] + (h * (fa[a] + fb[a])) - xk1[a], a);
                } else {
                    for (int a = 0; a < NX; ++a) cv[a] = sc.over_ss(xk1[a] - fa[a], a);
                }
                continue;
            }
            if (!jeq) continue;
            typename BlockPtr<BLK>::type J = BlockPtr<BLK>::make(jeq) + (size_t)i * NX * W;          // [NX x W] row-major block of step i
            const int col = c - 1;
            auto cdiff = [&](const double *xx, const double *uu, int v, bool isu, double *out) {
                double xp[NX], up[NU], f1[NX], f2[NX];
                for (int a = 0; a < NX; ++a) xp[a] = xx[a];
                for (int a = 0; a < NU; ++a) up[a] = uu[a];
                const double base = isu ? uu[v] : xx[v];
                const double d = dv * fmax(fabs(base), 1.0);
                if (isu) up[v] = base + d; else xp[v] = base + d;
                call_f<Mdl>(f1, xp, up, prm, i);
                if (isu) up[v] = base - d; else xp[v] = base - d;
                call_f<Mdl>(f2, xp, up, prm, i);
                for (int a = 0; a < NX; ++a) out[a] = (f1[a] - f2[a]) / (2 * d);
            };
            double dcol[NX];
            if (col < NX) {                    // d c_i / d x_i  (not a decision variable for i = 0: kept for the caller to drop)
                cdiff(xk, uk, col, false, dcol);
                for (int a = 0; a < NX; ++a) {
                    const double sa = sc.over_ss(sc.by_ss(dcol[a], col), a);
                    J[a * W + col] = CT ? ((a == col ? 1.0 : 0.0) + h * sa) : -sa;
                }
            } else if (col < 2 * NX) {         // d c_i / d x_{i+1}
                const int v = col - NX;
                if (CT) {
                    cdiff(xk1, uk, v, false, dcol);
                    for (int a = 0; a < NX; ++a) J[a * W + col] = (a == v ? -1.0 : 0.0) + h * sc.over_ss(sc.by_ss(dcol[a], v), a);
                } else {
                    for (int a = 0; a < NX; ++a) J[a * W + col] = (a == v ? 1.0 : 0.0);
                }
            } else {                           // d c_i / d u_i
                const int v = col - 2 * NX;
                cdiff(xk, uk, v, true, dcol);
                if (CT) {
                    double d2[NX];
                    cdiff(xk1, uk, v, true, d2);
                    for (int a = 0; a < NX; ++a) J[a * W + col] = sc.by_su(h * sc.over_ss(dcol[a] + d2[a], a), v);
                } else {
                    for (int a = 0; a < NX; ++a) J[a * W + col] = sc.by_su(-sc.over_ss(dcol[a], a), v);
                }
            }
        }
    }

    if constexpr (Mdl::VECTOR_HOOKS) {
        if (cineq || jineq) hook_constraints<Mdl>(M, z, Xs, Us, Ys, hk, lane, cineq, jineq);
        nl_wave_sync();
        return;
    } else {
    // ---- Constraints::evaluateIneq + computeIneqJacobian (dense [nineq x nz], row-major)
    if (cineq)
        for (int k = lane; k < nineq; k += 64) ((gwp)cineq)[k] = Mdl::ineq(k, X0, U0, e, ph, prm);
    if (jineq) {
        gwp J = (gwp)jineq;
        for (int k = lane; k < nz; k += 64) {
            if (k < ph * NX) {
                const int i = k / NX, j = k - i * NX;
                const double dx = dv * Xa(j);
                const Pert Xp{Xs, NX, i + 1, -1, j, dx}, Xm{Xs, NX, i + 1, -1, j, -dx};
                if (jin_fill) for (int r = 0; r < nineq; ++r) J[(size_t)r * nz + k] = 0.0;
                int first, count;
                Mdl::ineq_rows_of_x(i + 1, first, count);           // the lane's own rows: no divergence over the union of rows
                for (int t = 0; t < count; ++t) {
                    const int r = first + t;
                    // the state columns are multiplied by the state scaling (Constraints.hpp:269-284)
                    J[(size_t)r * nz + k] = sc.by_ss((Mdl::ineq(r, Xp, U0, e, ph, prm) - Mdl::ineq(r, Xm, U0, e, ph, prm)) / (2 * dx), j);
                }
            } else if (k < nz - 1) {
                const int q = k - ph * NX, bl = q / NU, j = q - bl * NU;
                const double du = dv * Ua(j);
                const int i_first = bl, i_last = bl == ch - 1 ? ph - 1 : bl;       // the steps this block drives
                // A block that drives several steps (the last one) adds their contributions up.  Where the model declares that the
                // steps' row ranges do not overlap every entry has one contribution: plain stores, no zeroing and no read-modify-
                // write chain through memory (each a full memory latency for the few lanes that own the last block).
                constexpr bool disjoint = Mdl::INEQ_U_ROWS_DISJOINT;
                if (jin_fill) for (int r = 0; r < nineq; ++r) J[(size_t)r * nz + k] = 0.0;
                else if (!disjoint)
                    for (int i = i_first; i <= i_last; ++i) {      // the entries that get contributions below start from zero
                        int first, count;
                        Mdl::ineq_rows_of_u(i, first, count);
                        for (int t = 0; t < count; ++t) J[(size_t)(first + t) * nz + k] = 0.0;
                    }
                for (int i = i_first; i <= i_last; ++i) {          // every input row of the block on its own (no pairing here)
                    int first, count;
                    Mdl::ineq_rows_of_u(i, first, count);
                    const Pert Up{Us, NU, i, -1, j, du}, Um{Us, NU, i, -1, j, -du};
                    for (int t = 0; t < count; ++t) {
                        const int r = first + t;
                        const double dj = sc.by_su((Mdl::ineq(r, X0, Up, e, ph, prm) - Mdl::ineq(r, X0, Um, e, ph, prm)) / (2 * du), j);
                        if constexpr (disjoint) J[(size_t)r * nz + k] = 0.0 + dj; else J[(size_t)r * nz + k] += dj;
                    }
                }
            } else {
                const double de = fmax(dv, fabs(e)) * dv;
                for (int r = 0; r < nineq; ++r) {
                    if (!Mdl::INEQ_USES_SLACK) { if (jin_fill) J[(size_t)r * nz + k] = 0.0; continue; }
                    J[(size_t)r * nz + k] = (Mdl::ineq(r, X0, U0, e + de, ph, prm) - Mdl::ineq(r, X0, U0, e - de, ph, prm)) / (2 * de);
                }
            }
        }
    }
    // ---- Constraints::evaluateEq + computeEqJacobian (Constraints.hpp:365-442, 731-832): rows nineq.. of the same arrays.
    // Steps: the perturbed element's own magnitude for the states, row ph-1's for every input step, last input row paired.
    const int nue = M.nue;
    if (nue && cineq)
        for (int k = lane; k < nue; k += 64) ((gwp)cineq)[nineq + k] = Mdl::eq(k, X0, U0, ph, prm);
    if (nue && jineq) {
        gwp J = (gwp)jineq + (size_t)nineq * nz;
        for (int k = lane; k < nz; k += 64) {
            if (k < ph * NX) {
                const int i = k / NX, j = k - i * NX;
                const double dx = dv * fmax(fabs(Xs[(i + 1) * NX + j]), 1.0);
                const Pert Xp{Xs, NX, i + 1, -1, j, dx}, Xm{Xs, NX, i + 1, -1, j, -dx};
                for (int r = 0; r < nue; ++r)
                    J[(size_t)r * nz + k] = sc.by_ss((Mdl::eq(r, Xp, U0, ph, prm) - Mdl::eq(r, Xm, U0, ph, prm)) / (2 * dx), j);
            } else if (k < nz - 1) {
                const int q = k - ph * NX, bl = q / NU, j = q - bl * NU;
                const double du = dv * fmax(fabs(Us[(ph - 1) * NU + j]), 1.0);
                const int i_first = bl, i_last = bl == ch - 1 ? ph - 1 : bl;
                for (int r = 0; r < nue; ++r) {
                    double s = 0;
                    for (int i = i_first; i <= i_last; ++i) {
                        const Pert Up{Us, NU, i, i == ph - 1 ? ph : -1, j, du}, Um{Us, NU, i, i == ph - 1 ? ph : -1, j, -du};
                        s += (Mdl::eq(r, X0, Up, ph, prm) - Mdl::eq(r, X0, Um, ph, prm)) / (2 * du);
                    }
                    J[(size_t)r * nz + k] = sc.by_su(s, j);
                }
            } else {
                for (int r = 0; r < nue; ++r) J[(size_t)r * nz + k] = 0.0;
            }
        }
    }
    nl_wave_sync();
    }
}

template <class Mdl>
__device__ __forceinline__ void evaluate_body(const NlmpcDev &M, const NlmpcBatchDev &Bt)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int ph = M.ph, nz = M.nz;
    double *Xs = smem + (size_t)wave * M.lds_per_wave;  // (ph+1) x NX
    double *Us = Xs + (ph + 1) * NX;                    // (ph+1) x NU
    double *Jm = Us + (ph + 1) * NU;                    // ph x NU scratch (gradient wrt the input rows)
    double *Ys = Jm + ph * NU;                          // (ph+1) x NY outputs along the trajectory (hook models with an output function)
    for (int b = blockIdx.x * wpb + wave; b < Bt.batch; b += gridDim.x * wpb) {
        auto at = [&](double *p, size_t stride) { return p ? p + (size_t)b * stride : nullptr; };
        eval_instance<Mdl>(M, Bt.z + (size_t)b * nz, Bt.x0 + (size_t)b * NX, Xs, Us, Jm, Ys, at(Bt.hook_ws, Bt.hook_ld), lane,
                           at(Bt.cost, 1), at(Bt.grad, nz), at(Bt.ceq, M.neq), at(Bt.jeq, (size_t)ph * NX * (2 * NX + NU)),
                           at(Bt.cineq, M.nineq + M.nue), at(Bt.jineq, (size_t)(M.nineq + M.nue) * nz), true,
                           Bt.params_b ? Bt.params_b + (size_t)b * Bt.nparams : nullptr);
    }
}
template <class Mdl>
__global__ __launch_bounds__(256) void nlmpc_evaluate(const NlmpcDev M, const NlmpcBatchDev Bt) { evaluate_body<Mdl>(M, Bt); }

// ---------------------------------------------------------------------------------------------------
// SQP
// ---------------------------------------------------------------------------------------------------
// The working set's Schur complement S = L L' as a Cholesky factor held on two levels: rows 0 .. NL-1 packed in LDS (row r at
// r (r + 1) / 2), rows NL .. in the workspace (row r at (r - NL) * ldg), 1 / L_rr of every row in LDS.  A vector of up to 128
// elements lives in two registers per lane (element e on lane e & 63).  Every step of a substitution is one broadcast from
// a register and one multiply-add per lane; over the LDS rows the operands of step k + 1 are requested before step k
// computes, over the workspace rows eight steps' operands are requested together, so the chain pays arithmetic latency
// (plus one memory latency per eight workspace rows).  Working sets beyond NL rows are the exception (the first iterations
// of a cold start); the LDS slice is sized for the rule.
struct Factor {
    double *Lp, *invd, *yb;      // LDS: packed rows, reciprocal diagonal [KW], a vector [KW]
    double *Lg;                  // workspace rows
    int NL, ldg;
};
__device__ __forceinline__ double read_lane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// L y = t on the leading n rows; t in (t0, t1), y returned in the same registers
template <bool TWO>
__device__ __forceinline__ void chol_forward(const Factor &F, int n, double &t0, double &t1, int lane)
{
    const double *Lp = F.Lp, *invd = F.invd;
    const int nl = min(n, F.NL);
    {
        const int o0 = lane * (lane + 1) / 2, o1 = (lane + 64) * (lane + 65) / 2;
        auto at0 = [&](int k) { return (lane > k && lane < nl) ? o0 + k : 0; };
        auto at1 = [&](int k) { return (lane + 64 > k && lane + 64 < nl) ? o1 + k : 0; };
        double a0 = Lp[at0(0)], a1 = Lp[at1(0)], id = invd[0];
        for (int k = 0; k < nl; ++k) {
            const int kn = min(k + 1, nl - 1);
            const double a0n = Lp[at0(kn)], a1n = Lp[at1(kn)], idn = invd[kn];
            const double yk = (k < 64 ? read_lane(t0, k) : read_lane(t1, k - 64)) * id;
            if (lane == k) t0 = yk;
            if (lane + 64 == k) t1 = yk;
            if (lane > k && lane < nl) t0 = fma(-a0, yk, t0);
            if (lane + 64 > k && lane + 64 < nl) t1 = fma(-a1, yk, t1);
            a0 = a0n; a1 = a1n; id = idn;
        }
    }
    if (!TWO || n <= F.NL) return;
    // workspace rows: first their part against y_0 .. y_{NL-1} (one dot product per row, each lane walks its own row) ...
    const int NL = F.NL;
    if (lane < NL) F.yb[lane] = t0;
    if (lane + 64 < NL) F.yb[lane + 64] = t1;
    nl_wave_sync();
    const bool g0 = lane >= NL && lane < n, g1 = lane + 64 >= NL && lane + 64 < n;
    const double *r0 = F.Lg + (size_t)(g0 ? lane - NL : 0) * F.ldg, *r1 = F.Lg + (size_t)(g1 ? lane + 64 - NL : 0) * F.ldg;
    if (g0) { const double s = gdot(r0, 1, F.yb, NL); t0 -= s; }
    if (g1) { const double s = gdot(r1, 1, F.yb, NL); t1 -= s; }
    // ... then the triangle among themselves
    for (int kb = NL; kb < n; kb += 8) {
        double a0[8], a1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(kb + u, n - 1);
            a0[u] = (g0 && lane > k) ? r0[k] : 0.0;
            a1[u] = (g1 && lane + 64 > k) ? r1[k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = kb + u;
            if (k < n) {
                const double yk = (k < 64 ? read_lane(t0, k) : read_lane(t1, k - 64)) * invd[k];
                if (lane == k) t0 = yk;
                if (lane + 64 == k) t1 = yk;
                t0 = fma(-a0[u], yk, t0);
                t1 = fma(-a1[u], yk, t1);
            }
        }
    }
}
// L' x = y on the leading n rows
template <bool TWO>
__device__ __forceinline__ void chol_backward(const Factor &F, int n, double &t0, double &t1, int lane)
{
    const double *Lp = F.Lp, *invd = F.invd;
    const int NL = F.NL, nl = min(n, NL);
    if (TWO && n > NL) {
        // workspace rows, last first: the triangle among themselves ...
        for (int kb = n - 1; kb >= NL; kb -= 8) {
            double a0[8], a1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = max(kb - u, NL);
                const double *rk = F.Lg + (size_t)(k - NL) * F.ldg;
                a0[u] = (lane >= NL && lane < k) ? rk[lane] : 0.0;
                a1[u] = (lane + 64 >= NL && lane + 64 < k) ? rk[lane + 64] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = kb - u;
                if (k >= NL) {
                    const double xk = (k < 64 ? read_lane(t0, k) : read_lane(t1, k - 64)) * invd[k];
                    if (lane == k) t0 = xk;
                    if (lane + 64 == k) t1 = xk;
                    t0 = fma(-a0[u], xk, t0);
                    t1 = fma(-a1[u], xk, t1);
                }
            }
        }
        // ... then what they take from the LDS rows' right-hand side: column j of the workspace rows against x_NL .. x_{n-1}
        if (lane >= NL && lane < n) F.yb[lane] = t0;
        if (lane + 64 >= NL && lane + 64 < n) F.yb[lane + 64] = t1;
        nl_wave_sync();
        if (lane < NL) { const double s = gdot(F.Lg + lane, (size_t)F.ldg, F.yb + NL, n - NL); t0 -= s; }
        if (lane + 64 < NL) { const double s = gdot(F.Lg + lane + 64, (size_t)F.ldg, F.yb + NL, n - NL); t1 -= s; }
    }
    auto at0 = [&](int k) { return lane < k ? k * (k + 1) / 2 + lane : 0; };
    auto at1 = [&](int k) { return lane + 64 < k ? k * (k + 1) / 2 + lane + 64 : 0; };
    double a0 = Lp[at0(nl - 1)], a1 = Lp[at1(nl - 1)], id = invd[nl - 1];
    for (int k = nl - 1; k >= 0; --k) {
        const int kn = max(k - 1, 0);
        const double a0n = Lp[at0(kn)], a1n = Lp[at1(kn)], idn = invd[kn];
        const double xk = (k < 64 ? read_lane(t0, k) : read_lane(t1, k - 64)) * id;
        if (lane == k) t0 = xk;
        if (lane + 64 == k) t1 = xk;
        if (lane < k) t0 = fma(-a0, xk, t0);
        if (lane + 64 < k) t1 = fma(-a1, xk, t1);
        a0 = a0n; a1 = a1n; id = idn;
    }
}
// row n of the factor from y = L^-1 (column n of S) and S_nn; false: the row depends on the ones above
template <bool TWO>
__device__ __forceinline__ bool chol_append(const Factor &F, int n, double y0, double y1, double snn, double dmax, int lane)
{
    const double d2 = snn - wave_sum((lane < n ? y0 * y0 : 0.0) + (lane + 64 < n ? y1 * y1 : 0.0));
    const bool ok = d2 > 1e-13 * dmax;
    const double dd = sqrt(ok ? d2 : 1e-13 * dmax + 1e-300);
    if (!TWO || n < F.NL) {
        if (lane < n) F.Lp[n * (n + 1) / 2 + lane] = y0;
        if (lane + 64 < n) F.Lp[n * (n + 1) / 2 + lane + 64] = y1;
        if (lane == 0) F.Lp[n * (n + 1) / 2 + n] = dd;
    } else {
        double *row = F.Lg + (size_t)(n - F.NL) * F.ldg;
        if (lane < n) row[lane] = y0;
        if (lane + 64 < n) row[lane + 64] = y1;
        if (lane == 0) row[n] = dd;
    }
    if (lane == 0) F.invd[n] = 1.0 / dd;
    nl_wave_sync();
    return ok;
}
// the factor of the leading n x n block of S (workspace, row stride ld).  The LDS rows by a right-looking elimination (the
// updates of a column step are independent of each other: LDS throughput, not a chain of substitutions), the workspace
// rows one at a time by substitution.  Called with the LDS arrays as offsets into the block's dynamic shared memory, so
// that the accesses stay LDS accesses across the call.
template <bool TWO>
__device__ __attribute__((noinline)) bool chol_factor(int lp_off, int invd_off, int yb_off, double *Lg, int NL, int ldg, const double *Sg, int ld, int n, int lane)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Factor F{smem + lp_off, smem + invd_off, smem + yb_off, Lg, NL, ldg};
    double *Lp = F.Lp;
    const int nl = min(n, NL);
    double dmax = 0.0;
    for (int r = lane; r < n; r += 64) dmax = fmax(dmax, Sg[r * ld + r]);
    dmax = wave_max(dmax);
    for (int e = lane; e < nl * (nl + 1) / 2; e += 64) {          // lower triangle of the LDS rows, packed
        int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (r * (r + 1) / 2 > e) --r;
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        Lp[e] = Sg[r * ld + (e - r * (r + 1) / 2)];
    }
    nl_wave_sync();
    bool ok = true;
    const int o0 = lane * (lane + 1) / 2, o1 = (lane + 64) * (lane + 65) / 2;
    for (int k = 0; k < nl; ++k) {
        const double dkk = Lp[k * (k + 1) / 2 + k];
        ok &= dkk > 1e-13 * dmax;
        const double dd = sqrt(dkk > 1e-13 * dmax ? dkk : 1e-13 * dmax + 1e-300), id = 1.0 / dd;
        double l0 = 0.0, l1 = 0.0;                                // this lane's rows' entries of column k
        if (lane > k && lane < nl) { l0 = Lp[o0 + k] * id; Lp[o0 + k] = l0; }
        if (lane + 64 > k && lane + 64 < nl) { l1 = Lp[o1 + k] * id; Lp[o1 + k] = l1; }
        nl_wave_sync();
        for (int j = k + 1; j < nl; ++j) {
            const double ljk = Lp[j * (j + 1) / 2 + k];
            if (lane >= j && lane < nl) Lp[o0 + j] = fma(-l0, ljk, Lp[o0 + j]);
            if (lane + 64 >= j && lane + 64 < nl) Lp[o1 + j] = fma(-l1, ljk, Lp[o1 + j]);
        }
        if (lane == 0) { Lp[k * (k + 1) / 2 + k] = dd; F.invd[k] = id; }
        nl_wave_sync();
    }
    if constexpr (TWO) {
        for (int i = nl; i < n; ++i) {
            double t0 = lane < i ? Sg[i * ld + lane] : 0.0, t1 = lane + 64 < i ? Sg[i * ld + lane + 64] : 0.0;
            const double sii = Sg[i * ld + i];
            chol_forward<TWO>(F, i, t0, t1, lane);
            ok &= chol_append<TWO>(F, i, t0, t1, sii, dmax, lane);
        }
    }
    return ok;
}

// TWO: working sets may outgrow the LDS factor (M.kw > M.nl); otherwise that code is left out
template <class Mdl, bool TWO = true, bool BLK = false>
__device__ __forceinline__ void sqp_body(const NlmpcDev &M, const NlmpcSolveDev &S)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU, W = 2 * NX + NU;
    constexpr int GU = Mdl::NX >= 12 ? 32 : 8;            // loads in flight in the long dot products (one wavefront per SIMD: 512 registers)
    constexpr int WB = GU == 32 ? 8 : 2, SB = GU == 32 ? 4 : 1;   // rows / entries in flight in the warm start of the sub-problem
    constexpr bool MAYBE_CT = Mdl::CONTINUOUS;            // hook models: true, the run-time flag decides
    const bool CT = is_ct<Mdl>(M);
    const int KW = M.kw, SLD = KW + 1;                     // working-set capacity of this controller
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    const int ph = M.ph, ch = M.ch, nz = M.nz, mi = M.nineq, m = M.nineq + M.nue, nzu = M.nzu, nr = M.nr, nxs = ph * NX;
    // user rows: [0, mi) inequalities g <= 0, [mi, m) equalities h = 0; then the bounds
    const int mt = m + M.nbnd;                           // sub-problem rows: user inequalities, then the finite bounds on z
    const int nq = S.hard ? nzu : nr;                     // variables of the sub-problem (slack only when soft)
    const int mld = (mt + 1) & ~1;
    const Scale sc(M);
    double *Xs = smem + (size_t)wave * M.lds_per_wave;
    double *Us = Xs + (ph + 1) * NX;
    double *Ys = Us + (ph + 1) * NU;                      // (ph+1) x NY, hook models with an output function only (else empty)
    double *tq = Ys + ((Mdl::VECTOR_HOOKS && M.has_output) ? (ph + 1) * Mdl::NY : 0);   // KW
    double *uq = tq + KW;                                 // KW  multipliers of the working set
    double *wq = uq + KW;                                 // KW  row numbers (as doubles)
    double *sgq = wq + KW;                                // KW  orientation of the row in the working set (+1; -1 for an equality entered from below)
    double *invd = sgq + KW;                              // KW  reciprocal diagonal of the working set's factor
    double *ybuf = invd + KW;                             // KW  a vector of the substitutions (working sets beyond NL rows)
    double *s1v = ybuf + KW;                              // mt  first entry of every sub-problem row in the sparse form ...
    int *s1m = reinterpret_cast<int *>(s1v + mt);         // mt  ... its index (low 16 bits) and the row's entry count (high bits; kSpDense: not sparse)
    double *v0 = s1v + mt + ((mt + 1) >> 1);              // 4 vectors of nr
    double *v1 = v0 + nr, *v2 = v1 + nr, *v3 = v2 + nr;
    // the rest of the slice is the step (dXs, dUs), the transcription's scratch (Jm) and the condensing's (aug); none of
    // them lives across the sub-problem, whose factor Lp takes the whole tail: rows 0 .. NL-1 of the working set
    double *dXs = v3 + nr;
    double *dUs = dXs + (ph + 1) * NX;
    double *Jm = dUs + (ph + 1) * NU;                     // ph x NU
    double *aug = Jm + ph * NU;                           // NX x 2NX
    double *Lp = dXs;
    const int NL = M.nl;

    for (int b = blockIdx.x * wpb + wave; b < S.batch; b += gridDim.x * wpb) {
        double *w = S.ws + (size_t)b * M.ws.total;
        double *z = w + M.ws.z, *d = w + M.ws.d, *g = w + M.ws.g, *c = w + M.ws.c, *jeq = w + M.ws.jeq, *gin = w + M.ws.gin,
               *jin = w + M.ws.jin, *r = w + M.ws.r, *phi = w + M.ws.phi, *einv = w + M.ws.einv, *gr = w + M.ws.gr,
               *art = w + M.ws.art, *br = w + M.ws.br, *hinv = w + M.ws.hinv, *mu = w + M.ws.mu, *glold = w + M.ws.glold,
               *sv = w + M.ws.s, *p = w + M.ws.p, *qn = w + M.ws.qn, *qv = w + M.ws.qv, *Ssm = w + M.ws.qs, *Sbig = w + M.ws.qs2, *scal = w + M.ws.scal,
               *lamw = w + M.ws.lamw, *hk = w + M.ws.hook, *spv = w + M.ws.sp;
        int *spi = reinterpret_cast<int *>(spv + (size_t)mt * kNlSparse);   // entries 1.. of the sub-problem's sparse rows (entry 0: LDS)
        if constexpr (BLK) {
            // the dynamics blocks and the sweeps' right-hand sides live in the LDS slice (nlmpc_plan)
            jeq = Xs + M.lds_blocks; einv = jeq + ph * NX * W; c = einv + ph * NX * NX; lamw = c + nxs; p = lamw + nxs;
        }
        const double *x0 = S.x0 + (size_t)b * NX, *u0 = S.u0 + (size_t)b * NU;
        const double *prm = S.params_b ? S.params_b + (size_t)b * S.nparams : M.params;     // per-instance model parameters (built-in systems)

        // ---- initial guess (NLOptimizer.hpp:431-510): cold = (x0, u0) replicated; warm = previous solution shifted one step
        if (S.z_warm) {
            const double *zw = S.z_warm + (size_t)b * nz;
            for (int k = lane; k < nxs; k += 64) { const int i = k / NX; z[k] = zw[i == ph - 1 ? k : k + NX]; }
            for (int k = lane; k < nzu; k += 64) {
                const int bl = k / NU, j = k - bl * NU;
                const int step = min(bl + 1, ph - 1);                       // first step of the block, shifted by one
                z[nxs + k] = zw[nxs + min(step, ch - 1) * NU + j];
            }
            if (lane == 0) z[nz - 1] = zw[nz - 1];
        } else {
            for (int k = lane; k < nxs; k += 64) z[k] = x0[k % NX];
            for (int k = lane; k < nzu; k += 64) z[nxs + k] = u0[k % NU];
            if (lane == 0) z[nz - 1] = 0.0;
        }
        if (!S.keep_curvature)                              // otherwise: the estimate the previous tick's solve left in the workspace
            for (int k = lane; k < nr * nr; k += 64) hinv[k] = (k / nr == k % nr) ? 1.0 : 0.0;
        for (int k = lane; k < mt; k += 64) mu[k] = 0.0;
        // NLOptimizer::fixOptimalSolution (NLOptimizer.hpp:705-716): a start outside the bounds goes to (ub - lb) / 2 (sic)
        for (int k = lane; k < nz; k += 64) {
            const double lo = M.zlb[k], hi = M.zub[k];
            if (z[k] < lo || z[k] > hi) z[k] = (hi - lo) / 2.0;
        }
        nl_wave_sync();

        // user constraints read few states: one bit per (row, state) entry of d g / d x that can hold anything, from the
        // structure the model declares (the finite differences leave exact zeros everywhere else)
        // the sparse form of row k: entry count (-1: not sparse), and entry j -- the first from LDS, further ones from the workspace
        auto sp_count = [&](int k) { const int mw = s1m[k]; return mw == kSpDense ? -1 : mw >> 16; };
        auto sp_index = [&](int k, int j) { return j == 0 ? (s1m[k] & 0xffff) : spi[k * kNlSparse + j]; };
        auto sp_value = [&](int k, int j) { return j == 0 ? s1v[k] : spv[k * kNlSparse + j]; };
        auto structure_word = [&](int e0, int nchunk) {
            const int k = e0 / nchunk, col = (e0 - k * nchunk) * 64 + lane;
            return __ballot(col < nxs && (k < mi ? Mdl::ineq_reads_x(k, col / NX + 1) : Mdl::eq_reads_x(k - mi, col / NX + 1)));
        };
        // Phi = d x / d p (the condensed sensitivities, [ph nx x ch nu]) is only needed where a row of the sub-problem reads a
        // state: a user constraint that does, or a bound on a state.  Without such rows the reduced gradient comes from one
        // backward sweep, the state step from one forward sweep, and Phi is never formed.
        bool needs_phi;
        {
            const int nchunk = (nxs + 63) >> 6;
            bool any = false;
            for (int e0 = 0; e0 < m * nchunk && !any; ++e0) any = structure_word(e0, nchunk) != 0ull;
            for (int kb = lane; kb < M.nbnd; kb += 64) any |= M.bnd_idx[kb] < nxs;
            needs_phi = __ballot(any) != 0ull;
        }
        // the sparse form of a row pays where the rows of the reduced problem are long, or where every row is sparse (nothing
        // reads a state); next to short dense rows (one load per lane) it only adds dependent loads (measured: configs 3, 5 and the 6-oscillator case)
        const bool sparse_rows = nq > 64 || !needs_phi;
        double nu_pen = 0.0, a_prev = 0.0;
        bool have_old = false;
        int resets = 0;
        int nw_keep = 0;                                    // rows active at the end of the previous sub-problem (in wq)
        // Run-time guard for the failure described below: every phase boundary checks that the whole wavefront arrived (the wave-level
        // reductions and the hooks behind pointers rely on it); an instance that ever lost lanes is reported with nlopt's FORCED_STOP code (-5: nothing else produces it), never as a result.
        bool exec_full = true;
#ifdef MPCX_NL_STATS
        // Statistics for tools/nlmpc_phases.py (debug_workspace), compiled in only with -DMPCX_NL_STATS (make stats) -- the
        // counters are live across the whole iteration and cost the product kernels registers they do not have:
        // per-phase cycle counts; sub-problem: steps, inner passes, rows at the end (sum, max), rows kept, rows shed at the
        // warm start, cycles of the warm start, cycles of the factorisations
        long long cyc[6] = {0, 0, 0, 0, 0, 0}, tstamp = __builtin_readcyclecounter();
        long long qst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto lap = [&](int ph_) { const long long now = __builtin_readcyclecounter(); cyc[ph_] += now - tstamp; tstamp = now; exec_full &= __builtin_amdgcn_read_exec() == ~0ull; };
#define MPCX_STAT(x) x
#else
        // Phase boundaries keep a scheduling barrier in the product build.  History: a build of round 2 without one (an empty `lap`) reached
        // the condensing phase of the single-level 6-oscillator instantiation with lanes missing from EXEC (ROCm debug agent), and ran
        // correctly with the cycle counters, a memory clobber or this barrier.  Round 4 looked again (DESIGN.md section 9-2,
        // tools/micro/exec_mask.sh, profiles/r04_probe_exec_mask.txt): on the present source the two builds compile to the same ISA up to
        // three reordered scalar instructions, the build without the barrier passes the whole NLMPC suite and the shape sweep, and the guard
        // has never tripped -- the failure belonged to a source state that no longer exists and cannot be reduced further.  The barrier costs
        // nothing (same ISA) and stays; so does the guard, with a status code of its own (-5).
#ifdef MPCX_NL_NO_LAP_BARRIER                                   // (tools/micro/exec_mask.sh: the build that shows the failure)
        auto lap = [&](int) { exec_full &= __builtin_amdgcn_read_exec() == ~0ull; };
#else
        auto lap = [&](int) { __builtin_amdgcn_sched_barrier(0); exec_full &= __builtin_amdgcn_read_exec() == ~0ull; };
#endif
#define MPCX_STAT(x)
#endif
        double f_prev = 0, step_l1 = 0, z_l1 = 0, step_max = 0;       // the last accepted step, for nlopt's stopping rules
        bool stepped = false;
        const bool tol_on = S.ftol_abs > 0 || S.ftol_rel > 0 || S.xtol_abs > 0 || S.xtol_rel > 0;
        int it = 0, code = 5;       // nlopt codes: 3 FTOL_REACHED, 4 XTOL_REACHED, 5 MAXEVAL_REACHED, -1 FAILURE, -3 OUT_OF_MEMORY, -4 ROUNDOFF_LIMITED
        // One call site for the transcription (the solver has to stay inside the 64 KB instruction cache, and three inlined
        // copies of it do not): every pass of the loop starts with the evaluation at the current point -- everything on the
        // first pass and after a step, values only after the last, converged step.
        bool first_eval = true, final_eval = false;
        for (;;) {
            if (!first_eval) lap(4);
            eval_instance<Mdl, BLK>(M, z, x0, Xs, Us, Jm, Ys, hk, lane, scal, final_eval ? nullptr : g, c, final_eval ? nullptr : jeq, gin,
                               final_eval ? nullptr : jin, first_eval, prm);     // structural zeros are written once
            if (!first_eval) lap(5); else { MPCX_STAT(tstamp = __builtin_readcyclecounter();) }
            first_eval = false;
            if (final_eval) break;
            if (stepped && tol_on) {
                // nlopt's stopping rules (set_ftol_rel / set_ftol_abs / set_xtol_rel / set_xtol_abs, NLOptimizer.hpp:135-138;
                // nlopt_stop_ftol and nlopt_stop_x with the unit weights of :140), applied as SLSQP applies them: to a step
                // that ended at a feasible point
                double vmax = 0;
                for (int k = lane; k < nxs; k += 64) vmax = fmax(vmax, fabs(c[k]));
                for (int k = lane; k < m; k += 64) vmax = fmax(vmax, k < mi ? gin[k] : fabs(gin[k]));
                vmax = wave_max(vmax);
                if (vmax <= S.tol_con) {
                    const double fn = scal[0], df = fabs(fn - f_prev);
                    const bool ft = (S.ftol_abs > 0 && df < S.ftol_abs) ||
                                    (S.ftol_rel > 0 && (df < S.ftol_rel * 0.5 * (fabs(fn) + fabs(f_prev)) || fn == f_prev));
                    const bool xt = (S.xtol_rel > 0 && step_l1 <= S.xtol_rel * z_l1) || (S.xtol_abs > 0 && step_max < S.xtol_abs);
                    if (ft) { code = 3; break; }                 // NLOPT_FTOL_REACHED
                    if (xt) { code = 4; break; }                 // NLOPT_XTOL_REACHED
                }
            }
            stepped = false;
            if (it >= S.max_iter) break;
            // ---- condensing: inverses of E_i = dc_i/dx_{i+1} (identity for one-step models)
            if (CT) {
                // Gauss-Jordan on [E | I] with one column per lane held in registers: the pivot column's entries reach the
                // other lanes by shuffles, so a pivot step is NX shuffles and NX FMAs with no LDS traffic and no barrier; 2 NX
                // lanes serve one step, the wavefront inverts 64 / (2 NX) steps at a time.  Partial pivoting as before.
                constexpr int GW = 2 * NX, G = 64 / GW;
                const int g = lane / GW, cidx = lane - g * GW, base = g * GW;
                for (int i0 = 0; i0 < ph; i0 += G) {
                    const int i = i0 + g;
                    const bool live = g < G && i < ph;
                    double col[NX];
#pragma unroll
                    for (int a = 0; a < NX; ++a)
                        col[a] = (live && cidx < NX) ? jeq[(size_t)i * NX * W + a * W + NX + cidx] : ((cidx < NX ? cidx : cidx - NX) == a ? 1.0 : 0.0);
#pragma unroll
                    for (int k = 0; k < NX; ++k) {
                        int pr = k;
                        double best = fabs(col[k]);
#pragma unroll
                        for (int a = k + 1; a < NX; ++a) { const double v = fabs(col[a]); if (v > best) { best = v; pr = a; } }
                        pr = __shfl(pr, base + k);                     // the pivot column's choice
                        double cp = col[k];
#pragma unroll
                        for (int a = k + 1; a < NX; ++a) if (a == pr) { cp = col[a]; col[a] = col[k]; }
                        col[k] = cp;                                    // rows k and pr swapped in every column
                        const double piv = __shfl(col[k], base + k);
                        double mlt[NX];
#pragma unroll
                        for (int a = 0; a < NX; ++a) mlt[a] = __shfl(col[a], base + k);     // pivot column, before anyone updates
                        const double cs = col[k] / piv;
#pragma unroll
                        for (int a = 0; a < NX; ++a) col[a] = a == k ? cs : fma(-mlt[a], cs, col[a]);
                    }
                    if (live && cidx >= NX)
#pragma unroll
                        for (int a = 0; a < NX; ++a) einv[(size_t)i * NX * NX + a * NX + (cidx - NX)] = col[a];
                }
                nl_wave_sync();
            }
            // forward sweep, one column per lane: dx = r + Phi p with dx_0 = 0.  A_i = dc_i/dx_i (and E_i^-1) are the same for
            // every lane: they are staged in LDS, the next step's share already in flight while this step computes.
            // mode 0: every column (Phi and r); 1: r alone; 2: the state step r + Phi p itself (into d), p applied on the way
            // One column alone (modes 1 and 2) is a chain of ph small matrix-vector products: four lanes share a row, so that a
            // step is two rounds of (a few multiply-adds, a two-step butterfly, one LDS exchange) instead of one lane's 2 nx^2.
            auto sweep_column = [&](const int mode) {
                constexpr int CH = (NX + 3) / 4, CU = (NU + 3) / 4;
                const int a = lane >> 2, part = lane & 3;
                const bool rowlive = a < NX;
                double *vs = aug, *ts = aug + NX;
                double *out = mode == 2 ? d : r;
                double an[CH], en[CH], bn[CU], cn = 0.0;
                auto fetch = [&](int i) {
                    const double *Jb = jeq + (size_t)i * NX * W + (size_t)(rowlive ? a : 0) * W;
                    const double *Ei = einv + (size_t)i * NX * NX + (size_t)(rowlive ? a : 0) * NX;
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        const int bb = min(part + 4 * u, NX - 1);
                        an[u] = part + 4 * u < NX ? Jb[bb] : 0.0;
                        en[u] = (CT && part + 4 * u < NX) ? Ei[bb] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < CU; ++u) bn[u] = (mode == 2 && part + 4 * u < NU) ? Jb[2 * NX + min(part + 4 * u, NU - 1)] : 0.0;
                    cn = part == 0 ? c[i * NX + (rowlive ? a : 0)] : 0.0;
                };
                if (lane < NX) vs[lane] = 0.0;
                fetch(0);
                nl_wave_sync();
                for (int i = 0; i < ph; ++i) {
                    double ac[CH], ec[CH], s2 = cn;
#pragma unroll
                    for (int u = 0; u < CH; ++u) { ac[u] = an[u]; ec[u] = en[u]; }
                    if (mode == 2) {
                        const double *pb = p + min(i, ch - 1) * NU;
#pragma unroll
                        for (int u = 0; u < CU; ++u) s2 = fma(bn[u], pb[min(part + 4 * u, NU - 1)], s2);
                    }
                    if (i + 1 < ph) fetch(i + 1);
#pragma unroll
                    for (int u = 0; u < CH; ++u) s2 = fma(ac[u], vs[min(part + 4 * u, NX - 1)], s2);
                    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
                    double va;
                    if (CT) {
                        if (rowlive && part == 0) ts[a] = s2;
                        nl_wave_sync();
                        double s3 = 0.0;
#pragma unroll
                        for (int u = 0; u < CH; ++u) s3 = fma(ec[u], ts[min(part + 4 * u, NX - 1)], s3);
                        s3 += __shfl_xor(s3, 1); s3 += __shfl_xor(s3, 2);
                        va = -s3;
                    } else {
                        va = -s2;
                    }
                    nl_wave_sync();                                     // every lane has read v_i
                    if (rowlive && part == 0) { vs[a] = va; out[i * NX + a] = va; }
                    nl_wave_sync();
                }
            };
            auto sweep = [&](const int mode) {
            if (NX >= 8 && NX <= 16 && mode != 0) { sweep_column(mode); return; }
            if constexpr (NX >= 8) {
            for (int q0 = mode == 0 ? 0 : (nzu & ~63); q0 <= nzu; q0 += 64) {
                    const int q = q0 + lane;
                    const bool qlive = mode == 0 ? q <= nzu : q == nzu;
                    const int bq = q / NU, jq = q - bq * NU;
                    constexpr int NST = MAYBE_CT ? 2 * NX * NX : NX * NX;     // staged doubles per step: A | Einv
                    constexpr int PL = (NST + 63) / 64;
                    double *As = aug, *Es = aug + NX * NX;
                    double nxt[PL], rhs[NX], rhs_n[NX], v[NX], t[NX];
                    auto fetch = [&](int i, double (&dst)[PL], double (&rh)[NX]) {
                        const double *Jb = jeq + (size_t)i * NX * W;
    #pragma unroll
                        for (int u = 0; u < PL; ++u) {
                            const int e = min(lane + 64 * u, NST - 1);
                            dst[u] = e < NX * NX ? Jb[(e / NX) * W + e % NX] : einv[(size_t)i * NX * NX + (e - NX * NX)];
                        }
    #pragma unroll
                        for (int a = 0; a < NX; ++a)
                            rh[a] = !qlive ? 0.0 : (q == nzu ? c[i * NX + a] : (min(i, ch - 1) == bq ? Jb[a * W + 2 * NX + jq] : 0.0));
                        if (mode == 2 && q == nzu) {
                            const double *pb = p + min(i, ch - 1) * NU;
    #pragma unroll
                            for (int a = 0; a < NX; ++a)
    #pragma unroll
                                for (int j = 0; j < NU; ++j) rh[a] = fma(Jb[a * W + 2 * NX + j], pb[j], rh[a]);
                        }
                    };
    #pragma unroll
                    for (int a = 0; a < NX; ++a) v[a] = 0.0;
                    fetch(0, nxt, rhs_n);
                    for (int i = 0; i < ph; ++i) {
    #pragma unroll
                        for (int u = 0; u < PL; ++u) if (lane + 64 * u < NST) aug[lane + 64 * u] = nxt[u];
    #pragma unroll
                        for (int a = 0; a < NX; ++a) rhs[a] = rhs_n[a];
                        nl_wave_sync();
                        if (i + 1 < ph) fetch(i + 1, nxt, rhs_n);
    #pragma unroll
                        for (int a = 0; a < NX; ++a) {
                            double s2 = rhs[a];
    #pragma unroll
                            for (int bb = 0; bb < NX; ++bb) s2 = fma(As[a * NX + bb], v[bb], s2);
                            t[a] = s2;
                        }
                        if (CT) {
    #pragma unroll
                            for (int a = 0; a < NX; ++a) {
                                double s2 = 0;
    #pragma unroll
                                for (int bb = 0; bb < NX; ++bb) s2 = fma(Es[a * NX + bb], t[bb], s2);
                                v[a] = -s2;
                            }
                        } else {
    #pragma unroll
                            for (int a = 0; a < NX; ++a) v[a] = -t[a];
                        }
                        if (qlive) {
                            if (q == nzu) for (int a = 0; a < NX; ++a) (mode == 2 ? d : r)[i * NX + a] = v[a];
                            else for (int a = 0; a < NX; ++a) phi[(size_t)(i * NX + a) * nzu + q] = v[a];
                        }
                        nl_wave_sync();
                    }
                }
            } else {
                for (int q = lane; q <= nzu; q += 64) {
                    if (mode != 0 && q != nzu) continue;
                    // small blocks come straight from L2, the next step's already requested while this one computes (the
                    // stores to Phi in between keep the compiler from moving the loads up by itself)
                    double v[NX], t[NX], An[NX * NX], rn[NX], En[MAYBE_CT ? NX * NX : 1];
                    for (int a = 0; a < NX; ++a) v[a] = 0.0;
                    const int bq = q / NU, jq = q - bq * NU;
                    auto fetch = [&](int i) {
                        const double *Jb = jeq + (size_t)i * NX * W;
#pragma unroll
                        for (int a = 0; a < NX; ++a) {
                            rn[a] = q == nzu ? c[i * NX + a] : (min(i, ch - 1) == bq ? Jb[a * W + 2 * NX + jq] : 0.0);
                            if (mode == 2 && q == nzu)
#pragma unroll
                                for (int j = 0; j < NU; ++j) rn[a] = fma(Jb[a * W + 2 * NX + j], p[min(i, ch - 1) * NU + j], rn[a]);
#pragma unroll
                            for (int bb = 0; bb < NX; ++bb) An[a * NX + bb] = Jb[a * W + bb];
                        }
                        if (CT) {
                            const double *Ei = einv + (size_t)i * NX * NX;
#pragma unroll
                            for (int e2 = 0; e2 < NX * NX; ++e2) En[e2] = Ei[e2];
                        }
                    };
                    fetch(0);
                    for (int i = 0; i < ph; ++i) {
                        double Ac[NX * NX], rc[NX], Ec[MAYBE_CT ? NX * NX : 1];
#pragma unroll
                        for (int e2 = 0; e2 < NX * NX; ++e2) { Ac[e2] = An[e2]; if (MAYBE_CT) Ec[e2] = En[e2]; }
#pragma unroll
                        for (int a = 0; a < NX; ++a) rc[a] = rn[a];
                        if (i + 1 < ph) fetch(i + 1);
#pragma unroll
                        for (int a = 0; a < NX; ++a) {
                            double s2 = rc[a];
#pragma unroll
                            for (int bb = 0; bb < NX; ++bb) s2 = fma(Ac[a * NX + bb], v[bb], s2);
                            t[a] = s2;
                        }
                        if (CT) {
#pragma unroll
                            for (int a = 0; a < NX; ++a) {
                                double s2 = 0;
#pragma unroll
                                for (int bb = 0; bb < NX; ++bb) s2 = fma(Ec[a * NX + bb], t[bb], s2);
                                v[a] = -s2;
                            }
                        } else {
#pragma unroll
                            for (int a = 0; a < NX; ++a) v[a] = -t[a];
                        }
                        if (q == nzu) for (int a = 0; a < NX; ++a) (mode == 2 ? d : r)[i * NX + a] = v[a];
                        else for (int a = 0; a < NX; ++a) phi[(size_t)(i * NX + a) * nzu + q] = v[a];
                    }
                }
            }
            nl_wave_sync();
            };
            sweep(needs_phi ? 0 : 1);
            lap(0);
            // Jx' lam = -lamw by a backward sweep over the blocks (Jx: the block-bidiagonal Jacobian of the dynamics defects wrt
            // the states); returns this lane's share of max |lam|, optionally leaves lam in lamw
            auto backsolve = [&](const bool store) {
            double lam_max = 0;
            {
                double *tl = aug, *ln = aug + NX;                  // t and lam_{i+1}
                // the operands of step i-1 are requested before step i computes: the sweep is a chain of ph dependent steps and
                // would otherwise pay a memory latency in each
                double an[NX], en[NX], wn = 0.0;
                auto fetch = [&](int i) {
                    if (lane < NX) {
                        wn = lamw[i * NX + lane];
                        const double *Jb = jeq + (size_t)min(i + 1, ph - 1) * NX * W;
#pragma unroll
                        for (int bb = 0; bb < NX; ++bb) an[bb] = i + 1 < ph ? Jb[bb * W + lane] : 0.0;
                        if (CT) {
                            const double *Ei = einv + (size_t)i * NX * NX;
#pragma unroll
                            for (int bb = 0; bb < NX; ++bb) en[bb] = Ei[bb * NX + lane];
                        }
                    }
                };
                if (lane < NX) ln[lane] = 0.0;
                fetch(ph - 1);
                nl_wave_sync();
                for (int i = ph - 1; i >= 0; --i) {
                    double ac[NX], ec[NX];
                    const double wc = wn;
#pragma unroll
                    for (int bb = 0; bb < NX; ++bb) { ac[bb] = an[bb]; ec[bb] = en[bb]; }
                    if (i > 0) fetch(i - 1);
                    if (lane < NX) {
                        double s2 = wc;
#pragma unroll
                        for (int bb = 0; bb < NX; ++bb) s2 = fma(ac[bb], ln[bb], s2);
                        tl[lane] = s2;
                    }
                    nl_wave_sync();
                    if (lane < NX) {
                        double lam;
                        if (CT) {
                            lam = 0;
#pragma unroll
                            for (int bb = 0; bb < NX; ++bb) lam = fma(-ec[bb], tl[bb], lam);
                        } else {
                            lam = -tl[lane];
                        }
                        ln[lane] = lam;
                        if (store) lamw[i * NX + lane] = lam;
                        lam_max = fmax(lam_max, fabs(lam));
                    }
                    nl_wave_sync();
                }
            }
            nl_wave_sync();
            return lam_max;
            };
            // The sub-problem's rows in the sparse form (index, value lists of at most kNlSparse entries; spn < 0: not sparse, the
            // row is column k of art).  A user row that reads no state is the row of the user Jacobian's input part (and the
            // slack column) as it is: one lane scans one row -- the finite differences leave exact zeros -- and only a row with
            // more entries is needed in art at all.  A bound on an input is one entry.
            auto scan_user_rows = [&](const unsigned long long *fm, const int nchunk) {      // fm: structure words (null: no row reads a state)
                for (int k0 = 0; k0 < m; k0 += 64) {
                    const int k = k0 + lane;
                    const bool live = k < m;
                    bool xfree = sparse_rows;
                    if (fm) for (int cb = 0; cb < nchunk; ++cb) xfree &= fm[(live ? k : 0) * nchunk + cb] == 0ull;
                    const double *jr = jin + (size_t)(live ? k : 0) * nz;
                    int cnt = 0, i0 = 0, i1 = 0, i2 = 0, i3 = 0;
                    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
                    if (xfree) {
                        for (int q0 = 0; q0 < nq; q0 += 8) {
                            double av[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { const int q = min(q0 + u, nq - 1); av[u] = jr[q == nzu ? nz - 1 : nxs + q]; }
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                if (q0 + u < nq && av[u] != 0.0) {
                                    if (cnt == 0) { i0 = q0 + u; e0 = av[u]; } else if (cnt == 1) { i1 = q0 + u; e1 = av[u]; }
                                    else if (cnt == 2) { i2 = q0 + u; e2 = av[u]; } else if (cnt == 3) { i3 = q0 + u; e3 = av[u]; }
                                    ++cnt;
                                }
                            }
                        }
                    }
                    const bool sp = xfree && cnt <= kNlSparse;
                    if (live) {
                        s1v[k] = sp ? e0 : 0.0; s1m[k] = sp ? (cnt << 16) | i0 : kSpDense;
                        spi[k * kNlSparse] = sp ? i0 : 0; spi[k * kNlSparse + 1] = sp ? i1 : 0; spi[k * kNlSparse + 2] = sp ? i2 : 0; spi[k * kNlSparse + 3] = sp ? i3 : 0;
                        spv[k * kNlSparse] = sp ? e0 : 0.0; spv[k * kNlSparse + 1] = sp ? e1 : 0.0; spv[k * kNlSparse + 2] = sp ? e2 : 0.0; spv[k * kNlSparse + 3] = sp ? e3 : 0.0;
                        if (!fm) {
                            br[k] = gin[k];
                            if (!sp) for (int q = 0; q < nr; ++q) art[(size_t)q * mld + k] = jr[q == nzu ? nz - 1 : nxs + q];
                        }
                    }
                }
            };
            auto input_bounds = [&]() {
                for (int kb = lane; kb < M.nbnd; kb += 64) {
                    const int zi = M.bnd_idx[kb], k = m + kb;
                    if (zi < nxs || !sparse_rows) { s1v[k] = 0.0; s1m[k] = kSpDense; continue; } // a bound on a state: a row of Phi (dense_bounds)
                    const double sg = M.bnd_sign[kb];
                    s1v[k] = sg; s1m[k] = (1 << 16) | (zi - nxs);
                    spi[k * kNlSparse] = zi - nxs; spi[k * kNlSparse + 1] = 0; spi[k * kNlSparse + 2] = 0; spi[k * kNlSparse + 3] = 0;
                    spv[k * kNlSparse] = sg; spv[k * kNlSparse + 1] = 0.0; spv[k * kNlSparse + 2] = 0.0; spv[k * kNlSparse + 3] = 0.0;
                    br[k] = sg * (z[zi] - M.bnd_val[kb]);
                }
            };
            auto dense_bounds = [&]() {                                  // the bounds kept as columns of art, 64 descriptors at a time
                for (int kb0 = 0; kb0 < M.nbnd; kb0 += 64) {
                    const int kbl = min(kb0 + lane, M.nbnd - 1);
                    const int zl = M.bnd_idx[kbl];
                    const double sgl = M.bnd_sign[kbl], vall = M.bnd_val[kbl];
                    unsigned long long mk = __ballot(kb0 + lane < M.nbnd && (zl < nxs || !sparse_rows));
                    while (mk) {
                        const int l = (int)__builtin_ctzll(mk);
                        mk &= mk - 1;
                        const int zi = __builtin_amdgcn_readlane(zl, l), kb = kb0 + l;
                        const double sg = read_lane(sgl, l), val = read_lane(vall, l);
                        for (int q = lane; q < nr; q += 64)
                            art[(size_t)q * mld + m + kb] = zi < nxs ? (q < nzu ? sg * phi[(size_t)zi * nzu + q] : 0.0) : (q == zi - nxs ? sg : 0.0);
                        if (lane == 0) br[m + kb] = sg * (z[zi] + (zi < nxs ? r[zi] : 0.0) - val);
                    }
                }
            };
            double lam_dyn = 0;                                         // this lane's share of the largest dynamics multiplier (sweep below)
            if (needs_phi) {
                // reduced gradient, reduced inequality rows (transposed: art[q][k]) and their offsets
                for (int q = lane; q < nr; q += 64) {
                    if (q == nzu) { gr[q] = g[nz - 1]; continue; }
                    const double s = g[nxs + q] + gdot2(phi + q, nzu, g, 1, nxs);
                    gr[q] = s;
                }
                const int nchunk = (nxs + 63) >> 6;
                // the structure words, in the part of the LDS slice that is idle between the condensing and the sub-problem
                unsigned long long *fmask = reinterpret_cast<unsigned long long *>(dXs);
                for (int e0 = 0; e0 < m * nchunk; ++e0) {
                    const unsigned long long bal = structure_word(e0, nchunk);
                    if (lane == 0) fmask[e0] = bal;
                }
                nl_wave_sync();
                for (int q = lane; q < nr; q += 64) {
                    const bool realq = q < nzu;
                    const size_t qq = realq ? q : 0, ucol = q == nzu ? nz - 1 : nxs + q;
                    if (nchunk <= 2) {
                        // four rows at a time, up to four entries of each gathered first so that their loads are in flight together
                        for (int k0 = 0; k0 < m; k0 += 4) {
                            int c4[4][4];
                            unsigned long long rest0[4], rest1[4];
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const int k = min(k0 + rr, m - 1);
                                unsigned long long m0 = fmask[k * nchunk], m1 = nchunk > 1 ? fmask[k * nchunk + 1] : 0ull;
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    int cc = -1;
                                    if (m0) { cc = (int)__builtin_ctzll(m0); m0 &= m0 - 1; }
                                    else if (m1) { cc = 64 + (int)__builtin_ctzll(m1); m1 &= m1 - 1; }
                                    c4[rr][u] = cc;
                                }
                                rest0[rr] = m0; rest1[rr] = m1;
                            }
                            double a0[4], jv[4][4], pv[4][4];
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const size_t k = min(k0 + rr, m - 1);
                                a0[rr] = jin[k * nz + ucol];
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const size_t cc = c4[rr][u] < 0 ? 0 : c4[rr][u];
                                    jv[rr][u] = jin[k * nz + cc];
                                    pv[rr][u] = phi[cc * nzu + qq];
                                }
                            }
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                double acc = a0[rr];
#pragma unroll
                                for (int u = 0; u < 4; ++u) if (realq && c4[rr][u] >= 0) acc = fma(jv[rr][u], pv[rr][u], acc);
                                const size_t k = min(k0 + rr, m - 1);
                                unsigned long long m0 = rest0[rr], m1 = rest1[rr];
                                while (realq && (m0 | m1)) {                   // rows with more than four entries
                                    int cc;
                                    if (m0) { cc = (int)__builtin_ctzll(m0); m0 &= m0 - 1; } else { cc = 64 + (int)__builtin_ctzll(m1); m1 &= m1 - 1; }
                                    acc = fma(jin[k * nz + cc], phi[(size_t)cc * nzu + qq], acc);
                                }
                                if (k0 + rr < m) art[(size_t)q * mld + k0 + rr] = acc;
                            }
                        }
                    } else {
                        for (int k = 0; k < m; ++k) {
                            double acc = jin[(size_t)k * nz + ucol];
                            if (realq)
                                for (int cb = 0; cb < nchunk; ++cb) {
                                    unsigned long long mk = fmask[k * nchunk + cb];
                                    while (mk) {
                                        const int row = cb * 64 + (int)__builtin_ctzll(mk);
                                        mk &= mk - 1;
                                        acc += jin[(size_t)k * nz + row] * phi[(size_t)row * nzu + q];
                                    }
                                }
                            art[(size_t)q * mld + k] = acc;
                        }
                    }
                }
                // rows of the bounds lb <= z + d <= ub (NLOptimizer::setStateBounds / setInputBounds): a row of [Phi; I]
                input_bounds();
                dense_bounds();
                for (int k = lane; k < m; k += 64) {
                    double s = gin[k];
                    for (int cb = 0; cb < nchunk; ++cb) {
                        unsigned long long mk = fmask[k * nchunk + cb];
                        while (mk) {
                            const int row = cb * 64 + (int)__builtin_ctzll(mk);
                            mk &= mk - 1;
                            s += jin[(size_t)k * nz + row] * r[row];
                        }
                    }
                    br[k] = s;
                }
                scan_user_rows(fmask, nchunk);
            } else {
                // no row reads a state.  Reduced gradient gr = g_u + Ju' lam with Jx' lam = -g_x: one backward sweep.
                for (int row = lane; row < nxs; row += 64) lamw[row] = g[row];
                nl_wave_sync();
                lam_dyn = backsolve(true);
                for (int q = lane; q < nr; q += 64) {
                    if (q == nzu) { gr[q] = g[nz - 1]; continue; }
                    const int bq = q / NU, jq = q - bq * NU;
                    double s = g[nxs + q];
                    for (int i = bq; i < (bq == ch - 1 ? ph : bq + 1); ++i) {
                        const double *Jb = jeq + (size_t)i * NX * W + 2 * NX + jq;
                        double bv[NX], lv[NX];
#pragma unroll
                        for (int a = 0; a < NX; ++a) { bv[a] = Jb[a * W]; lv[a] = lamw[i * NX + a]; }
#pragma unroll
                        for (int a = 0; a < NX; ++a) s = fma(bv[a], lv[a], s);
                    }
                    gr[q] = s;
                }
                scan_user_rows(nullptr, 0);
                input_bounds();
                if (!sparse_rows) dense_bounds();
            }
            nl_wave_sync();
            lap(1);
            // ---- damped BFGS update of the inverse Hessian estimate (Powell), s = a p, y = change of the reduced Lagrangian gradient
            if (have_old) {
                double sBs = 0, sy = 0;
                for (int q = lane; q < nq; q += 64) {
                    double gl = gr[q];
                    for (int t = 0; t < nw_keep; ++t) {
                        const int k = (int)wq[t], cn = sp_count(k);
                        const double ml = sgq[t] * uq[t];
                        if (cn < 0) gl += art[(size_t)q * mld + k] * ml;
                        else for (int j = 0; j < cn; ++j) if (sp_index(k, j) == q) gl += sp_value(k, j) * ml;
                    }
                    const double y = gl - glold[q], Bs = -a_prev * glold[q];
                    v0[q] = y; v1[q] = Bs;
                    sBs += sv[q] * Bs; sy += sv[q] * y;
                }
                sBs = wave_sum(sBs); sy = wave_sum(sy);
                if (sy < 0.2 * sBs) {
                    const double th = 0.8 * sBs / (sBs - sy);
                    for (int q = lane; q < nq; q += 64) v0[q] = th * v0[q] + (1 - th) * v1[q];
                    sy = th * sy + (1 - th) * sBs;
                }
                nl_wave_sync();
                if (sy > 1e-300) {
                    const double rho = 1.0 / sy;
                    double yHy = 0;
                    for (int q = lane; q < nq; q += 64) {
                        const double s = gdot<GU>(hinv + q, nr, v0, nq);
                        v2[q] = s; yHy += s * v0[q];
                    }
                    yHy = wave_sum(yHy);
                    nl_wave_sync();
                    const double cc = rho * rho * yHy + rho;
                    for (int q = lane; q < nq; q += 64) v3[q] = sv[q];          // s next to Hy in LDS
                    nl_wave_sync();
                    for (int q = lane; q < nq; q += 64) {
                        const double hyq = v2[q], sq = v3[q];
                        int i = 0;
                        for (; i + 8 <= nq; i += 8) {                           // eight loads, eight updates, eight stores
                            double hv[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) hv[u] = hinv[(size_t)(i + u) * nr + q];
#pragma unroll
                            for (int u = 0; u < 8; ++u) hv[u] += -rho * (v3[i + u] * hyq + v2[i + u] * sq) + cc * v3[i + u] * sq;
#pragma unroll
                            for (int u = 0; u < 8; ++u) hinv[(size_t)(i + u) * nr + q] = hv[u];
                        }
                        for (; i < nq; ++i) hinv[(size_t)i * nr + q] += -rho * (v3[i] * hyq + v2[i] * sq) + cc * v3[i] * sq;
                    }
                }
                nl_wave_sync();
            }

            lap(2);
            // ---- sub-problem: min 1/2 p'Bp + gr'p  s.t.  art' p + br <= 0   (Goldfarb-Idnani, range-space form on B^-1)
            double *xq = v0, *np_ = v1, *vv = v2, *zd = v3;
            for (int k = lane; k < mt; k += 64) mu[k] = 0.0;
            nl_wave_sync();
            int nw = 0, qp_fail = 0; bool qp_ok = true, qp_done = false;
            int ndense_w = 0;                                           // working rows that are not in the sparse form
            auto sp_dot = [&](int k, const double *x) {                 // (column k of art)' x for a row in the sparse form
                const int mw = s1m[k];
                double acc = s1v[k] * x[mw & 0xffff];
                if ((mw >> 16) > 1) {
                    const double *v = spv + (size_t)k * kNlSparse; const int *ix = spi + (size_t)k * kNlSparse;
                    acc = fma(v[3], x[ix[3]], fma(v[2], x[ix[2]], fma(v[1], x[ix[1]], acc)));
                }
                return acc;
            };
            int fac_n = 0;                                              // rows the factor in LDS stands for (-1: stale)
            double y0 = 0, y1 = 0;                                      // L^-1 t of the last solve: the factor's next row if the entering row joins
            // tq <- S^-1 tq over the working set, with its factor (brought up to date first if rows left)
            const Factor Fac{Lp, invd, ybuf, Sbig, NL, SLD};
            auto solve_ws = [&]() -> bool {
                bool ok = true;
                if (fac_n != nw) {
                    ok = chol_factor<TWO>((int)(Lp - smem), (int)(invd - smem), (int)(ybuf - smem), Sbig, NL, SLD, Ssm, SLD, nw, lane);
                    fac_n = ok ? nw : -1;
                }
                double t0 = lane < nw ? tq[lane] : 0.0, t1 = lane + 64 < nw ? tq[lane + 64] : 0.0;
                chol_forward<TWO>(Fac, nw, t0, t1, lane);
                y0 = t0; y1 = t1;
                chol_backward<TWO>(Fac, nw, t0, t1, lane);
                if (lane < nw) tq[lane] = t0;
                if (lane + 64 < nw) tq[lane + 64] = t1;
                nl_wave_sync();
                return ok;
            };
            auto drop_row = [&](int kdrop) {                            // working-set slot kdrop <- the last slot
                const int last = nw - 1;
                if (sp_count((int)wq[kdrop]) < 0) --ndense_w;
                fac_n = -1;
                if (kdrop != last) {
                    for (int q = lane; q < nq; q += 64) { qn[(size_t)kdrop * nr + q] = qn[(size_t)last * nr + q]; qv[(size_t)kdrop * nr + q] = qv[(size_t)last * nr + q]; }
                    nl_wave_sync();
                    for (int r = lane; r < nw; r += 64) Ssm[r * SLD + kdrop] = Ssm[r * SLD + last];
                    nl_wave_sync();
                    for (int r = lane; r < nw; r += 64) Ssm[kdrop * SLD + r] = Ssm[last * SLD + r];
                    nl_wave_sync();
                    if (lane == 0) { uq[kdrop] = uq[last]; wq[kdrop] = wq[last]; sgq[kdrop] = sgq[last]; }
                }
                --nw;
                nl_wave_sync();
            };
            // One sweep over B^-1 serves the unconstrained minimiser x = -B^-1 gr and B^-1 n for three rows of the previous
            // working set at a time (their normals parked in LDS): every product with B^-1 costs a full pass of loads, and
            // the warm start needs one per row.
            // A row in the sparse form needs no sweep: B^-1 n is a combination of a few columns of B^-1.
            // (WB rows at a time: their columns of B^-1 are requested together)
            for (int t0 = 0; t0 < nw_keep; t0 += WB) {
                int kk[WB], cc[WB], i0[WB];
                double v0s[WB];
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int tt = min(t0 + u, nw_keep - 1);
                    kk[u] = (int)wq[tt];
                    cc[u] = t0 + u < nw_keep ? sp_count(kk[u]) : -1;
                    i0[u] = s1m[kk[u]] & 0xffff;
                    v0s[u] = sgq[tt] * s1v[kk[u]];
                }
                for (int q = lane; q < nq; q += 64) {
                    double hv[WB], nvl[WB];
#pragma unroll
                    for (int u = 0; u < WB; ++u) hv[u] = cc[u] >= 1 ? hinv[(size_t)i0[u] * nr + q] : 0.0;
#pragma unroll
                    for (int u = 0; u < WB; ++u) { hv[u] *= v0s[u]; nvl[u] = (cc[u] >= 1 && i0[u] == q) ? v0s[u] : 0.0; }
#pragma unroll
                    for (int u = 0; u < WB; ++u) {
                        if (cc[u] > 1) {
                            const double sg = sgq[t0 + u];
                            for (int j = 1; j < cc[u]; ++j) {
                                const int ix = sp_index(kk[u], j);
                                const double v = sg * sp_value(kk[u], j);
                                if (ix == q) nvl[u] += v;
                                hv[u] = fma(hinv[(size_t)ix * nr + q], v, hv[u]);
                            }
                        }
                        if (cc[u] >= 0) { qn[(size_t)(t0 + u) * nr + q] = nvl[u]; qv[(size_t)(t0 + u) * nr + q] = hv[u]; }
                    }
                }
            }
            int cur = 0;
            for (bool first = true; first || cur < nw_keep; first = false) {
                int slot[3] = {0, 0, 0}, nv = 0;
                while (cur < nw_keep && nv < 3) {
                    if (sp_count((int)wq[cur]) < 0) { if (nv == 0) slot[0] = cur; else if (nv == 1) slot[1] = cur; else slot[2] = cur; ++nv; }
                    ++cur;
                }
                if (!first && nv == 0) break;
                ndense_w += nv;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (u >= nv) continue;
                    const int k = (int)wq[slot[u]];
                    for (int q = lane; q < nq; q += 64) v1[u * nr + q] = sgq[slot[u]] * art[(size_t)q * mld + k];
                }
                nl_wave_sync();
                if (nv == 0) {                                         // x = -B^-1 gr alone
                    for (int q = lane; q < nq; q += 64) {
                        const double s = gdot<GU>(hinv + q, nr, gr, nq);
                        xq[q] = -s;
                    }
                    nl_wave_sync();
                    continue;
                }
                for (int q = lane; q < nq; q += 64) {
                    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    const double *hq = hinv + q;
                    int j = 0;
                    for (; j + 8 <= nq; j += 8) {
                        double h[8], gv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { h[u] = hq[(size_t)(j + u) * nr]; gv[u] = first ? gr[j + u] : 0.0; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            a0 = fma(h[u], gv[u], a0);
                            a1 = fma(h[u], v1[j + u], a1); a2 = fma(h[u], v1[nr + j + u], a2); a3 = fma(h[u], v1[2 * nr + j + u], a3);
                        }
                    }
                    for (; j < nq; ++j) {
                        const double hv = hq[(size_t)j * nr];
                        a0 = fma(hv, first ? gr[j] : 0.0, a0);
                        a1 = fma(hv, v1[j], a1); a2 = fma(hv, v1[nr + j], a2); a3 = fma(hv, v1[2 * nr + j], a3);
                    }
                    if (first) xq[q] = -a0;
                    const double acc[3] = {a1, a2, a3};
#pragma unroll
                    for (int u = 0; u < 3; ++u)
                        if (u < nv) { qn[(size_t)slot[u] * nr + q] = v1[u * nr + q]; qv[(size_t)slot[u] * nr + q] = acc[u]; }
                }
                nl_wave_sync();
            }
            // warm start: the rows active in the previous sub-problem, as long as their multipliers stay non-negative --
            // the minimiser on that set with u >= 0 is a valid state of the dual method
            MPCX_STAT(const long long tw0 = __builtin_readcyclecounter();)
            MPCX_STAT(qst[4] += nw_keep;)
            if (nw_keep > 0) {
                nw = nw_keep;
                for (int e0 = 0; e0 < nw * nw; e0 += 64 * SB) {         // SB entries per lane in flight
                    double s4[SB];
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int e2 = min(e0 + 64 * u + lane, nw * nw - 1);
                        const int a = e2 / nw, b2 = e2 - a * nw;
                        const int ka = (int)wq[a];
                        s4[u] = sp_count(ka) >= 0 ? sgq[a] * sp_dot(ka, qv + (size_t)b2 * nr) : gdot2(qn + (size_t)a * nr, 1, qv + (size_t)b2 * nr, 1, nq);
                    }
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int e2 = e0 + 64 * u + lane;
                        if (e2 < nw * nw) Ssm[(e2 / nw) * SLD + e2 % nw] = s4[u];
                    }
                }
                nl_wave_sync();
                while (nw > 0) {
                    if (ndense_w == 0) {
                        for (int t = lane; t < nw; t += 64) { const int k = (int)wq[t]; tq[t] = sgq[t] * br[k] + sgq[t] * sp_dot(k, xq); }
                    } else if (nw <= 16 && nq <= 64) {
                        for (int t0 = 0; t0 < nw; t0 += 8) {
                            double part[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) part[u] = (t0 + u < nw && lane < nq) ? qn[(size_t)(t0 + u) * nr + lane] : 0.0;
                            const double xl = lane < nq ? xq[lane] : 0.0;
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const double sred = wave_sum(part[u] * xl);
                                if (lane == 0 && t0 + u < nw) tq[t0 + u] = sgq[t0 + u] * br[(int)wq[t0 + u]] + sred;
                            }
                        }
                    } else {
                        for (int t = lane; t < nw; t += 64) {
                            const double s2 = sgq[t] * br[(int)wq[t]] + gdot(qn + (size_t)t * nr, 1, xq, nq);
                            tq[t] = s2;
                        }
                    }
                    nl_wave_sync();
                    if (!solve_ws()) { nw = 0; break; }                              // dependent rows: start cold
                    // Every row with a negative multiplier leaves at once.  (Measured against shedding only the most negative ones
                    // per round, config 5: fewer steps of the dual method afterwards, 11 instead of 14, but more rounds here and
                    // rows leaving inside the dual method, each a re-factorisation: 12 % slower overall.)  Equalities stay.
                    auto sheds = [&](int t) { const int k = (int)wq[t]; return tq[t] < 0.0 && !(k >= mi && k < m); };
                    int neg = -1;
                    for (int t = 0; t < nw; ++t) if (sheds(t)) neg = t;
                    if (neg < 0) break;
                    for (int t = nw - 1; t >= 0; --t) {
                        if (sheds(t)) { const double tl = tq[nw - 1]; MPCX_STAT(++qst[5];) drop_row(t); if (lane == 0) tq[t] = tl; nl_wave_sync(); }
                    }
                }
                if (nw > 0) {
                    for (int q = lane; q < nq; q += 64) {
                        const double s2 = xq[q] - gdot<GU>(qv + q, nr, tq, nw);
                        xq[q] = s2;
                    }
                    for (int r = lane; r < nw; r += 64) uq[r] = tq[r];
                    nl_wave_sync();
                }
            }
            if (nw == 0) { fac_n = 0; ndense_w = 0; }
            MPCX_STAT(qst[6] += __builtin_readcyclecounter() - tw0;)
            for (int qit = 0; qit < 8 * (mt + nq) + 16; ++qit) {
                MPCX_STAT(++qst[0];)
                double vmax = -1e300; int pidx = 0x7fffffff;
                for (int k = lane; k < mt; k += 64) {
                    double s = br[k] + (sp_count(k) >= 0 ? sp_dot(k, xq) : gdot(art + k, mld, xq, nq));
                    if (k >= mi && k < m) s = fabs(s);                   // an equality is violated on either side
                    bool inw = mu[k] == -1.0 && !(k >= mi && k < m);     // set aside (see below)
                    for (int t = 0; t < nw; ++t) inw |= ((int)wq[t] == k);
                    if (!inw && s > vmax) { vmax = s; pidx = k; }
                }
                wave_argmax(vmax, pidx);
                if (mt == 0 || vmax <= 1e-12) { qp_done = true; break; }   // primal feasible (well inside the reported 1e-10): optimal
                if (nw >= KW) { qp_ok = false; qp_fail = -3; break; }           // working set full
                // an equality enters oriented so that it reads "n'p + b <= 0, violated"; it is never shed afterwards
                const bool p_is_eq = pidx >= mi && pidx < m;
                const int pcn = sp_count(pidx);
                double sgn = 1.0;
                if (p_is_eq) {
                    double part = 0;
                    if (pcn >= 0) part = lane == 0 ? sp_dot(pidx, xq) : 0.0;
                    else for (int q = lane; q < nq; q += 64) part += art[(size_t)q * mld + pidx] * xq[q];
                    sgn = br[pidx] + wave_sum(part) < 0.0 ? -1.0 : 1.0;
                }
                // the normal n of the entering row and v = B^-1 n (they stay as they are while rows leave to make room)
                if (pcn >= 0) {
                    for (int q = lane; q < nq; q += 64) {
                        double nvl = 0, hv = 0;
                        for (int j = 0; j < pcn; ++j) {
                            const int ix = sp_index(pidx, j);
                            const double v = sgn * sp_value(pidx, j);
                            if (ix == q) nvl += v;
                            hv = fma(hinv[(size_t)ix * nr + q], v, hv);
                        }
                        np_[q] = nvl; vv[q] = hv;
                    }
                } else {
                    for (int q = lane; q < nq; q += 64) np_[q] = sgn * art[(size_t)q * mld + pidx];
                    nl_wave_sync();
                    for (int q = lane; q < nq; q += 64) {
                        const double s = gdot<GU>(hinv + q, nr, np_, nq);
                        vv[q] = s;
                    }
                }
                nl_wave_sync();
                double up = 0.0, sp = vmax;
                bool added = false;
                for (int inner = 0; inner <= KW + 1 && !added; ++inner) {
                    // t = N_W v (also the new column of S), rr = S^-1 t
                    if (ndense_w == 0) {
                        for (int t = lane; t < nw; t += 64) tq[t] = sgq[t] * sp_dot((int)wq[t], vv);
                    } else if (nw <= 16 && nq <= 64) {
                        // few rows: the lanes split each dot product (coalesced loads, all rows in flight, a butterfly per row)
                        // instead of each walking one row on its own
                        for (int t0 = 0; t0 < nw; t0 += 8) {
                            double part[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) part[u] = (t0 + u < nw && lane < nq) ? qn[(size_t)(t0 + u) * nr + lane] : 0.0;
                            const double vl = lane < nq ? vv[lane] : 0.0;
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const double sred = wave_sum(part[u] * vl);
                                if (lane == 0 && t0 + u < nw) tq[t0 + u] = sred;
                            }
                        }
                    } else {
                        for (int t = lane; t < nw; t += 64) {
                            const double s = gdot(qn + (size_t)t * nr, 1, vv, nq);
                            tq[t] = s;
                        }
                    }
                    nl_wave_sync();
                    const double tcol = lane < nw ? tq[lane] : 0.0, tcol2 = lane + 64 < nw ? tq[lane + 64] : 0.0;   // keep N_W v: it becomes S[:, new]
                    MPCX_STAT(++qst[1]; const long long tf0 = __builtin_readcyclecounter();)
                    if (nw) solve_ws();
                    MPCX_STAT(qst[7] += __builtin_readcyclecounter() - tf0;)
                    double zn = 0;
                    for (int q = lane; q < nq; q += 64) {
                        const double s = vv[q] - gdot<GU>(qv + q, nr, tq, nw);
                        zd[q] = s; zn += s * np_[q];
                    }
                    zn = wave_sum(zn);
                    double npn = 0;
                    for (int q = lane; q < nq; q += 64) npn += np_[q] * np_[q];
                    npn = wave_sum(npn);
                    // dual ratio test
                    double t1 = 1e300; int kdrop = -1;
                    {
                        double tneg = -1e300; int tidx = 0x7fffffff;      // the smallest ratio (lowest slot on ties), lanes = slots
                        for (int t = lane; t < nw; t += 64) {
                            const double rr = tq[t];
                            const int kt = (int)wq[t];
                            if (rr > 1e-14 && !(kt >= mi && kt < m)) { const double tj = uq[t] / rr; if (-tj > tneg) { tneg = -tj; tidx = t; } }
                        }
                        wave_argmax(tneg, tidx);
                        if (tneg > -1e300) { t1 = -tneg; kdrop = tidx; }
                    }
                    const bool can_move = zn > 1e-13 * fmax(1.0, npn);
                    const double t2 = can_move ? sp / zn : 1e300;
                    const double tt = fmin(t1, t2);
                    if (tt >= 1e300) {
                        // no step: the row is a combination of working rows.  Violated by round-off only (a copy of an
                        // active row): set it aside; violated for real: the linearised constraints are inconsistent.
                        if (sp <= 1e-7 && !p_is_eq) { if (lane == 0) mu[pidx] = -1.0; nl_wave_sync(); added = true; break; }
                        qp_ok = false; break;
                    }
                    nl_wave_sync();
                    if (can_move) {
                        for (int q = lane; q < nq; q += 64) xq[q] -= tt * zd[q];
                        sp -= tt * zn;
                    }
                    for (int r = lane; r < nw; r += 64) uq[r] -= tt * tq[r];
                    up += tt;
                    nl_wave_sync();
                    if (t2 <= t1) {                                     // full step: the row joins the working set
                        for (int q = lane; q < nq; q += 64) { qn[(size_t)nw * nr + q] = np_[q]; qv[(size_t)nw * nr + q] = vv[q]; }
                        if (lane < nw) { Ssm[lane * SLD + nw] = tcol; Ssm[nw * SLD + lane] = tcol; }
                        if (lane + 64 < nw) { Ssm[(lane + 64) * SLD + nw] = tcol2; Ssm[nw * SLD + lane + 64] = tcol2; }
                        double snn = 0;
                        for (int q = lane; q < nq; q += 64) snn += np_[q] * vv[q];
                        snn = wave_sum(snn);
                        if (lane == 0) { Ssm[nw * SLD + nw] = snn; uq[nw] = up; wq[nw] = (double)pidx; sgq[nw] = sgn; }
                        if (fac_n == nw) { chol_append<TWO>(Fac, nw, y0, y1, snn, snn, lane); fac_n = nw + 1; } else fac_n = -1;
                        ++nw; added = true;
                        if (pcn < 0) ++ndense_w;
                    } else {                                            // a multiplier hit zero: that row leaves, try again
                        drop_row(kdrop);
                    }
                    nl_wave_sync();
                }
                if (!qp_ok) break;
                if (!added) { qp_ok = false; break; }
            }
            if (!qp_ok || !qp_done) { code = qp_fail ? qp_fail : -1; break; }
            for (int k = lane; k < mt; k += 64) mu[k] = 0.0;
            nl_wave_sync();
            for (int t = lane; t < nw; t += 64) mu[(int)wq[t]] = sgq[t] * uq[t];
            nw_keep = nw;
            MPCX_STAT(qst[2] += nw; qst[3] = nw > qst[3] ? nw : qst[3];)
            for (int q = lane; q < nr; q += 64) p[q] = q < nq ? xq[q] : 0.0;
            nl_wave_sync();

            lap(3);
            // ---- full-space step d = [r + Phi p_u ; p]
            double dmax = 0, cmax = 0, gd = 0;
            if (needs_phi) {
                for (int row = lane; row < nxs; row += 64) {
                    const double s = r[row] + gdot2(phi + (size_t)row * nzu, 1, p, 1, nzu);
                    d[row] = s;
                }
            } else {
                sweep(2);
            }
            for (int q = lane; q < nr; q += 64) d[nxs + q] = p[q];
            nl_wave_sync();
            for (int k = lane; k < nz; k += 64) { dmax = fmax(dmax, fabs(d[k])); gd += g[k] * d[k]; }
            for (int k = lane; k < nxs; k += 64) cmax = fmax(cmax, fabs(c[k]));
            dmax = wave_max(dmax); cmax = wave_max(cmax); gd = wave_sum(gd);
            double zmax = 0;
            for (int k = lane; k < nz; k += 64) zmax = fmax(zmax, fabs(z[k]));
            zmax = wave_max(zmax);
            for (int k = mi + lane; k < m; k += 64) cmax = fmax(cmax, fabs(gin[k]));     // user equalities count as defects
            cmax = wave_max(cmax);
            if (dmax <= S.tol_step * fmax(1.0, zmax) && cmax <= S.tol_con) {
                // converged: take this last (tiny) step too -- it carries the final correction of the active constraints
                // (bounds on the decision vector are kept exactly, as nlopt's SLSQP keeps them: its iterates are clamped to [lb, ub])
                for (int k = lane; k < nz; k += 64) { const double zn = z[k] + d[k]; z[k] = M.nbnd > 0 ? fmin(fmax(zn, M.zlb[k]), M.zub[k]) : zn; }
                nl_wave_sync();
                code = 4; ++it;
                final_eval = true;
                continue;
            }

            // reduced Lagrangian gradient at this point with the new multipliers: the BFGS memory
            for (int q = lane; q < nq; q += 64) {
                // (the working set's normals, signs included, are still in qn: independent loads, coalesced over q)
                const double gl = gr[q] + gdot<GU>(qn + q, nr, uq, nw_keep);
                glold[q] = gl;
            }
            // multipliers of the dynamics equalities: Jx' lam = -(g_x + Jin_x' mu), a backward sweep over the blocks;
            // the weight of the l1 merit function has to dominate them and mu
            for (int row = lane; needs_phi && row < nxs; row += 64) {        // (no row reads a state otherwise: the sweep of the reduction stands)
                double s2 = g[row];
                for (int t = 0; t < nw_keep; ++t) {                     // mu lives on the working set
                    const int k = (int)wq[t];
                    if (k < m) s2 += jin[(size_t)k * nz + row] * (sgq[t] * uq[t]);
                    else if (M.bnd_idx[k - m] == row) s2 += M.bnd_sign[k - m] * uq[t];
                }
                lamw[row] = s2;
            }
            nl_wave_sync();
            double lam_max = needs_phi ? backsolve(false) : lam_dyn;
            for (int r = lane; r < nw_keep; r += 64) lam_max = fmax(lam_max, fabs(uq[r]));
            lam_max = wave_max(lam_max);
            if (1.1 * lam_max > nu_pen) nu_pen = 1.5 * lam_max;
            double viol = 0;
            for (int k = lane; k < nxs; k += 64) viol += fabs(c[k]);
            for (int k = lane; k < m; k += 64) viol += k < mi ? fmax(gin[k], 0.0) : fabs(gin[k]);
            viol = wave_sum(viol);
            const double phi0 = scal[0] + nu_pen * viol;
            const double dphi = fmin(gd - nu_pen * viol, 0.0);

            // ---- line search: lane l tries a = 2^-l on the l1 merit function
            unwrap<Mdl>(M, z, x0, Xs, Us, lane);
            for (int k = lane; k < (ph + 1) * NX; k += 64) { const int i = k / NX; dXs[k] = i == 0 ? 0.0 : sc.over_ss(d[k - NX], k - i * NX); }
            for (int k = lane; k < (ph + 1) * NU; k += 64) {
                const int i = k / NU, j = k - i * NU;
                dUs[k] = sc.by_su(d[nxs + min(min(i, ph - 1), ch - 1) * NU + j], j);
            }
            nl_wave_sync();
            double a_step;
            {
                // eight step lengths at a time, eight lanes each: lane (g, part) evaluates every eighth defect and constraint of
                // trial point a = 2^-(g + 8 round), the cost is one lane's (it is a black box over the whole horizon), a
                // three-step butterfly adds the parts up.  Almost always the first round holds an acceptable length.
                const int grp = lane >> 3, part = lane & 7;
                unsigned long long bal = 0;
                int round = 0;
                for (; round < 5 && !bal; ++round) {
                    const double al = ldexp(1.0, -(grp + 8 * round));
                    const Lin XL{Xs, dXs, NX, al}, UL{Us, dUs, NU, al};
                    const double et = z[nz - 1] + al * d[nz - 1];
                    double mer = 0.0, vio = 0;
                    if constexpr (Mdl::VECTOR_HOOKS) {
                        // whole-vector hooks: one lane of the group evaluates the cost, one the inequality vector, one the
                        // equality vector (into the trial point's row of the scratch), all of them share the defects below
                        using MX = typename Mdl::MatX; using MU = typename Mdl::MatU; using MY = typename Mdl::MatY;
                        using VI = typename Mdl::VecI; using VE = typename Mdl::VecE;
                        constexpr int NI = Mdl::NI, NE = Mdl::NE, NY = Mdl::NY;
                        double *lsb = hk + (size_t)128 * m, *ytr = lsb + (size_t)kNlTrials * m;
                        const MX XV = MX::trajectory(Xs, dXs, al);
                        const MU UV = MU::trajectory(Us, dUs, al);
                        MY YV = MY::zeros_view();
                        if (M.has_output) {
                            double *yt = ytr + (size_t)grp * (ph + 1) * NY;
                            // every lane makes the same number of calls (a hook may sit behind a function pointer: keep the
                            // call sites out of loops whose trip count differs between lanes); surplus lanes redo row ph
                            for (int i0 = 0; i0 <= ph; i0 += 8) {
                                const int i = min(i0 + part, ph);
                                double xr[NX], ur[NU], yr[NY > 0 ? NY : 1];
                                for (int a = 0; a < NX; ++a) xr[a] = XL(i, a);
                                for (int a = 0; a < NU; ++a) ur[a] = UL(i, a);
                                Mdl::out(yr, xr, ur, prm, (unsigned)i);
                                if (i0 + part <= ph) for (int a = 0; a < NY; ++a) yt[i * NY + a] = yr[a];
                            }
                            nl_wave_sync();
                            YV = MY::trajectory(yt);
                        }
                        double *gl = lsb + (size_t)grp * m;
                        // the eight lanes of a trial point make the same calls (they store the same values to the same place):
                        // no hook call sits in control flow that differs between lanes
                        const double ct = Mdl::cost(XV, YV, UV, et, prm);
                        if (part == 0) mer = ct;
                        if constexpr (NI > 0) { VI o = VI::output(gl, 1); Mdl::ineq_all(o, XV, YV, UV, et, prm); }
                        if constexpr (NE > 0) { VE o = VE::output(gl + NI, 1); Mdl::eq_all(o, XV, UV, prm); }
                        nl_wave_sync();
                        for (int k = part; k < m; k += 8) vio += k < mi ? fmax(gl[k], 0.0) : fabs(gl[k]);
                    } else {
                        if (part == 0) mer = Mdl::cost(XL, UL, et, ph, prm);
                        for (int k = part; k < mi; k += 8) vio += fmax(Mdl::ineq(k, XL, UL, et, ph, prm), 0.0);
                        for (int k = part; k < m - mi; k += 8) vio += fabs(Mdl::eq(k, XL, UL, ph, prm));
                    }
                    const double h = 0.5 * M.Ts;
                    for (int i0 = 0; i0 < ph; i0 += 8) {              // same trip count in every lane; a surplus lane redoes step ph-1
                        const bool live = i0 + part < ph;
                        const int i = live ? i0 + part : ph - 1;
                        double xk[NX], xk1[NX], uk[NU], fa[NX], fb[NX], v = 0;
                        for (int a = 0; a < NX; ++a) { xk[a] = XL(i, a); xk1[a] = XL(i + 1, a); }
                        for (int a = 0; a < NU; ++a) uk[a] = UL(i, a);
                        call_f<Mdl>(fa, xk, uk, prm, i);
                        if (CT) {
                            call_f<Mdl>(fb, xk1, uk, prm, i);
                            for (int a = 0; a < NX; ++a) v += fabs(sc.over_ss(xk[a] + (h * (fa[a] + fb[a])) - xk1[a], a));
                        } else {
                            for (int a = 0; a < NX; ++a) v += fabs(sc.over_ss(xk1[a] - fa[a], a));
                        }
                        if (live) vio += v;
                    }
                    mer += nu_pen * vio;
                    mer += __shfl_xor(mer, 1); mer += __shfl_xor(mer, 2); mer += __shfl_xor(mer, 4);
                    const bool ok = part == 0 && mer <= phi0 + 1e-4 * al * dphi;
                    bal = __ballot(ok);
                }
                if (!bal) {                                         // no decrease left within 2^-40: the iteration has stalled
                    if (cmax <= fmax(S.tol_con, 1e-8) && dmax <= 1e-3 * fmax(1.0, zmax)) { code = 4; break; }
                    if (resets >= 5) { code = -4; break; }
                    // far from a solution: the curvature estimate has gone bad -- forget it and try a steepest-descent-like step
                    ++resets;
                    for (int k = lane; k < nr * nr; k += 64) hinv[k] = (k / nr == k % nr) ? 1.0 : 0.0;
                    have_old = false;
                    nl_wave_sync();
                    ++it;
                    continue;
                }
                a_step = ldexp(1.0, -((int)__builtin_ctzll(bal) / 8 + 8 * (round - 1)));
            }
            for (int q = lane; q < nr; q += 64) sv[q] = a_step * p[q];
            f_prev = scal[0];
            step_l1 = 0; z_l1 = 0; step_max = 0;
            for (int k = lane; k < nz; k += 64) {
                const double dk = a_step * d[k];
                const double zn = z[k] + dk;
                z[k] = M.nbnd > 0 ? fmin(fmax(zn, M.zlb[k]), M.zub[k]) : zn;
                step_l1 += fabs(dk); z_l1 += fabs(z[k]); step_max = fmax(step_max, fabs(dk));
            }
            step_l1 = wave_sum(step_l1); z_l1 = wave_sum(z_l1); step_max = wave_max(step_max);
            a_prev = a_step; have_old = true; stepped = true;
            nl_wave_sync();
            ++it;
        }
        MPCX_STAT(if (lane == 0) { for (int k = 0; k < 6; ++k) scal[2 + k] = (double)cyc[k]; for (int k = 0; k < 8; ++k) scal[8 + k] = (double)qst[k]; })

        // ---- results (NLOptimizer.hpp:536-624): cmd = U.row(0), cost, status map, feasibility of the user inequalities
        double gmax = -1e300;
        double hmax = 0.0;
        for (int k = lane; k < m; k += 64) { if (k < mi) gmax = fmax(gmax, gin[k]); else hmax = fmax(hmax, fabs(gin[k])); }
        gmax = wave_max(gmax); hmax = wave_max(hmax);
        unwrap<Mdl>(M, z, x0, Xs, Us, lane);
        if (!exec_full) code = -5;
        const bool failed = code < 0;
        if (S.cmd) for (int j = lane; j < NU; j += 64) S.cmd[(size_t)b * NU + j] = failed ? u0[j] : Us[j];
        if (S.z_out) for (int k = lane; k < nz; k += 64) S.z_out[(size_t)b * nz + k] = z[k];
        if (S.mu_out) for (int k = lane; k < mt; k += 64) S.mu_out[(size_t)b * mt + k] = mu[k] == -1.0 ? 0.0 : mu[k];
        if (S.seq_state) for (int k = lane; k < (ph + 1) * NX; k += 64) S.seq_state[(size_t)b * (ph + 1) * NX + k] = failed ? 0.0 : Xs[k];
        if (S.seq_input) for (int k = lane; k < (ph + 1) * NU; k += 64) S.seq_input[(size_t)b * (ph + 1) * NU + k] = failed ? 0.0 : Us[k];
        if (S.seq_output)                                   // Model::getOutput (Model.hpp:72-96): row i = out(x_i, u_i), zeros without one
            for (int i = lane; i <= ph; i += 64) {
                double y[Mdl::NY > 0 ? Mdl::NY : 1];
                for (int a = 0; a < Mdl::NY; ++a) y[a] = 0.0;
                bool has_out = Mdl::HAS_OUTPUT;
                if constexpr (Mdl::VECTOR_HOOKS) has_out = M.has_output != 0;
                if (has_out && !failed) call_out<Mdl>(y, Xs + i * NX, Us + i * NU, prm, i);
                for (int a = 0; a < Mdl::NY; ++a) S.seq_output[((size_t)b * (ph + 1) + i) * Mdl::NY + a] = y[a];
            }
        if (lane == 0) {
            if (S.cost) S.cost[b] = failed ? __builtin_huge_val() : scal[0];
            if (S.solver_status) S.solver_status[b] = code;
            if (S.status) S.status[b] = (code == 3 || code == 4) ? 0 : (code == 5 ? 1 : 3);     // SUCCESS / MAX_ITERATION / ERROR (NLOptimizer.hpp:729-750)
            if (S.is_feasible) S.is_feasible[b] = ((mi == 0 || gmax <= S.ieq_tol) && hmax <= S.eq_tol) ? 1 : 0;    // Constraints.hpp:157-202
            if (S.iterations) S.iterations[b] = it;
        }
        nl_wave_sync();
    }
}

// Register budget: the per-lane state arrays grow with the state dimension (a dozen vectors of NX doubles are live in the
// sweeps and the finite differences), while the LDS slice of such a system already limits a CU to 3-6 wavefronts.  Large
// systems therefore get the whole register file of a SIMD (512 VGPRs, one wavefront per SIMD) instead of spilling.
template <class Mdl> constexpr int kSqpWavesPerSimd = Mdl::NX >= 12 ? 1 : 2;
template <class Mdl> constexpr bool kSqpLdsBlocks = Mdl::NX < 8 && !Mdl::VECTOR_HOOKS;     // instantiated with LDS-resident dynamics blocks as well
template <class Mdl, bool TWO, bool BLK = false>
__global__ __launch_bounds__(256, kSqpWavesPerSimd<Mdl>) void nlmpc_sqp(const NlmpcDev M, const NlmpcSolveDev S) { sqp_body<Mdl, TWO, BLK>(M, S); }

// ---- host side: workspace plan and launchers -----------------------------------------------------------------------
#if !defined(__HIPCC_RTC__)
// fills nzu, nr, nz, neq, kw, lds_per_wave and the workspace layout from the dimensions (and nbnd, vector_hooks, has_output)
inline void nlmpc_plan(NlmpcDev &m)
{
    const int nx = m.nx, nu = m.nu, ph = m.ph;
    auto imin = [](int a, int b) { return a < b ? a : b; };
    auto imax = [](int a, int b) { return a > b ? a : b; };
    m.nzu = m.ch * nu; m.nr = m.nzu + 1;
    m.nz = ph * nx + m.nzu + 1; m.neq = ph * nx;
    // a working set holds linearly independent rows: never more than there are rows or sub-problem variables
    m.kw = imin(kNlMaxWorking, imax(kNlLdsWorking, imin(m.nineq + m.nue + m.nbnd, m.nr)));
    const int KW = m.kw, mtot_ = m.nineq + m.nue + m.nbnd;
    const int ylds = (m.vector_hooks && m.has_output) ? (ph + 1) * m.ny : 0;
    // LDS slice of a wavefront: trajectories, working-set vectors and four vectors of the sub-problem stay; the tail holds
    // the step, the transcription's and the condensing's scratch, and during the sub-problem the packed factor of the
    // working set's Schur complement -- as many rows (nl) as fit while the CU keeps its wavefronts: 39 KB where a SIMD runs
    // one (nx >= 12; four blocks of one share 160 KB), 16 KB where it runs two (two blocks of four)
    const int fixed = (ph + 1) * (nx + nu) + ylds + 6 * KW + 4 * m.nr + mtot_ + (mtot_ + 1) / 2;
    const int tail_min = imax(imax((ph + 1) * (nx + nu) + ph * nu + 2 * nx * nx, kNlLdsWorking * (kNlLdsWorking + 1) / 2),
                              (m.nineq + m.nue) * ((ph * nx + 63) / 64));           // (the structure words of the reduction live there too)
    int cap = nx >= 12 ? 4992 : 2048;
    // Small built-in systems whose whole slice still fits 2048 doubles (workgroups of four wavefronts, eight per CU, as before): the
    // dynamics blocks and the sweeps' right-hand sides live in LDS for the whole solve -- the sweeps are chains of ph dependent steps run
    // by a few lanes, and a dependent load from L2 costs ~2.7k cycles on the loaded chip.  (A larger budget costs occupancy: config 3 at
    // six wavefronts per CU instead of eight lost more than it gained)
    int jl = (nx < 8 && !m.vector_hooks) ? ((ph * nx * (2 * nx + nu) + ph * nx * nx + 2 * ph * nx + m.nr + 1) & ~1) : 0;
    // (testing aids MPCX_DEBUG_LDS_BLOCKS = 0 | 2 and MPCX_DEBUG_LDS_CAP: read HERE, when a controller is created or its bounds change the plan -- never on a solve path)
    const int dbg_blocks = [] { const char *e = getenv("MPCX_DEBUG_LDS_BLOCKS"); return e ? atoi(e) : -1; }();
    const int dbg_cap = [] { const char *e = getenv("MPCX_DEBUG_LDS_CAP"); return e ? atoi(e) : -1; }();
    if (dbg_blocks == 0) jl = 0;                                        // A/B against the workspace form
    // (MPCX_DEBUG_LDS_BLOCKS=2, testing aid: the blocks in LDS next to whatever the factor gets -- with a two-level factor the
    // combination of DESIGN.md section 9-2 that the plan itself never selects)
    const bool force_blk = jl > 0 && dbg_blocks == 2;
    if (force_blk) cap = imax(fixed + tail_min, 2048 - jl);
    else if (jl > 0 && fixed + imax(tail_min, KW * (KW + 1) / 2) + jl <= 2048) cap = 2048 - jl; else jl = 0;
    if (dbg_cap >= 0) cap = dbg_cap;
    const int tail = imax(tail_min, imin(KW * (KW + 1) / 2, cap - fixed));
    m.nl = 0;
    while (m.nl < KW && (m.nl + 1) * (m.nl + 2) / 2 <= tail) ++m.nl;
    m.lds_per_wave = (fixed + tail + 1) & ~1;
    m.lds_blocks = -1;
    if (jl > 0 && (m.nl == KW || force_blk)) { m.lds_blocks = m.lds_per_wave; m.lds_per_wave += jl; }    // (a debug cap may have cut the factor: then not)
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 1) & ~1; return at; };
    NlmpcWsLayout &w = m.ws;
    const int mu_ = m.nineq + m.nue, mtot = mu_ + m.nbnd;
    const int mld = (mtot + 1) & ~1;
    w.z = take(m.nz); w.d = take(m.nz); w.g = take(m.nz); w.c = take(m.neq); w.jeq = take(ph * nx * (2 * nx + nu));
    w.gin = take(mu_); w.jin = take(mu_ * m.nz);
    w.r = take(m.neq); w.phi = take(m.neq * m.nzu); w.einv = take(ph * nx * nx);
    w.gr = take(m.nr); w.art = take(m.nr * mld); w.br = take(mtot);
    w.hinv = take(m.nr * m.nr); w.mu = take(mtot); w.glold = take(m.nr); w.s = take(m.nr); w.p = take(m.nr);
    w.qn = take(KW * m.nr); w.qv = take(KW * m.nr); w.qs = take(KW * (KW + 1)); w.qs2 = take(KW * (KW + 1)); w.scal = take(16);
    w.lamw = take(m.neq);
    w.hook = take(m.vector_hooks ? nlmpc_hook_scratch(m) : 0);
    w.sp = take(mtot * kNlSparse + (mtot * kNlSparse + mtot + 1) / 2);
    w.total = o;
}

template <class Mdl>
int launch_evaluate(void *, const NlmpcDev *m, const NlmpcBatchDev *b, void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int wpb = nlmpc_waves_per_block(*m);
    if (wpb < 1) return -2;
    int blocks = (b->batch + wpb - 1) / wpb;
    if (blocks > 4096) blocks = 4096;
    const size_t lds = (size_t)wpb * m->lds_per_wave * sizeof(double);
    hipLaunchKernelGGL(nlmpc_evaluate<Mdl>, dim3(blocks), dim3(wpb * 64), lds, s, *m, *b);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <class Mdl>
int launch_solve(void *, const NlmpcDev *m, const NlmpcSolveDev *b, void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int wpb = nlmpc_waves_per_block(*m);
    if (wpb < 1) return -2;
    const int blocks = (b->batch + wpb - 1) / wpb;
    const size_t lds = (size_t)wpb * m->lds_per_wave * sizeof(double);
    const bool two = m->kw > m->nl;                           // the working set can outgrow the LDS factor
    const bool blk = m->lds_blocks >= 0;                      // the dynamics blocks live in the LDS slice (nlmpc_plan; small built-in systems)
    if (blk && !kSqpLdsBlocks<Mdl>) return -2;
    auto go = [&](auto kern) {
        static const bool show = getenv("MPCX_DEBUG_OCCUPANCY") != nullptr;      // testing aid (read once): resident blocks per CU as the runtime sees them
        if (show) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, wpb * 64, lds);
            fprintf(stderr, "nlmpc_sqp: %d blocks of %d wavefronts, %zu bytes of LDS each; resident per CU: %d; factor rows in LDS %d of %d; blocks in LDS %d\n",
                    blocks, wpb, lds, nb, m->nl, m->kw, (int)blk);
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(wpb * 64), lds, s, *m, *b);
    };
    if constexpr (kSqpLdsBlocks<Mdl>) {
        // (the plan puts the blocks in LDS only where the whole factor fits too, so the product build carries no two-level instantiation of
        // that form.  It is not wrong -- the probe build runs it bit-identical to the workspace form, profiles/r04_probe_blk_two_level.txt --
        // it is slower: the larger slice costs two of eight wavefronts per CU)
        if (blk) {
#ifdef MPCX_NL_BLK_TWO_LEVEL                                  // (make blk2: the build of tools/micro/blk_two_level.sh)
            if (two) go(nlmpc_sqp<Mdl, true, true>); else go(nlmpc_sqp<Mdl, false, true>);
#else
            if (two) return -2;
            go(nlmpc_sqp<Mdl, false, true>);
#endif
        }
        else { if (two) go(nlmpc_sqp<Mdl, true, false>); else go(nlmpc_sqp<Mdl, false, false>); }
    } else {
        if (two) go(nlmpc_sqp<Mdl, true, false>); else go(nlmpc_sqp<Mdl, false, false>);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
#endif   // !__HIPCC_RTC__

}  // namespace engine
}  // namespace mpcx
