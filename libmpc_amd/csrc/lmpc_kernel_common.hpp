// Helpers shared by the LMPC kernel translation units (lmpc_kernels.hip, lmpc_fast.hip): address-space casts, wave-level
// primitives, the batched mat-vec, the fused record.  Everything lives in an anonymous namespace: each unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <type_traits>

#include "lmpc_device.hpp"

namespace mpcx {

namespace {

#ifndef MPCX_WAVES_PER_BLOCK
#define MPCX_WAVES_PER_BLOCK 2
#endif
constexpr int kWavesPerBlock = MPCX_WAVES_PER_BLOCK;
constexpr int kFallbackChunk = 64;      // instances one wavefront of the fallback kernel screens (one flag per lane)
// a working set with all signs right grows by the rows violated by at least this fraction of the largest violation: adding
// every violated row at once over-constrains, the surplus rows are shed one round later and the slowest instances ping-pong
// (max rounds 18-22 over six batches of 4096 with 0, 12-14 with 0.3; 0.1 and 0.5 are worse than either)
#ifndef MPCX_ADD_THETA
#define MPCX_ADD_THETA 0.3
#endif
// lean solve (solve_fast): thresholds of the first working set and of the rows that enter later, as fractions of the largest violation
// (round 6 tried 0.5: tools/activeset_sim.py over five batches of 4096 at N = 20 and one each at N = 10 / 50 has the mean number of rounds fall by 2 % at every
// horizon -- 3.512 -> 3.440, 3.548 -> 3.464, 3.579 -> 3.504; N = 50: 3.557 -> 3.485 -- and the GPU reports those counts; side by side on one box the benchmark
// batch is 1 % SLOWER with it (0.0459 vs 0.0463 ms: its time is its slowest instance's, whose later working sets are larger), 32768 instances 0.8 % faster,
// config 4's shard within the noise: kept at 0.3)
#ifndef MPCX_INIT_THETA
#define MPCX_INIT_THETA 0.3
#endif
#ifndef MPCX_FAST_ADD_THETA
#define MPCX_FAST_ADD_THETA 0.2
#endif
#ifndef MPCX_SOLVE_WAVES
#define MPCX_SOLVE_WAVES 2
#endif

// Pointers that come out of the model struct are generic pointers to the compiler, which
// would emit flat_load (tied to both vmcnt and lgkmcnt, serialising against LDS traffic).
// Everything they point to lives in HBM: say so.
#define MPCX_GAS __attribute__((address_space(1)))
typedef const double MPCX_GAS *gdp;
typedef const int MPCX_GAS *gip;
typedef double MPCX_GAS *gdw;
template <typename T> __device__ __forceinline__ const T MPCX_GAS *gl(const T *p) { return (const T MPCX_GAS *)p; }
template <typename T> __device__ __forceinline__ T MPCX_GAS *glw(T *p) { return (T MPCX_GAS *)p; }
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 ld2(gdp p) { return *reinterpret_cast<const d2 MPCX_GAS *>(p); }
__device__ __forceinline__ void st2(gdw p, double a, double b)
{
    d2 v; v.x = a; v.y = b;
    *reinterpret_cast<d2 MPCX_GAS *>(p) = v;
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }
// broadcast lane l's value (l wave-uniform): two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double readlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
constexpr int kRegCap = 16;      // working sets up to this size are factored in registers
// 1/d for a positive, well-scaled pivot: hardware estimate + two Newton steps (full precision, a third of the latency of the
// IEEE division sequence, which sits on the dependent chain of every elimination step)
__device__ __forceinline__ double pivot_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// acc += M[:, 0..ncols) * xs.  M column-major, leading dimension ld, R (even) valid rows.
template <int CP, int U = (CP == 1 ? 8 : (CP == 2 ? 4 : 2))>
__device__ __forceinline__ void matvec_acc(gdp M, int ld, int R, int ncols, const double *xs,
                                           double (&acc)[2 * CP], int lane)
{
    int off[CP];
    double t[2 * CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const int e = 128 * c + 2 * lane;
        off[c] = e < R ? e : 0;
        t[2 * c] = 0; t[2 * c + 1] = 0;
    }
    // explicit software pipelining: issue a batch of column fetches, then consume them
    int j = 0;
    for (; j + U <= ncols; j += U) {
        d2 m[U][CP];
        double xj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            gdp col = M + (size_t)(j + u) * ld;
#pragma unroll
            for (int c = 0; c < CP; ++c) m[u][c] = ld2(col + off[c]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xj[u] = xs[j + u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                t[2 * c] = fma(m[u][c].x, xj[u], t[2 * c]);
                t[2 * c + 1] = fma(m[u][c].y, xj[u], t[2 * c + 1]);
            }
        }
    }
    for (; j < ncols; ++j) {
        const double xj = xs[j];
        gdp col = M + (size_t)j * ld;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const d2 m = ld2(col + off[c]);
            t[2 * c] = fma(m.x, xj, t[2 * c]);
            t[2 * c + 1] = fma(m.y, xj, t[2 * c + 1]);
        }
    }
#pragma unroll
    for (int c = 0; c < CP; ++c)
        if (128 * c + 2 * lane < R) { acc[2 * c] += t[2 * c]; acc[2 * c + 1] += t[2 * c + 1]; }
}

template <int CP>
__device__ __forceinline__ void stage_store(double *xs, const double (&v)[2 * CP], int n, int lane)
{
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const int e = 128 * c + 2 * lane;
        if (e < n) *reinterpret_cast<double2 *>(xs + e) = make_double2(v[2 * c], v[2 * c + 1]);
    }
}

__device__ __forceinline__ double ref_at(gdp p, long bs, long ks, int b, int k, int a)
{
    return p[(size_t)b * bs + (size_t)k * ks + a];
}

__device__ __forceinline__ bool violates(double v, double lo, double hi, double ea, double er)
{
    // same slack OSQP's primal tolerance would grant a fixed row
    return (v < lo - (ea + er * fabs(lo))) || (v > hi + (ea + er * fabs(hi)));
}

#define GP(field) gl(M.field)

typedef double v4d __attribute__((ext_vector_type(4)));

// =====================================================================================
// the record of one instance without the workspace: ProblemBuilder::get as ONE mat-vec
// =====================================================================================
// Everything the solve needs of an instance -- f, t0 = -Hinv f, G t0, the row offsets, the feasibility rows and the cost
// constant -- is MF * vin with vin = [x0 | lastU | yref | 1] (lmpc_model.cpp: compose_fused_maps).  The wavefront computes it
// into its own LDS slice, in the layout of the workspace record (f | t0 | gt0 | lg | ug | c0, flag), and solve_one reads it
// from there: no assemble kernel, no 2.7 KB per instance written to HBM and read back.  MF streams from L2 (87 KB at N = 20).
constexpr int kCpFused = 3;            // rows of MF per lane pair: up to 384
#ifndef MPCX_FUSED_RECORD_INLINE
#define MPCX_FUSED_RECORD_INLINE __forceinline__      // (as a real call the lean kernels' fused form faulted on its by-reference arguments)
#endif
// where the pieces of the record go (the lean kernels keep them in padded LDS arrays)
struct RecPtrs { double *f, *t0, *gt0, *lg, *ug, *tail; };
__device__ MPCX_FUSED_RECORD_INLINE void fused_record(const LmpcDev &M, const LmpcBatchDev &Bt, const int b, const int lane, double *stage,
                                                      const RecPtrs &rp, const double *mf_lds)
{
    const int nx = M.nx, nu = M.nu, ny = M.ny, kin = M.kin;
    const int ldz = M.ldz, ldg = M.ldg, ldy = M.ldy, rowsF = M.rowsF;
    const int variant = Bt.fused - 1;
    // vin: the same k -> (x0 | lastU | yref | 1) map as lmpc_assemble_mfma
    if (lane < kin) {
        const int k = lane;
        double v = 0.0;
        if (k < M.nxp) { if (k < nx) v = gl(Bt.x0)[(size_t)b * nx + k]; }
        else if (k < M.nxp + M.nup) { const int c = k - M.nxp; if (c < nu) v = gl(Bt.u0)[(size_t)b * nu + c]; }
        else if (k < M.ione) { const int c = k - M.nxp - M.nup; if (variant && c < ny) v = gl(Bt.yref)[(size_t)b * Bt.yref_bs + c]; }
        else if (k == M.ione) v = 1.0;
        stage[k] = v;
    }
    wave_sync();
    double acc[2 * kCpFused];
#pragma unroll
    for (int s = 0; s < 2 * kCpFused; ++s) acc[s] = 0.0;
    if (mf_lds) {
        // the composed map sits in this workgroup's LDS (lmpc_solve_persistent loaded it once): sixteen-byte reads, lanes on
        // consecutive rows (no bank conflicts), four columns in flight
        int off[kCpFused];
#pragma unroll
        for (int c = 0; c < kCpFused; ++c) { const int e = 128 * c + 2 * lane; off[c] = e < rowsF ? e : 0; }
        for (int j = 0; j < kin; j += 4) {
            double2 m[4][kCpFused];
            double xj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double *col = mf_lds + (size_t)(j + u) * rowsF;
#pragma unroll
                for (int c = 0; c < kCpFused; ++c) m[u][c] = *reinterpret_cast<const double2 *>(col + off[c]);
                xj[u] = stage[j + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int c = 0; c < kCpFused; ++c) {
                    acc[2 * c] = fma(m[u][c].x, xj[u], acc[2 * c]);
                    acc[2 * c + 1] = fma(m[u][c].y, xj[u], acc[2 * c + 1]);
                }
            }
        }
    } else {
        // from L2, two columns per batch: the stream is bandwidth-bound there -- every wavefront of the launch reads the same
        // 87 KB -- and deeper batches only made the burst worse (146 us against 122 us for the launch at the benchmark batch)
        matvec_acc<kCpFused>(gl(variant ? M.MF1 : M.MF0), rowsF, rowsF, kin, stage, acc, lane);
    }
    const int r_goff = ldy, r_f = r_goff + ldg, r_s = r_f + ldz, r_q = r_s + M.nsp;
    double c0p = 0.0;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < kCpFused; ++c) {
        const int e = 128 * c + 2 * lane;          // block boundaries are even: a pair never straddles two blocks
        if (e >= rowsF) continue;
        const double a0 = acc[2 * c], a1 = acc[2 * c + 1];
        if (e < ldz) {                             // t0
            *reinterpret_cast<double2 *>(rp.t0 + e) = make_double2(a0, a1);
        } else if (e < r_goff) {                   // gt0
            *reinterpret_cast<double2 *>(rp.gt0 + (e - ldz)) = make_double2(a0, a1);
        } else if (e < r_f) {                      // row offsets -> bounds of this instance
            const int r = e - r_goff;
            const d2 l0 = ld2(GP(lg0) + r), u0 = ld2(GP(ug0) + r);
            *reinterpret_cast<double2 *>(rp.lg + r) = make_double2(l0.x - a0, l0.y - a1);
            *reinterpret_cast<double2 *>(rp.ug + r) = make_double2(u0.x - a0, u0.y - a1);
        } else if (e < r_s) {                      // linear term
            *reinterpret_cast<double2 *>(rp.f + (e - r_f)) = make_double2(a0, a1);
        } else if (e < r_q) {                      // rows that do not see the inputs: pure feasibility conditions on (x0, lastU)
            const int r = e - r_s;
            if (r < M.ns) bad |= violates(a0, GP(slo)[r], GP(shi)[r], M.eps_abs, M.eps_rel);
            if (r + 1 < M.ns) bad |= violates(a1, GP(slo)[r + 1], GP(shi)[r + 1], M.eps_abs, M.eps_rel);
        } else {                                   // cost constant: vin' Qc vin / 2
            const int k = e - r_q;
            c0p = fma(0.5 * stage[k], a0, c0p);
            c0p = fma(0.5 * stage[k + 1], a1, c0p);
        }
    }
    c0p = wave_sum(c0p);
    const bool anybad = wave_any(bad);
    if (lane == 0) *reinterpret_cast<double2 *>(rp.tail) = make_double2(c0p, anybad ? 1.0 : 0.0);
    wave_sync();
}

}  // namespace

}  // namespace mpcx
