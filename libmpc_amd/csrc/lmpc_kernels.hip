// Batched linear-MPC solve kernel for gfx950 (MI355X): one MPC instance per wavefront.
//
// Replaces, for a batch of instances sharing one controller, what the reference does
// per call in LOptimizer::run (include/mpc/LMPC/LOptimizer.hpp:189-368):
//   ProblemBuilder::get        (ProblemBuilder.hpp:528-633)  -> phase 1/2: free response
//                               rollout over the horizon + adjoint pass = linear term and
//                               constraint offsets of the condensed QP
//   osqp_setup/osqp_solve      (LOptimizer.hpp:261,284)      -> phase 3/4: ADMM iterations on
//                               the condensed QP with the shared, pre-inverted ADMM matrix,
//                               and an active-set polish (OSQP's polish, iterated until the
//                               KKT conditions verify) on a Schur complement staged in LDS
//   unpacking                   (LOptimizer.hpp:305-347)     -> phase 5
// A wavefront owns an instance from load to store; nothing is exchanged between
// wavefronts, so there is no workgroup barrier anywhere in the kernel.  Vectors live
// in registers as element pairs (see lmpc_device.hpp); the vector that a mat-vec
// broadcasts is staged through the wave's LDS slice; the shared matrices stream from
// L2 with 16-byte loads, coalesced across the wavefront.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>

#include "lmpc_device.hpp"

namespace mpcx {

namespace {

constexpr int kWavesPerBlock = 4;

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
__device__ __forceinline__ double2 ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// acc += M[:, 0..ncols) * xs, M column-major with leading dimension ld, rows < R (even)
template <int CP>
__device__ __forceinline__ void matvec_acc(const double *__restrict__ M, int ld, int R, int ncols,
                                           const double *xs, double (&acc)[2 * CP], int lane)
{
#pragma unroll 4
    for (int j = 0; j < ncols; ++j) {
        const double xj = xs[j];
        const double *col = M + (size_t)j * ld;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const int e = 128 * c + 2 * lane;
            if (e < R) {
                const double2 m = ld2(col + e);
                acc[2 * c] = fma(m.x, xj, acc[2 * c]);
                acc[2 * c + 1] = fma(m.y, xj, acc[2 * c + 1]);
            }
        }
    }
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

__device__ __forceinline__ double ref_at(const double *p, long bs, long ks, int b, int k, int a)
{
    return p[(size_t)b * bs + (size_t)k * ks + a];
}

__device__ __forceinline__ bool violates(double v, double lo, double hi, double ea, double er)
{
    // same slack OSQP's primal tolerance would grant a fixed row
    return (v < lo - (ea + er * fabs(lo))) || (v > hi + (ea + er * fabs(hi)));
}

template <int CPZ, int CPG>
__device__ void solve_one(const LmpcDev &M, const LmpcBatchDev &Bt, const int b, const int lane,
                          double *stage, double *nt0, double *arena)
{
    constexpr int NZS = 2 * CPZ, NGS = 2 * CPG;
    const int nx = M.nx, nu = M.nu, ny = M.ny, ndu = M.ndu, ph = M.ph;
    const int nz = M.nz, mg = M.mg, ldz = M.ldz, ldg = M.ldg, ldy = M.ldy;
    const double INF = __builtin_huge_val();
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // ------------------------------------------------------------------ phase 0: inputs
    double *xb = arena;                       // free response, (ph+1) x nx
    double *ey = xb + (ph + 1) * nx;          // weighted output error, (ph+1) x ny
    double *pv = ey + (ph + 1) * ny;          // adjoint ping-pong, 2 x nx
    double *u0s = pv + 2 * nx;                // lastU
    if (lane < nx) xb[lane] = Bt.x0[(size_t)b * nx + lane];
    if (lane < nu) u0s[lane] = Bt.u0[(size_t)b * nu + lane];
    wave_sync();

    auto dm = [&](int k, int dd) -> double { return ref_at(Bt.dmeas, Bt.dmeas_bs, Bt.dmeas_ks, b, k, dd); };

    bool bad = false;
    if (lane < nx) bad |= violates(xb[lane], M.lo0x[lane], M.hi0x[lane], M.eps_abs, M.eps_rel);
    if (lane < nu) bad |= violates(u0s[lane], M.lo0u[lane], M.hi0u[lane], M.eps_abs, M.eps_rel);
    if (lane < ny) {
        double y0 = 0;
        for (int c = 0; c < nx; ++c) y0 = fma(M.C[lane + c * ny], xb[c], y0);
        if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) y0 = fma(M.Dd[lane + dd * ny], dm(0, dd), y0);
        bad |= violates(y0, M.lo0y[lane], M.hi0y[lane], M.eps_abs, M.eps_rel);
    }
    if (lane == 0) {
        double s0 = 0;
        for (int c = 0; c < nx; ++c) s0 = fma(M.sX[c], xb[c], s0);
        for (int c = 0; c < nu; ++c) s0 = fma(M.sU[c], u0s[c], s0);
        bad |= violates(s0, M.s0lo, M.s0hi, M.eps_abs, M.eps_rel);
    }

    // ------------------------------------------------------------------ phase 1: horizon rollout
    for (int i = 1; i <= ph; ++i) {
        if (lane < nx) {
            const double *xp = xb + (i - 1) * nx;
            double s = 0;
            for (int c = 0; c < nx; ++c) s = fma(M.A[lane + c * nx], xp[c], s);
            if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) s = fma(M.Bd[lane + dd * nx], dm(i - 1, dd), s);
            xb[i * nx + lane] = s;
        }
        wave_sync();
    }
    double c0 = 0;
    for (int idx = lane; idx < (ph + 1) * ny; idx += 64) {
        const int i = idx / ny, a = idx - i * ny, k = i > 0 ? i - 1 : 0;
        double cx = 0;
        for (int c = 0; c < nx; ++c) cx = fma(M.C[a + c * ny], xb[i * nx + c], cx);
        double r = ref_at(Bt.yref, Bt.yref_bs, Bt.yref_ks, b, k, a);
        if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) r -= M.Dd[a + dd * ny] * dm(k, dd);
        const double w = M.Wy[i * ny + a];
        ey[idx] = w * (cx - r);
        c0 += w * (0.5 * cx * cx - r * cx);
    }
    if (lane < nu) {
        const double u = u0s[lane];
        c0 += M.Wu[lane] * (0.5 * u * u - ref_at(Bt.uref, Bt.uref_bs, Bt.uref_ks, b, 0, lane) * u);
        c0 += M.Wdu[lane] * (0.5 * u * u + ref_at(Bt.duref, Bt.duref_bs, Bt.duref_ks, b, 0, lane) * u);
    }
    c0 = wave_sum(c0);

    // constraint rows: bounds shifted by the free response
    double lg[NGS], ug[NGS], rg[NGS];
    bool eqg[NGS];
#pragma unroll
    for (int s = 0; s < NGS; ++s) {
        const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
        lg[s] = -INF; ug[s] = INF; rg[s] = 1.0; eqg[s] = false;
        if (r < ldg) {
            const int kind = M.g_kind[r], st = M.g_step[r], cp = M.g_comp[r];
            double off;
            if (kind == 0) off = xb[st * nx + cp];
            else if (kind == 1) {
                off = 0;
                for (int c = 0; c < nx; ++c) off = fma(M.C[cp + c * ny], xb[st * nx + c], off);
                if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) off = fma(M.Dd[cp + dd * ny], dm(st - 1, dd), off);
            } else {
                off = 0;
                for (int c = 0; c < nx; ++c) off = fma(M.sX[c], xb[st * nx + c], off);
            }
            const double l0 = M.lg0[r], u0 = M.ug0[r];
            lg[s] = l0 - off; ug[s] = u0 - off; rg[s] = M.rho_g[r];
            eqg[s] = (l0 == u0);
        }
    }
    for (int idx = lane; idx < M.n_fixed; idx += 64) {
        const int kind = M.f_kind[idx], st = M.f_step[idx], cp = M.f_comp[idx];
        double v = 0;
        if (kind == 0) v = xb[st * nx + cp];
        else if (kind == 1) {
            for (int c = 0; c < nx; ++c) v = fma(M.C[cp + c * ny], xb[st * nx + c], v);
            if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) v = fma(M.Dd[cp + dd * ny], dm(st - 1, dd), v);
        } else {
            for (int c = 0; c < nx; ++c) v = fma(M.sX[c], xb[st * nx + c], v);
        }
        bad |= violates(v, M.f_lo[idx], M.f_hi[idx], M.eps_abs, M.eps_rel);
    }
    const bool infeasible0 = wave_any(bad);

    // ------------------------------------------------------------------ phase 2: adjoint pass -> linear term
    for (int e = lane; e < ldz; e += 64) stage[e] = 0.0;
    if (lane < nx) { pv[lane] = 0.0; pv[nx + lane] = 0.0; }
    wave_sync();
    {
        int cur = 0;
        for (int i = ph; i >= 1; --i) {
            const double *pin = pv + cur * nx;
            double *pout = pv + (1 - cur) * nx;
            if (lane < nx) {
                double s = 0;
                for (int a = 0; a < ny; ++a) s = fma(M.C[a + lane * ny], ey[i * ny + a], s);
                for (int a = 0; a < nx; ++a) s = fma(M.A[a + lane * nx], pin[a], s);
                pout[lane] = s;
            }
            wave_sync();
            if (lane < nu) {
                double g = 0;
                for (int a = 0; a < nx; ++a) g = fma(M.B[a + lane * nx], pout[a], g);
                g -= M.Wu[i * nu + lane] * ref_at(Bt.uref, Bt.uref_bs, Bt.uref_ks, b, i - 1, lane);
                stage[M.blk[i] * nu + lane] += g;
            }
            cur = 1 - cur;
            wave_sync();
        }
        if (lane < nu) {
            const int j = lane;
            stage[M.blk[1] * nu + j] -= M.Wdu[j] * (u0s[j] + ref_at(Bt.duref, Bt.duref_bs, Bt.duref_ks, b, 0, j));
            for (int i = 1; i < ph; ++i) {
                const int bn = M.blk[i + 1], bp = M.blk[i];
                if (bn != bp) {
                    const double t = -M.Wdu[i * nu + j] * ref_at(Bt.duref, Bt.duref_bs, Bt.duref_ks, b, i - 1, j);
                    stage[bn * nu + j] += t;
                    stage[bp * nu + j] -= t;
                }
            }
        }
        wave_sync();
    }
    double f[NZS], lw[NZS], uw[NZS], rb[NZS];
    bool eqb[NZS];
#pragma unroll
    for (int s = 0; s < NZS; ++s) {
        const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
        f[s] = 0; lw[s] = -INF; uw[s] = INF; rb[s] = 0; eqb[s] = false;
        if (e < ldz) {
            f[s] = stage[e]; lw[s] = M.lw[e]; uw[s] = M.uw[e]; rb[s] = M.rho_b[e];
            eqb[s] = (lw[s] == uw[s]);
        }
    }
    wave_sync();

    // ------------------------------------------------------------------ phase 3: unconstrained optimum
    {
        double nf_[NZS];
#pragma unroll
        for (int s = 0; s < NZS; ++s) nf_[s] = -f[s];
        stage_store<CPZ>(stage, nf_, ldz, lane);
        wave_sync();
    }
    double t0[NZS], gt0[NGS];
#pragma unroll
    for (int s = 0; s < NZS; ++s) t0[s] = 0;
#pragma unroll
    for (int s = 0; s < NGS; ++s) gt0[s] = 0;
    matvec_acc<CPZ>(M.Y, ldy, ldz, nz, stage, t0, lane);
    matvec_acc<CPG>(M.Y + ldz, ldy, ldg, nz, stage, gt0, lane);
    stage_store<CPZ>(nt0, t0, ldz, lane);
    stage_store<CPG>(nt0 + ldz, gt0, ldg, lane);
    wave_sync();

    // ADMM state
    double x[NZS], zb[NZS], yb[NZS], dyb[NZS], zg[NGS], yg[NGS], dyg[NGS];
    int actb[NZS], actg[NGS], posb[NZS], posg[NGS];
    const double ptol = 1e-8;
#pragma unroll
    for (int s = 0; s < NZS; ++s) {
        x[s] = t0[s]; zb[s] = clampd(t0[s], lw[s], uw[s]); yb[s] = 0; dyb[s] = 0; posb[s] = 0;
        actb[s] = eqb[s] ? 1 : (t0[s] < lw[s] - ptol * fmax(1.0, fabs(lw[s])) ? -1 : (t0[s] > uw[s] + ptol * fmax(1.0, fabs(uw[s])) ? 1 : 0));
    }
#pragma unroll
    for (int s = 0; s < NGS; ++s) {
        zg[s] = clampd(gt0[s], lg[s], ug[s]); yg[s] = 0; dyg[s] = 0; posg[s] = 0;
        actg[s] = eqg[s] ? 1 : (gt0[s] < lg[s] - ptol * fmax(1.0, fabs(lg[s])) ? -1 : (gt0[s] > ug[s] + ptol * fmax(1.0, fabs(ug[s])) ? 1 : 0));
    }

    // polished point
    double wv[NZS], gw[NGS];
#pragma unroll
    for (int s = 0; s < NZS; ++s) wv[s] = t0[s];
#pragma unroll
    for (int s = 0; s < NGS; ++s) gw[s] = gt0[s];

    double *S = arena;                               // Schur complement, kMaxActive x kSld
    double *lam = S + kMaxActive * kSld;             // rhs -> multipliers
    double *wsb = lam + kMaxActive;                  // bound values of the working set
    double *dg0 = wsb + kMaxActive;                  // original diagonal (pivot scale)
    int *wsidx = reinterpret_cast<int *>(dg0 + kMaxActive);
    double dtol_last = 0;

    // -------- active-set polish with repair: returns true when the KKT conditions verify
    auto polish = [&](int rounds) -> bool {
        for (int rd = 0; rd < rounds; ++rd) {
            int na = 0;
#pragma unroll
            for (int s = 0; s < NZS; ++s) {
                const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
                const bool act = actb[s] != 0;
                const unsigned long long mk = __ballot(act);
                const int pos = na + __popcll(mk & lt_mask);
                posb[s] = pos;
                if (act && pos < kMaxActive) { wsidx[pos] = e; wsb[pos] = actb[s] < 0 ? lw[s] : uw[s]; }
                na += __popcll(mk);
            }
#pragma unroll
            for (int s = 0; s < NGS; ++s) {
                const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
                const bool act = actg[s] != 0;
                const unsigned long long mk = __ballot(act);
                const int pos = na + __popcll(mk & lt_mask);
                posg[s] = pos;
                if (act && pos < kMaxActive) { wsidx[pos] = ldz + r; wsb[pos] = actg[s] < 0 ? lg[s] : ug[s]; }
                na += __popcll(mk);
            }
            if (na > kMaxActive) return false;
            wave_sync();
            int dep_at = -1;
            if (na > 0) {
                for (int p = lane; p < na * na; p += 64) {
                    const int a = p / na, c = p - a * na;
                    S[a * kSld + c] = M.Y[(size_t)wsidx[a] * ldy + wsidx[c]];
                }
                if (lane < na) lam[lane] = nt0[wsidx[lane]] - wsb[lane];
                wave_sync();
                if (lane < na) dg0[lane] = S[lane * kSld + lane];
                wave_sync();
                for (int k = 0; k < na; ++k) {
                    const double d = S[k * kSld + k];
                    if (!(d > 1e-11 * dg0[k])) { dep_at = k; break; }
                    const double sd = sqrt(d);
                    if (lane > k && lane < na) S[lane * kSld + k] /= sd;
                    if (lane == k) S[k * kSld + k] = sd;
                    wave_sync();
                    if (lane > k && lane < na) {
                        const double lik = S[lane * kSld + k];
                        for (int j = k + 1; j <= lane; ++j) S[lane * kSld + j] -= lik * S[j * kSld + k];
                    }
                    wave_sync();
                }
                if (dep_at < 0) {
                    for (int k = 0; k < na; ++k) {
                        const double yk = lam[k] / S[k * kSld + k];
                        wave_sync();
                        if (lane == k) lam[k] = yk;
                        if (lane > k && lane < na) lam[lane] -= S[lane * kSld + k] * yk;
                        wave_sync();
                    }
                    for (int k = na - 1; k >= 0; --k) {
                        const double lk = lam[k] / S[k * kSld + k];
                        wave_sync();
                        if (lane == k) lam[k] = lk;
                        if (lane < k) lam[lane] -= S[k * kSld + lane] * lk;
                        wave_sync();
                    }
                }
            }
            if (dep_at >= 0) {
                // linearly dependent working set: drop the offending row and try again
                const int q = wsidx[dep_at];
#pragma unroll
                for (int s = 0; s < NZS; ++s)
                    if (128 * (s >> 1) + 2 * lane + (s & 1) == q) actb[s] = 0;
#pragma unroll
                for (int s = 0; s < NGS; ++s)
                    if (ldz + 128 * (s >> 1) + 2 * lane + (s & 1) == q) actg[s] = 0;
                wave_sync();
                continue;
            }
#pragma unroll
            for (int s = 0; s < NZS; ++s) wv[s] = t0[s];
#pragma unroll
            for (int s = 0; s < NGS; ++s) gw[s] = gt0[s];
            double lmax = 0;
            for (int a = 0; a < na; ++a) {
                const double la = lam[a];
                lmax = fmax(lmax, fabs(la));
                const double *row = M.Y + (size_t)wsidx[a] * ldy;
#pragma unroll
                for (int c = 0; c < CPZ; ++c) {
                    const int e = 128 * c + 2 * lane;
                    if (e < ldz) { const double2 m = ld2(row + e); wv[2 * c] = fma(-la, m.x, wv[2 * c]); wv[2 * c + 1] = fma(-la, m.y, wv[2 * c + 1]); }
                }
#pragma unroll
                for (int c = 0; c < CPG; ++c) {
                    const int r = 128 * c + 2 * lane;
                    if (r < ldg) { const double2 m = ld2(row + ldz + r); gw[2 * c] = fma(-la, m.x, gw[2 * c]); gw[2 * c + 1] = fma(-la, m.y, gw[2 * c + 1]); }
                }
            }
            const double dtol = 1e-9 * lmax + 1e-300;
            dtol_last = dtol;
            bool changed = false, nanv = false;
#pragma unroll
            for (int s = 0; s < NZS; ++s) {
                nanv |= !(wv[s] == wv[s]);
                if (actb[s] == 0) {
                    if (wv[s] < lw[s] - ptol * fmax(1.0, fabs(lw[s]))) { actb[s] = -1; changed = true; }
                    else if (wv[s] > uw[s] + ptol * fmax(1.0, fabs(uw[s]))) { actb[s] = 1; changed = true; }
                } else if (!eqb[s]) {
                    const double l = lam[posb[s]];
                    if ((actb[s] < 0 && l > dtol) || (actb[s] > 0 && l < -dtol)) { actb[s] = 0; changed = true; }
                }
            }
#pragma unroll
            for (int s = 0; s < NGS; ++s) {
                nanv |= !(gw[s] == gw[s]);
                if (actg[s] == 0) {
                    if (gw[s] < lg[s] - ptol * fmax(1.0, fabs(lg[s]))) { actg[s] = -1; changed = true; }
                    else if (gw[s] > ug[s] + ptol * fmax(1.0, fabs(ug[s]))) { actg[s] = 1; changed = true; }
                } else if (!eqg[s]) {
                    const double l = lam[posg[s]];
                    if ((actg[s] < 0 && l > dtol) || (actg[s] > 0 && l < -dtol)) { actg[s] = 0; changed = true; }
                }
            }
            if (wave_any(nanv)) return false;
            if (!wave_any(changed)) return true;
            wave_sync();
        }
        return false;
    };

    // -------- one ADMM iteration (OSQP's splitting on the condensed QP)
    const double alpha = M.alpha, sigma = M.sigma;
    auto admm_iter = [&]() {
        double tmpg[NGS];
#pragma unroll
        for (int s = 0; s < NGS; ++s) tmpg[s] = rg[s] * zg[s] - yg[s];
        stage_store<CPG>(stage, tmpg, ldg, lane);
        wave_sync();
        double rhs[NZS];
#pragma unroll
        for (int s = 0; s < NZS; ++s) rhs[s] = sigma * x[s] - f[s] + (rb[s] * zb[s] - yb[s]);
        matvec_acc<CPZ>(M.Gr, ldz, ldz, mg, stage, rhs, lane);
        wave_sync();
        stage_store<CPZ>(stage, rhs, ldz, lane);
        wave_sync();
        double xt[NZS];
#pragma unroll
        for (int s = 0; s < NZS; ++s) xt[s] = 0;
        matvec_acc<CPZ>(M.Kinv, ldz, ldz, nz, stage, xt, lane);
        wave_sync();
        stage_store<CPZ>(stage, xt, ldz, lane);
        wave_sync();
        double ztg[NGS];
#pragma unroll
        for (int s = 0; s < NGS; ++s) ztg[s] = 0;
        matvec_acc<CPG>(M.Gc, ldg, ldg, nz, stage, ztg, lane);
        wave_sync();
#pragma unroll
        for (int s = 0; s < NZS; ++s) {
            x[s] = alpha * xt[s] + (1.0 - alpha) * x[s];
            if (rb[s] > 0.0) {
                const double zr = alpha * xt[s] + (1.0 - alpha) * zb[s];
                const double zn = clampd(zr + yb[s] / rb[s], lw[s], uw[s]);
                dyb[s] = rb[s] * (zr - zn);
                yb[s] += dyb[s];
                zb[s] = zn;
            }
        }
#pragma unroll
        for (int s = 0; s < NGS; ++s) {
            const double zr = alpha * ztg[s] + (1.0 - alpha) * zg[s];
            const double zn = clampd(zr + yg[s] / rg[s], lg[s], ug[s]);
            dyg[s] = rg[s] * (zr - zn);
            yg[s] += dyg[s];
            zg[s] = zn;
        }
    };

    // -------- OSQP's primal-infeasibility certificate on (delta y)
    auto certificate = [&]() -> bool {
        double pb[NZS], pg[NGS];
        double nrm = 0, lhs = 0;
#pragma unroll
        for (int s = 0; s < NZS; ++s) {
            double d = dyb[s];
            const bool iu = !(uw[s] < INF), il = !(lw[s] > -INF);
            if (iu && il) d = 0; else if (iu) d = fmin(d, 0.0); else if (il) d = fmax(d, 0.0);
            pb[s] = d; nrm = fmax(nrm, fabs(d));
            if (d > 0) lhs += uw[s] * d; else if (d < 0) lhs += lw[s] * d;
        }
#pragma unroll
        for (int s = 0; s < NGS; ++s) {
            double d = dyg[s];
            const bool iu = !(ug[s] < INF), il = !(lg[s] > -INF);
            if (iu && il) d = 0; else if (iu) d = fmin(d, 0.0); else if (il) d = fmax(d, 0.0);
            pg[s] = d; nrm = fmax(nrm, fabs(d));
            if (d > 0) lhs += ug[s] * d; else if (d < 0) lhs += lg[s] * d;
        }
        nrm = wave_max(nrm);
        lhs = wave_sum(lhs);
        if (!(nrm > 1e-30)) return false;
        if (!(lhs < -M.eps_prim_inf * nrm)) return false;
        stage_store<CPG>(stage, pg, ldg, lane);
        wave_sync();
        matvec_acc<CPZ>(M.Gr, ldz, ldz, mg, stage, pb, lane);
        wave_sync();
        double n2 = 0;
#pragma unroll
        for (int s = 0; s < NZS; ++s) n2 = fmax(n2, fabs(pb[s]));
        n2 = wave_max(n2);
        return n2 < M.eps_prim_inf * nrm;
    };

    // -------- OSQP's residual test on the ADMM iterate: 0 none, 1 solved, 2 inaccurate
    auto residual_status = [&]() -> int {
        double gx[NGS], hx[NZS], aty[NZS];
#pragma unroll
        for (int s = 0; s < NGS; ++s) gx[s] = 0;
#pragma unroll
        for (int s = 0; s < NZS; ++s) { hx[s] = 0; aty[s] = yb[s]; }
        stage_store<CPZ>(stage, x, ldz, lane);
        wave_sync();
        matvec_acc<CPG>(M.Gc, ldg, ldg, nz, stage, gx, lane);
        matvec_acc<CPZ>(M.H, ldz, ldz, nz, stage, hx, lane);
        wave_sync();
        stage_store<CPG>(stage, yg, ldg, lane);
        wave_sync();
        matvec_acc<CPZ>(M.Gr, ldz, ldz, mg, stage, aty, lane);
        wave_sync();
        double pr = 0, pn = 0, dr = 0, dn = 0;
#pragma unroll
        for (int s = 0; s < NZS; ++s) {
            if (rb[s] > 0.0) { pr = fmax(pr, fabs(x[s] - zb[s])); pn = fmax(pn, fmax(fabs(x[s]), fabs(zb[s]))); }
            dr = fmax(dr, fabs(hx[s] + f[s] + aty[s]));
            dn = fmax(dn, fmax(fabs(hx[s]), fmax(fabs(f[s]), fabs(aty[s]))));
        }
#pragma unroll
        for (int s = 0; s < NGS; ++s) {
            pr = fmax(pr, fabs(gx[s] - zg[s]));
            pn = fmax(pn, fmax(fabs(gx[s]), fabs(zg[s])));
        }
        pr = wave_max(pr); pn = wave_max(pn); dr = wave_max(dr); dn = wave_max(dn);
        if (pr < M.eps_abs + M.eps_rel * pn && dr < M.eps_abs + M.eps_rel * dn) return 1;
        if (pr < 10 * (M.eps_abs + M.eps_rel * pn) && dr < 10 * (M.eps_abs + M.eps_rel * dn)) return 2;
        return 0;
    };

    // ------------------------------------------------------------------ phase 4: solve
    int iters = 0;
    bool solved = false, polished = false, infeasible = infeasible0;
    int solver_status = -10;
    if (!infeasible) {
        if (M.polish) { solved = polish(M.polish_rounds0); polished = solved; }
        while (!solved && iters < M.max_iter) {
            const int nblk = min(M.check_every, M.max_iter - iters);
            for (int k = 0; k < nblk; ++k) admm_iter();
            iters += nblk;
            if (certificate()) { infeasible = true; break; }
            if (M.polish) {
#pragma unroll
                for (int s = 0; s < NZS; ++s)
                    actb[s] = (rb[s] > 0.0) ? (eqb[s] ? 1 : ((zb[s] - lw[s] < -yb[s]) ? -1 : ((uw[s] - zb[s] < yb[s]) ? 1 : 0))) : 0;
#pragma unroll
                for (int s = 0; s < NGS; ++s)
                    actg[s] = eqg[s] ? 1 : ((zg[s] - lg[s] < -yg[s]) ? -1 : ((ug[s] - zg[s] < yg[s]) ? 1 : 0));
                solved = polish(M.polish_rounds);
                polished = solved;
            } else {
                solved = residual_status() == 1;
            }
        }
        if (infeasible) solver_status = -3;
        else if (solved) solver_status = 1;
        else {
            const int rs = residual_status();
            solver_status = rs == 1 ? 1 : (rs == 2 ? 2 : -2);
        }
    } else {
        solver_status = -3;
    }

    // ------------------------------------------------------------------ phase 5: unpack
    double w[NZS];
#pragma unroll
    for (int s = 0; s < NZS; ++s) w[s] = polished ? wv[s] : x[s];
    const double qnan = __builtin_nan("");
    double cost;
    if (infeasible) {
#pragma unroll
        for (int s = 0; s < NZS; ++s) w[s] = qnan;
    }
    wave_sync();
    stage_store<CPZ>(stage, w, ldz, lane);      // stays staged for the sequence roll-out below
    wave_sync();
    if (infeasible) {
        cost = 1e30;
    } else {
        double hw[NZS];
#pragma unroll
        for (int s = 0; s < NZS; ++s) hw[s] = 0;
        matvec_acc<CPZ>(M.H, ldz, ldz, nz, stage, hw, lane);
        double j = 0;
#pragma unroll
        for (int s = 0; s < NZS; ++s) j += w[s] * (0.5 * hw[s] + f[s]);
        cost = wave_sum(j) + c0;
    }
#pragma unroll
    for (int s = 0; s < NZS; ++s) {
        const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
        if (e < nu) Bt.cmd[(size_t)b * nu + e] = w[s];
    }
    if (lane == 0) {
        if (Bt.cost) Bt.cost[b] = cost;
        if (Bt.solver_status) Bt.solver_status[b] = solver_status;
        if (Bt.status) {
            // LOptimizer.hpp:386-415
            int st = 4;
            if (solver_status == 1 || solver_status == 2) st = 0;
            else if (solver_status == -2) st = 1;
            else if (solver_status == -3) st = 2;
            Bt.status[b] = st;
        }
        if (Bt.is_feasible) Bt.is_feasible[b] = (solver_status == 1 || solver_status == 2 || solver_status == -2) ? 1 : 0;
        if (Bt.iterations) Bt.iterations[b] = iters;
    }

    if (Bt.active_lower && Bt.active_upper) {
        // bits assembled in LDS, written out as whole words
        wave_sync();
        unsigned *bl = reinterpret_cast<unsigned *>(nt0);
        unsigned *bu = bl + M.active_words;
        for (int wd = lane; wd < 2 * M.active_words; wd += 64) bl[wd] = 0u;
        wave_sync();
        if (!infeasible) {
#pragma unroll
            for (int s = 0; s < NZS; ++s) {
                const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
                if (e >= nz) continue;
                int side = 0;
                if (polished) {
                    if (actb[s] != 0) {
                        const double l = lam[posb[s]];
                        if (fabs(l) > dtol_last) side = l < 0 ? -1 : 1;
                    }
                } else if (rb[s] > 0.0) {
                    side = (zb[s] - lw[s] < -yb[s]) ? -1 : ((uw[s] - zb[s] < yb[s]) ? 1 : 0);
                }
                if (side == 0) continue;
                for (int p = M.boxrow_ptr[e]; p < M.boxrow_ptr[e + 1]; ++p) {
                    const int rr = M.boxrow_ref[p];
                    if (side < 0 && M.boxrow_lo[p] == lw[s]) atomicOr(&bl[rr >> 5], 1u << (rr & 31));
                    if (side > 0 && M.boxrow_hi[p] == uw[s]) atomicOr(&bu[rr >> 5], 1u << (rr & 31));
                }
            }
#pragma unroll
            for (int s = 0; s < NGS; ++s) {
                const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
                if (r >= mg) continue;
                int side = 0;
                if (polished) {
                    if (actg[s] != 0) {
                        const double l = lam[posg[s]];
                        if (fabs(l) > dtol_last) side = l < 0 ? -1 : 1;
                    }
                } else {
                    side = (zg[s] - lg[s] < -yg[s]) ? -1 : ((ug[s] - zg[s] < yg[s]) ? 1 : 0);
                }
                if (side == 0) continue;
                const int rr = M.g_refrow[r];
                if (side < 0) atomicOr(&bl[rr >> 5], 1u << (rr & 31));
                else atomicOr(&bu[rr >> 5], 1u << (rr & 31));
            }
        }
        wave_sync();
        for (int wd = lane; wd < M.active_words; wd += 64) {
            Bt.active_lower[(size_t)b * M.active_words + wd] = bl[wd];
            Bt.active_upper[(size_t)b * M.active_words + wd] = bu[wd];
        }
        wave_sync();
    }

    if (Bt.seq_state || Bt.seq_input || Bt.seq_output) {
        // OptSequence (LOptimizer.hpp:305-338): roll the model forward with the optimal inputs
        wave_sync();
        double *xs0 = arena, *xs1 = arena + nx;      // ping-pong state
        if (lane < nx) xs0[lane] = infeasible ? qnan : Bt.x0[(size_t)b * nx + lane];
        wave_sync();
        for (int i = 0; i <= ph; ++i) {
            const double *xc = (i & 1) ? xs1 : xs0;
            double *xn = (i & 1) ? xs0 : xs1;
            const int k = i > 0 ? i - 1 : 0;
            if (Bt.seq_state && lane < nx) Bt.seq_state[((size_t)b * (ph + 1) + i) * nx + lane] = xc[lane];
            if (Bt.seq_input && lane < nu) {
                const int ii = (i + 1 <= ph) ? i + 1 : ph;
                Bt.seq_input[((size_t)b * (ph + 1) + i) * nu + lane] = stage[M.blk[ii] * nu + lane];
            }
            if (Bt.seq_output && lane < ny) {
                double yv = 0;
                for (int c = 0; c < nx; ++c) yv = fma(M.C[lane + c * ny], xc[c], yv);
                if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) yv = fma(M.Dd[lane + dd * ny], dm(k, dd), yv);
                Bt.seq_output[((size_t)b * (ph + 1) + i) * ny + lane] = yv;
            }
            if (i < ph && lane < nx) {
                double s = 0;
                for (int c = 0; c < nx; ++c) s = fma(M.A[lane + c * nx], xc[c], s);
                for (int c = 0; c < nu; ++c) s = fma(M.B[lane + c * nx], stage[M.blk[i + 1] * nu + c], s);
                if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) s = fma(M.Bd[lane + dd * nx], dm(i, dd), s);
                xn[lane] = s;
            }
            wave_sync();
        }
    }
    wave_sync();
}

template <int CPZ, int CPG>
__global__ __launch_bounds__(kWavesPerBlock * 64, 2) void lmpc_solve_kernel(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LmpcDev &M = *Mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *ws = smem + (size_t)wave * M.lds_per_wave;
    double *stage = ws;
    double *nt0 = stage + M.stage_len;
    double *arena = nt0 + M.ldy;
    const int wpb = blockDim.x >> 6;
    for (int b = blockIdx.x * wpb + wave; b < Bt.batch; b += gridDim.x * wpb)
        solve_one<CPZ, CPG>(M, Bt, b, lane, stage, nt0, arena);
}

template <int CPZ, int CPG>
int launch_variant(const LmpcDev &m, const LmpcDev *m_dev, const LmpcBatchDev &b, hipStream_t stream)
{
    const size_t lds = (size_t)kWavesPerBlock * m.lds_per_wave * sizeof(double);
    if (lds > 160 * 1024) return -2;
    auto kern = lmpc_solve_kernel<CPZ, CPG>;
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        configured = lds;
    }
    int blocks = (b.batch + kWavesPerBlock - 1) / kWavesPerBlock;
    const int cap = 256 * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWavesPerBlock * 64), lds, stream, m_dev, b);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

int lmpc_kernel_variant(int ldz, int ldg)
{
    const int n = ldz > ldg ? ldz : ldg;
    if (n <= 128) return 1;
    if (n <= 256) return 2;
    if (n <= 512) return 4;
    return -1;
}

int lmpc_lds_per_wave(const LmpcDev &m, int *stage_len, int *arena_len)
{
    int st = m.ldz > m.ldg ? m.ldz : m.ldg;
    st = (st + 1) / 2 * 2;
    int a1 = (m.ph + 1) * (m.nx + m.ny) + 2 * m.nx + m.nu + 8;
    int a2 = kMaxActive * kSld + 3 * kMaxActive + kMaxActive;      // S, lam, wsb, dg0, wsidx (ints)
    int ar = a1 > a2 ? a1 : a2;
    ar = (ar + 1) / 2 * 2;
    // the active-set bitmaps are assembled in the nt0 slice
    int need_bits = (2 * m.active_words + 1) / 2;
    int ldy = m.ldy;
    if (need_bits > ldy) ar += (need_bits - ldy + 1) / 2 * 2;
    *stage_len = st;
    *arena_len = ar;
    return st + ldy + ar;
}

int lmpc_launch(const LmpcDev &m, const LmpcDev *m_dev, const LmpcBatchDev &b, void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (lmpc_kernel_variant(m.ldz, m.ldg)) {
    case 1: return launch_variant<1, 1>(m, m_dev, b, s);
    case 2: return launch_variant<2, 2>(m, m_dev, b, s);
    case 4: return launch_variant<4, 4>(m, m_dev, b, s);
    default: return -2;
    }
}

}  // namespace mpcx
