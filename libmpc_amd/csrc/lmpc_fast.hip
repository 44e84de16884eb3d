// Lean polish kernels of the batched LMPC solve (see the comment at solve_fast); launched by lmpc_launch (lmpc_kernels.hip)
// through lmpc_launch_fast.
#include "lmpc_kernel_common.hpp"
#include <cstdlib>

namespace mpcx {

namespace {

// =====================================================================================
// solve, lean form: the polish-only path of lmpc_solve, one instance per wavefront
// =====================================================================================
// Same algorithm and same outputs as solve_one<.., ADMM = false> (OSQP's polish promoted to the main iteration, a verified
// point is the exact optimum), rebuilt around what the round-2 profile showed: the kernel is bound by the number of
// instructions a wavefront issues per round and by the rounds of the slowest instances, not by arithmetic.
//   * repair rule: wrong-signed rows leave AND violated rows enter in the same round (drop-then-add doubled the rounds of the
//     hard instances); the first working set holds the rows violated at the unconstrained optimum by at least 0.3 x the
//     largest violation, later rounds add from 0.2 x.  tools/activeset_sim.py, config 2, 32768 instances: mean 5.1 -> 3.5
//     rounds, maximum 12 -> 8.
//   * the Schur system is solved by Gauss-Jordan elimination with the right-hand side as an extra column, lane i owning row i:
//     no forward / backward substitution, no transposition through LDS, no predication -- 3 instructions per eliminated entry
//     (two v_readlane, one fma) instead of 5 plus two substitutions.  On the Schur complements of config 2 / config 4 its error
//     is that of the Cholesky factorisation (6e-15 relative).
//   * the rows of Y the update w = t0 - Y[:, A] lambda needs are requested together with the Schur entries, before the
//     elimination: one trip to L2 per round instead of three; the multipliers reach the update by v_readlane, not through LDS;
//     the working-set indices are wave-uniform and live in SGPRs (scalar row addresses).
//   * wave-wide maxima by DPP row operations and four v_readlane pairs (the ds_bpermute butterfly of __shfl_xor is a chain of
//     six LDS round trips);
//   * what a round only reads -- t0, G t0, the linear term and the bounds -- stays in LDS: the wave's slice has the layout of the
//     workspace record (f | t0 | gt0 | lg | ug | c0, flag) followed by a small arena, the box bounds are shared by the
//     workgroup.  The one-chunk variant fits 128 VGPRs and 3.3 KB of LDS per wavefront: four wavefronts per SIMD, i.e. at the
//     benchmark batch every instance is resident at once (the dispatch order no longer matters).
// Working sets of more than kFastCap rows are left to the fallback kernel (none in 32768 instances of config 2, none in 8192 of
// config 4).
#ifndef MPCX_FAST_PF_MAXCAP
#define MPCX_FAST_PF_MAXCAP 12
#endif
#ifndef MPCX_FAST_RB
#define MPCX_FAST_RB 4
#endif
#ifndef MPCX_FAST_PF
#define MPCX_FAST_PF 4
#endif
constexpr int kFastCap = 16;
#ifndef MPCX_FAST_ADD_THETA_LATE
#define MPCX_FAST_ADD_THETA_LATE 0.02       // ... and from the fourth round on (the instances still open then are the ones that set the launch time)
#endif
#ifndef MPCX_FAST_SAFE_AFTER
#define MPCX_FAST_SAFE_AFTER 12       // rounds of the block repair rule before the single-exchange rule takes over
#endif

template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// maximum over the wavefront of non-NaN values, the same in every lane
__device__ __forceinline__ double wave_max_dpp(double v)
{
    v = fmax(v, dpp_mov_d<0xB1>(v));     // quad_perm [1,0,3,2]
    v = fmax(v, dpp_mov_d<0x4E>(v));     // quad_perm [2,3,0,1]
    v = fmax(v, dpp_mov_d<0x141>(v));    // row_half_mirror
    v = fmax(v, dpp_mov_d<0x140>(v));    // row_mirror: every row of 16 lanes is uniform now
    return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}

// One solve of the working-set system and the update of the primal point, working set of at most CAP (<= 16) rows.
// In: wsidx / wsb in LDS (unified indices and bound values of the na working rows), t0s = [t0 (128 CPZ) | G t0] in LDS.
// Out: lam[0..na) in LDS, wv / gw = w and G w in registers, lmax = largest |multiplier|; returns the position of a linearly
// dependent row, or -1.
// profiling aid (tools/phase_cycles.py): cycles per phase of a round, compiled in with -DMPCX_PROFILE_ROUNDS only
#ifdef MPCX_PROFILE_ROUNDS
#define MPCX_LAP(k) do { const long long now_ = (long long)__builtin_readcyclecounter(); pacc[k] += now_ - plast; plast = now_; } while (0)
#define MPCX_LAP_WAIT(k) do { __builtin_amdgcn_s_waitcnt(0); MPCX_LAP(k); } while (0)
#define MPCX_PROF_ARGS , long long (&pacc)[6], long long &plast
#define MPCX_PROF_PASS , pacc, plast
#else
#define MPCX_LAP(k) do {} while (0)
#define MPCX_LAP_WAIT(k) do {} while (0)
#define MPCX_PROF_ARGS
#define MPCX_PROF_PASS
#endif

template <int CAP, int CPZ, int CPG>
__device__ __forceinline__ int ws_solve_reg(const gdp gY, const int ldy, const int ldz, const int na, const int lane,
                                            const int *wsidx, const double *wsb, const double *t0s, double *lam,
                                            const int (&offz)[CPZ], const int (&offg)[CPG],
                                            double (&wv)[2 * CPZ], double (&gw)[2 * CPG], double &lmax MPCX_PROF_ARGS)
{
    // rows of Y requested ahead of the elimination (none for the largest working sets: their elimination needs the registers, and a
    // spilled register is HBM traffic -- scratch is memory)
    constexpr int PF = CAP > MPCX_FAST_PF_MAXCAP ? 0 : (CAP < MPCX_FAST_PF ? CAP : MPCX_FAST_PF);
    const bool real = lane < na;
    const int li = real ? lane : 0;
    const int qi = wsidx[li];
    int qc[CAP];                                    // wave-uniform: SGPRs
#pragma unroll
    for (int c = 0; c < CAP; ++c) qc[c] = __builtin_amdgcn_readfirstlane(wsidx[c < na ? c : 0]);
    double Sr[CAP];
    const int rowoff = qi * ldy;
#pragma unroll
    for (int c = 0; c < CAP; ++c) Sr[c] = (gY + qc[c])[rowoff];
    d2 pz[PF > 0 ? PF : 1][CPZ], pg[PF > 0 ? PF : 1][CPG];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const gdp row = gY + (size_t)qc[u] * ldy;
#pragma unroll
        for (int c = 0; c < CPZ; ++c) pz[u][c] = ld2(row + offz[c]);
#pragma unroll
        for (int c = 0; c < CPG; ++c) pg[u][c] = ld2(row + offg[c]);
    }
    double y = real ? t0s[qi < ldz ? qi : qi - ldz + 128 * CPZ] - wsb[li] : 0.0;       // [t0 | G t0] by unified index (padded arrays)
    MPCX_LAP_WAIT(1);                              // 1: working set to registers + every load of the round answered
    // rows and columns past na: the identity (the elimination below is straight-line code over all CAP steps)
#pragma unroll
    for (int c = 0; c < CAP; ++c) Sr[c] = (real && c < na) ? Sr[c] : (lane == c ? 1.0 : 0.0);
    double pthr = 0.0, mydinv = 1.0;               // this lane's pivot threshold (relative to its original diagonal)
#pragma unroll
    for (int c = 0; c < CAP; ++c) if (lane == c) pthr = 1e-11 * Sr[c];
    unsigned long long dep = 0ull;
    // Gauss-Jordan on [S | y], lane i = row i; a failed pivot test is recorded and looked at once, after the loop
#pragma unroll
    for (int k = 0; k < CAP; ++k) {
        dep |= __ballot(lane == k && !(Sr[k] > pthr));
        const double rinv = pivot_rcp(readlane_d(Sr[k], k));
        double pj[CAP];
#pragma unroll
        for (int j = k + 1; j < CAP; ++j) pj[j] = readlane_d(Sr[j], k);
        const double py = readlane_d(y, k);
        if (lane == k) mydinv = rinv;
        else {
            const double m = Sr[k] * rinv;
#pragma unroll
            for (int j = k + 1; j < CAP; ++j) Sr[j] = fma(-m, pj[j], Sr[j]);
            y = fma(-m, py, y);
        }
    }
    MPCX_LAP(2);                                   // 2: elimination
    if (dep != 0ull) return (int)__builtin_ctzll(dep);
    y *= mydinv;                                   // lanes >= na: 0
    if (real) lam[lane] = y;
    // w = t0 - Y[:, A] lambda
#pragma unroll
    for (int c = 0; c < CPZ; ++c) {
        const double2 v = *reinterpret_cast<const double2 *>(t0s + 128 * c + 2 * lane);
        wv[2 * c] = v.x; wv[2 * c + 1] = v.y;
    }
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const double2 v = *reinterpret_cast<const double2 *>(t0s + 128 * CPZ + 128 * c + 2 * lane);
        gw[2 * c] = v.x; gw[2 * c + 1] = v.y;
    }
    double lm = 0.0;
#pragma unroll
    for (int a = 0; a < PF; ++a) {
        const double la = readlane_d(y, a);        // rows past na: copies of row 0 with multiplier 0
        lm = fmax(lm, fabs(la));
#pragma unroll
        for (int c = 0; c < CPZ; ++c) { wv[2 * c] = fma(-la, pz[a][c].x, wv[2 * c]); wv[2 * c + 1] = fma(-la, pz[a][c].y, wv[2 * c + 1]); }
#pragma unroll
        for (int c = 0; c < CPG; ++c) { gw[2 * c] = fma(-la, pg[a][c].x, gw[2 * c]); gw[2 * c + 1] = fma(-la, pg[a][c].y, gw[2 * c + 1]); }
    }
    if constexpr (CAP > PF) {
        constexpr int RB = CAP - PF < MPCX_FAST_RB ? CAP - PF : MPCX_FAST_RB;      // rows of Y per later batch (the elimination's registers are free by now)
#pragma unroll
        for (int a0 = PF; a0 < CAP; a0 += RB) {
            if (a0 < na) {
                d2 mz[RB][CPZ], mgv[RB][CPG];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const gdp row = gY + (size_t)qc[a0 + u < CAP ? a0 + u : 0] * ldy;
#pragma unroll
                    for (int c = 0; c < CPZ; ++c) mz[u][c] = ld2(row + offz[c]);
#pragma unroll
                    for (int c = 0; c < CPG; ++c) mgv[u][c] = ld2(row + offg[c]);
                }
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    if (a0 + u < CAP) {
                        const double la = readlane_d(y, a0 + u < CAP ? a0 + u : 0);
                        lm = fmax(lm, fabs(la));
#pragma unroll
                        for (int c = 0; c < CPZ; ++c) { wv[2 * c] = fma(-la, mz[u][c].x, wv[2 * c]); wv[2 * c + 1] = fma(-la, mz[u][c].y, wv[2 * c + 1]); }
#pragma unroll
                        for (int c = 0; c < CPG; ++c) { gw[2 * c] = fma(-la, mgv[u][c].x, gw[2 * c]); gw[2 * c + 1] = fma(-la, mgv[u][c].y, gw[2 * c + 1]); }
                    }
                }
            }
        }
    }
    lmax = lm;
    return -1;
}

// the pads of a slice (written once per wavefront: nothing ever overwrites them except the solution w, whose pads are 0 too)
template <int CPZ, int CPG>
__device__ __forceinline__ void fast_init_pads(double *slice, const int ldz, const int ldg, const int lane)
{
    constexpr int ZP = 128 * CPZ, GPD = 128 * CPG;
    const double INF = __builtin_huge_val();
    double *t0s = slice, *gt0s = t0s + ZP, *lgs = gt0s + GPD, *ugs = lgs + GPD, *fs = ugs + GPD, *lam = fs + ZP + 2;
#pragma unroll
    for (int c = 0; c < CPZ; ++c) {
        const int e = 128 * c + 2 * lane;
        if (e >= ldz) { *reinterpret_cast<double2 *>(t0s + e) = make_double2(0.0, 0.0); *reinterpret_cast<double2 *>(fs + e) = make_double2(0.0, 0.0); }
    }
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const int r = 128 * c + 2 * lane;
        if (r >= ldg) {
            *reinterpret_cast<double2 *>(gt0s + r) = make_double2(0.0, 0.0);
            *reinterpret_cast<double2 *>(lgs + r) = make_double2(-INF, -INF);
            *reinterpret_cast<double2 *>(ugs + r) = make_double2(INF, INF);
        }
    }
    if (lane == 0) { lam[kFastCap] = 0.0; lam[kFastCap + 1] = 0.0; }      // where the rows outside the working set look their multiplier up
}

// this wavefront's LDS slice (doubles), ZP = 128 CPZ, GPD = 128 CPG -- every array padded to whole lane pairs so that no lane needs a
// range predicate: t0 pads 0, bounds pad -inf / +inf (a padded row is never violated, never active):
//   t0 [ZP] | gt0 [GPD] | lg [GPD] | ug [GPD] | f, later w [ZP] | c0, flag | lam [kFastCap + 2] | wsb [kFastCap + 2] | wsidx (ints) | scratch
// lwuw: the workgroup's copy of the box bounds [lw (ZP) | uw (ZP)], padded the same way
template <int CPZ, int CPG> constexpr int fast_slice_fixed() { return 2 * 128 * CPZ + 3 * 128 * CPG + 2 + 2 * (kFastCap + 2) + (kFastCap + 2) / 2 + 1; }

// SRC: where the instance's record comes from -- 0 the workspace (two-kernel path), 1 computed here by fused_record (one mat-vec
// with the composed maps), 2 already in the slice (lmpc_solve_group: the workgroup's MFMA assemble phase put it there)
template <int CPZ, int CPG, int SRC = 0>
__device__ void solve_fast(const LmpcDev &M, const LmpcBatchDev &Bt, const int b, const int lane,
                           double *slice, const double *lwuw, gdw ws, const double *mf_lds = nullptr, double *outs = nullptr, const int eqbits = 0)
{
    // eqbits (SRC = 2, lmpc_solve_group): bit s = this lane's general row s is an equality (lg0 == ug0), looked up by the caller with its other loads
    // outs (lmpc_solve_group): the instance's scalar results and its command go to this LDS record [cost, status, solver status,
    // feasible, iterations, rounds, active rows, done | cmd (nu)] instead of to HBM; the workgroup writes sixteen of them coalesced
    constexpr int NZS = 2 * CPZ, NGS = 2 * CPG, ZP = 128 * CPZ, GPD = 128 * CPG;
    const int nx = M.nx, nu = M.nu, ny = M.ny, ndu = M.ndu, ph = M.ph;
    const int nz = M.nz, mg = M.mg, ldz = M.ldz, ldg = M.ldg, ldy = M.ldy;
    const gdp gY = GP(Y);
    double *t0s = slice, *gt0s = t0s + ZP, *lgs = gt0s + GPD, *ugs = lgs + GPD, *fs = ugs + GPD, *tail = fs + ZP;
    double *lam = tail + 2, *wsb = lam + (kFastCap + 2);
    int *wsidx = reinterpret_cast<int *>(wsb + (kFastCap + 2));
    double *scratch = wsb + (kFastCap + 2) + (kFastCap + 2) / 2 + 1;
    const double *lws = lwuw, *uws = lwuw + ZP;
    const double INF = __builtin_huge_val();

    long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;          // profiling aid: start, loaded, solved, unpacked (Bt.dbg_cycles)
#ifdef MPCX_PROFILE_ROUNDS
    long long pacc[6] = {0, 0, 0, 0, 0, 0}, plast = 0;
#endif
    auto stamp = [&](long long &t) { if (Bt.dbg_cycles) t = (long long)__builtin_readcyclecounter(); };
    stamp(ts0);

    // offsets of this lane's element pairs into the rows of Y (clamped: lanes past the end re-read pair 0, their results are never used)
    int offz[CPZ], offg[CPG];
#pragma unroll
    for (int c = 0; c < CPZ; ++c) { const int e = 128 * c + 2 * lane; offz[c] = e < ldz ? e : 0; }
#pragma unroll
    for (int c = 0; c < CPG; ++c) { const int r = 128 * c + 2 * lane; offg[c] = ldz + (r < ldg ? r : 0); }

    constexpr bool FUSED = SRC != 0;                 // the workspace holds no record of this instance
    // ---- the assembled problem: the workspace record the assemble kernel left, copied into the slice, or computed in place
    if constexpr (SRC != 2) fast_init_pads<CPZ, CPG>(slice, ldz, ldg, lane);
    if constexpr (SRC == 2) {
        // nothing to do: the record is in place
    } else if constexpr (SRC == 1) {
        RecPtrs rp{fs, t0s, gt0s, lgs, ugs, tail};
        fused_record(M, Bt, b, lane, scratch, rp, mf_lds);
    } else {
#pragma unroll
        for (int c = 0; c < CPZ; ++c) {
            const int e = 128 * c + 2 * lane;
            if (e < ldz) {
                const d2 vf = ld2(ws + e), vt = ld2(ws + ldz + e);
                *reinterpret_cast<double2 *>(fs + e) = make_double2(vf.x, vf.y);
                *reinterpret_cast<double2 *>(t0s + e) = make_double2(vt.x, vt.y);
            }
        }
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const int r = 128 * c + 2 * lane;
            if (r < ldg) {
                const d2 vt = ld2(ws + 2 * ldz + r), vl = ld2(ws + ldz + ldy + r), vu = ld2(ws + ldz + ldy + ldg + r);
                *reinterpret_cast<double2 *>(gt0s + r) = make_double2(vt.x, vt.y);
                *reinterpret_cast<double2 *>(lgs + r) = make_double2(vl.x, vl.y);
                *reinterpret_cast<double2 *>(ugs + r) = make_double2(vu.x, vu.y);
            }
        }
        if (lane == 0) { const d2 t = ld2(ws + ldz + ldy + 2 * ldg); *reinterpret_cast<double2 *>(tail) = make_double2(t.x, t.y); }
        wave_sync();
    }
    const double c0 = tail[0];
    const double flag0 = tail[1];
    // a violated step-0 / input-independent row: see solve_one
    const bool fixed_violation = flag0 == 1.0;
    const bool infeasible = fixed_violation && M.strict_infeasible;

    // working-set state of this lane's rows: 0 free, -1 / +1 at the lower / upper bound, 2 equality row (always in, never shed)
    int actb[NZS], actg[NGS];
    const double ptol = 1e-8;
    auto viol = [&](double v, double lo, double hi) {
        return fmax(fmax(lo - ptol * fmax(1.0, fabs(lo)) - v, v - hi - ptol * fmax(1.0, fabs(hi))), 0.0);
    };
    {
        // first working set: equalities, and the rows the unconstrained optimum violates by >= MPCX_INIT_THETA x the largest violation
        double vb[NZS], vg[NGS], vinit = 0.0;
        bool lowb[NZS], lowg[NGS];
#pragma unroll
        for (int c = 0; c < CPZ; ++c) {
            const int e = 128 * c + 2 * lane;
            const double2 l = *reinterpret_cast<const double2 *>(lws + e), u = *reinterpret_cast<const double2 *>(uws + e);
            const double2 t = *reinterpret_cast<const double2 *>(t0s + e);
            vb[2 * c] = viol(t.x, l.x, u.x); vb[2 * c + 1] = viol(t.y, l.y, u.y);
            lowb[2 * c] = t.x < l.x; lowb[2 * c + 1] = t.y < l.y;
            actb[2 * c] = l.x == u.x ? 2 : 0; actb[2 * c + 1] = l.y == u.y ? 2 : 0;
            vinit = fmax(vinit, fmax(vb[2 * c], vb[2 * c + 1]));
        }
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const int r = 128 * c + 2 * lane;
            const double2 l = *reinterpret_cast<const double2 *>(lgs + r), u = *reinterpret_cast<const double2 *>(ugs + r);
            const double2 t = *reinterpret_cast<const double2 *>(gt0s + r);
            vg[2 * c] = viol(t.x, l.x, u.x); vg[2 * c + 1] = viol(t.y, l.y, u.y);
            lowg[2 * c] = t.x < l.x; lowg[2 * c + 1] = t.y < l.y;
            if constexpr (SRC == 2) {
                actg[2 * c] = ((eqbits >> (2 * c)) & 1) ? 2 : 0; actg[2 * c + 1] = ((eqbits >> (2 * c + 1)) & 1) ? 2 : 0;
            } else {
                const d2 l0 = ld2(GP(lg0) + (offg[c] - ldz)), u0 = ld2(GP(ug0) + (offg[c] - ldz));
                const bool ok = r < ldg;
                actg[2 * c] = (ok && l0.x == u0.x) ? 2 : 0; actg[2 * c + 1] = (ok && l0.y == u0.y) ? 2 : 0;
            }
            vinit = fmax(vinit, fmax(vg[2 * c], vg[2 * c + 1]));
        }
        const double thr0 = MPCX_INIT_THETA * wave_max_dpp(vinit);
#pragma unroll
        for (int s = 0; s < NZS; ++s) actb[s] = (actb[s] == 0 && vb[s] > 0.0 && vb[s] >= thr0) ? (lowb[s] ? -1 : 1) : actb[s];
#pragma unroll
        for (int s = 0; s < NGS; ++s) actg[s] = (actg[s] == 0 && vg[s] > 0.0 && vg[s] >= thr0) ? (lowg[s] ? -1 : 1) : actg[s];
    }
    stamp(ts1);   // 1: loaded

    if (Bt.warm_lower) {
        // warm start: the first working set is the previous solve's active set (see solve_one)
        const unsigned MPCX_GAS *wl = gl(Bt.warm_lower) + (size_t)b * M.active_words;
        const unsigned MPCX_GAS *wu = gl(Bt.warm_upper) + (size_t)b * M.active_words;
        const int na = M.nx + M.nu, n1 = M.ph + 1;
        const int b_box = M.neq_ref, b_out = b_box + n1 * na, b_du = b_out + n1 * M.ny, b_sc = b_du + M.ph * M.nu;
        auto look = [&](int rr) {
            if (!Bt.warm_shift) return rr;
            if (rr < b_out) return rr + na < b_out ? rr + na : rr;
            if (rr < b_du) return rr + M.ny < b_du ? rr + M.ny : rr;
            if (rr < b_sc) return rr;
            return rr + 1 < M.m_ref ? rr + 1 : rr;
        };
#pragma unroll
        for (int s = 0; s < NZS; ++s) {
            const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
            if (e >= nz || actb[s] == 2) continue;
            const double lo = lws[e], hi = uws[e];
            int side = 0;
            for (int p = GP(boxrow_ptr)[e]; p < GP(boxrow_ptr)[e + 1]; ++p) {
                const int rr = look(GP(boxrow_ref)[p]);
                if (((wl[rr >> 5] >> (rr & 31)) & 1u) && GP(boxrow_lo)[p] == lo) side = -1;
                if (((wu[rr >> 5] >> (rr & 31)) & 1u) && GP(boxrow_hi)[p] == hi) side = 1;
            }
            actb[s] = side;
        }
#pragma unroll
        for (int s = 0; s < NGS; ++s) {
            const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
            if (r >= mg || actg[s] == 2) continue;
            const int rr = look(GP(g_refrow)[r]);
            actg[s] = ((wl[rr >> 5] >> (rr & 31)) & 1u) ? -1 : (((wu[rr >> 5] >> (rr & 31)) & 1u) ? 1 : 0);
        }
    }

    double wv[NZS], gw[NGS];
    int posb[NZS], posg[NGS];                        // position in the working set (kFastCap: not in it)
    double dtol_last = 0;
    int na_last = 0, rounds_total = 0;
    bool solved = false;

    if (!infeasible && M.polish) {
        const int rounds = M.polish_rounds0;
        for (int rd = 0; rd < rounds; ++rd) {
            ++rounds_total;
#ifdef MPCX_PROFILE_ROUNDS
            plast = (long long)__builtin_readcyclecounter();
#endif
            // A launch lasts as long as its slowest instance, and the slow ones are those that need many rounds: from the fourth
            // round on a wavefront asks the instruction arbiter for precedence over its three neighbours on the SIMD.
            if (rd == 3) __builtin_amdgcn_s_setprio(1);
            else if (rd == 5) __builtin_amdgcn_s_setprio(2);
            else if (rd == 7) __builtin_amdgcn_s_setprio(3);
            // ---- the working set, in row order, to LDS: unified index and bound value per row.  Straight-line code: a row that
            // is not in the set writes to the spare slot kFastCap (so does a row past the capacity, which ends the solve below).
            int na = 0;
#pragma unroll
            for (int c = 0; c < CPZ; ++c) {
                const int e = 128 * c + 2 * lane;
                const double2 l = *reinterpret_cast<const double2 *>(lws + e), u = *reinterpret_cast<const double2 *>(uws + e);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int s = 2 * c + h;
                    const bool act = actb[s] != 0;
                    const unsigned long long mk = __ballot(act);
                    int pos = na + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                    pos = (act && pos < kFastCap) ? pos : kFastCap;
                    posb[s] = pos;
                    wsidx[pos] = e + h;
                    wsb[pos] = actb[s] < 0 ? (h ? l.y : l.x) : (h ? u.y : u.x);
                    na += __popcll(mk);
                }
            }
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                const int r = 128 * c + 2 * lane;
                const double2 l = *reinterpret_cast<const double2 *>(lgs + r), u = *reinterpret_cast<const double2 *>(ugs + r);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int s = 2 * c + h;
                    const bool act = actg[s] != 0;
                    const unsigned long long mk = __ballot(act);
                    int pos = na + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                    pos = (act && pos < kFastCap) ? pos : kFastCap;
                    posg[s] = pos;
                    wsidx[pos] = ldz + r + h;
                    wsb[pos] = actg[s] < 0 ? (h ? l.y : l.x) : (h ? u.y : u.x);
                    na += __popcll(mk);
                }
            }
            if (na > kFastCap) break;                // left to the fallback kernel
            na_last = na;
            wave_sync();
            MPCX_LAP_WAIT(0);                        // 0: working set built and in LDS
            int dep_at = -1;
            double lmax = 0.0;
            if (na == 0) {
#pragma unroll
                for (int c = 0; c < CPZ; ++c) { const double2 v = *reinterpret_cast<const double2 *>(t0s + 128 * c + 2 * lane); wv[2 * c] = v.x; wv[2 * c + 1] = v.y; }
#pragma unroll
                for (int c = 0; c < CPG; ++c) { const double2 v = *reinterpret_cast<const double2 *>(gt0s + 128 * c + 2 * lane); gw[2 * c] = v.x; gw[2 * c + 1] = v.y; }
            } else if (na <= 4) dep_at = ws_solve_reg<4, CPZ, CPG>(gY, ldy, ldz, na, lane, wsidx, wsb, t0s, lam, offz, offg, wv, gw, lmax MPCX_PROF_PASS);
            else if (na <= 6) dep_at = ws_solve_reg<6, CPZ, CPG>(gY, ldy, ldz, na, lane, wsidx, wsb, t0s, lam, offz, offg, wv, gw, lmax MPCX_PROF_PASS);
            else if (na <= 8) dep_at = ws_solve_reg<8, CPZ, CPG>(gY, ldy, ldz, na, lane, wsidx, wsb, t0s, lam, offz, offg, wv, gw, lmax MPCX_PROF_PASS);
            else if (na <= 10) dep_at = ws_solve_reg<10, CPZ, CPG>(gY, ldy, ldz, na, lane, wsidx, wsb, t0s, lam, offz, offg, wv, gw, lmax MPCX_PROF_PASS);
            else if (na <= 12) dep_at = ws_solve_reg<12, CPZ, CPG>(gY, ldy, ldz, na, lane, wsidx, wsb, t0s, lam, offz, offg, wv, gw, lmax MPCX_PROF_PASS);
            else if (na <= 14) dep_at = ws_solve_reg<14, CPZ, CPG>(gY, ldy, ldz, na, lane, wsidx, wsb, t0s, lam, offz, offg, wv, gw, lmax MPCX_PROF_PASS);
            else dep_at = ws_solve_reg<kFastCap, CPZ, CPG>(gY, ldy, ldz, na, lane, wsidx, wsb, t0s, lam, offz, offg, wv, gw, lmax MPCX_PROF_PASS);
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
            wave_sync();                             // lam is in LDS
            MPCX_LAP_WAIT(3);                        // 3: w = t0 - Y[:, A] lambda, multipliers in LDS
            const double dtol = 1e-9 * lmax + 1e-300;
            dtol_last = dtol;
            // ---- how wrong is each working row's multiplier (> dtol: wrong sign), how violated each free row: straight-line code,
            // every slot looks a multiplier up (0 in the spare slot) and measures a violation, selects decide which one counts
            double badb[NZS], badg[NGS], vb[NZS], vg[NGS], vm = 0.0, bm = 0.0, chk = 0.0;
            bool lowb[NZS], lowg[NGS];
            double lmb[NZS], lmg[NGS];
#pragma unroll
            for (int s = 0; s < NZS; ++s) lmb[s] = lam[posb[s]];
#pragma unroll
            for (int s = 0; s < NGS; ++s) lmg[s] = lam[posg[s]];
#pragma unroll
            for (int c = 0; c < CPZ; ++c) {
                const int e = 128 * c + 2 * lane;
                const double2 l = *reinterpret_cast<const double2 *>(lws + e), u = *reinterpret_cast<const double2 *>(uws + e);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int s = 2 * c + h;
                    const double lo = h ? l.y : l.x, hi = h ? u.y : u.x;
                    chk += wv[s];
                    lowb[s] = wv[s] < lo;
                    const double v = viol(wv[s], lo, hi);
                    vb[s] = actb[s] == 0 ? v : 0.0;
                    badb[s] = actb[s] == -1 ? lmb[s] : (actb[s] == 1 ? -lmb[s] : 0.0);
                    vm = fmax(vm, vb[s]); bm = fmax(bm, badb[s]);
                }
            }
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                const int r = 128 * c + 2 * lane;
                const double2 l = *reinterpret_cast<const double2 *>(lgs + r), u = *reinterpret_cast<const double2 *>(ugs + r);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int s = 2 * c + h;
                    const double lo = h ? l.y : l.x, hi = h ? u.y : u.x;
                    chk += gw[s];
                    lowg[s] = gw[s] < lo;
                    const double v = viol(gw[s], lo, hi);
                    vg[s] = actg[s] == 0 ? v : 0.0;
                    badg[s] = actg[s] == -1 ? lmg[s] : (actg[s] == 1 ? -lmg[s] : 0.0);
                    vm = fmax(vm, vg[s]); bm = fmax(bm, badg[s]);
                }
            }
            if (wave_any(!(chk - chk == 0.0))) break;       // a NaN or an infinity in the point: left to the fallback kernel
            MPCX_LAP(4);                             // 4: multipliers and violations
            const bool any_bad = wave_any(bm > dtol), any_viol = wave_any(vm > 0.0);
            if (!any_bad && !any_viol) { solved = true; break; }
            if (rd < MPCX_FAST_SAFE_AFTER) {
                // every wrong-signed row leaves, and the rows violated by at least MPCX_FAST_ADD_THETA x the largest violation enter
                const double thr = any_viol ? (rd < 3 ? MPCX_FAST_ADD_THETA : MPCX_FAST_ADD_THETA_LATE) * wave_max_dpp(vm) : INF;
#pragma unroll
                for (int s = 0; s < NZS; ++s) actb[s] = badb[s] > dtol ? 0 : ((vb[s] > 0.0 && vb[s] >= thr) ? (lowb[s] ? -1 : 1) : actb[s]);
#pragma unroll
                for (int s = 0; s < NGS; ++s) actg[s] = badg[s] > dtol ? 0 : ((vg[s] > 0.0 && vg[s] >= thr) ? (lowg[s] ? -1 : 1) : actg[s]);
            } else {
                // an instance that is still here (none in 40 000 of the benchmark workloads: the rule above may cycle in principle) goes
                // on with one exchange per round: the most wrong multiplier leaves, else the most violated row enters
                const double mx = wave_max_dpp(any_bad ? bm : vm);
                int slot = -1;
#pragma unroll
                for (int s = 0; s < NZS; ++s) if (slot < 0 && (any_bad ? badb[s] : vb[s]) == mx) slot = s;
#pragma unroll
                for (int s = 0; s < NGS; ++s) if (slot < 0 && (any_bad ? badg[s] : vg[s]) == mx) slot = NZS + s;
                const unsigned long long mk = __ballot(slot >= 0);
                if (slot >= 0 && lane == (int)__builtin_ctzll(mk)) {
#pragma unroll
                    for (int s = 0; s < NZS; ++s) if (slot == s) actb[s] = any_bad ? 0 : (lowb[s] ? -1 : 1);
#pragma unroll
                    for (int s = 0; s < NGS; ++s) if (slot == NZS + s) actg[s] = any_bad ? 0 : (lowg[s] ? -1 : 1);
                }
            }
            wave_sync();
            MPCX_LAP(5);                             // 5: repair
        }
    }
    __builtin_amdgcn_s_setprio(0);
    stamp(ts2);   // 2: solved

    if (!infeasible && !solved) {
        // left for the fallback kernel (flag stays 0 / 1), which reads the workspace record
        if constexpr (FUSED) {
#pragma unroll
            for (int c = 0; c < CPZ; ++c) {
                const int e = 128 * c + 2 * lane;
                if (e < ldz) {
                    const double2 vf = *reinterpret_cast<const double2 *>(fs + e), vt = *reinterpret_cast<const double2 *>(t0s + e);
                    st2(ws + e, vf.x, vf.y); st2(ws + ldz + e, vt.x, vt.y);
                }
            }
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                const int r = 128 * c + 2 * lane;
                if (r < ldg) {
                    const double2 vt = *reinterpret_cast<const double2 *>(gt0s + r), vl = *reinterpret_cast<const double2 *>(lgs + r),
                                  vu = *reinterpret_cast<const double2 *>(ugs + r);
                    st2(ws + ldz + ldz + r, vt.x, vt.y);
                    st2(ws + ldz + ldy + r, vl.x, vl.y);
                    st2(ws + ldz + ldy + ldg + r, vu.x, vu.y);
                }
            }
            if (lane == 0) st2(ws + ldz + ldy + 2 * ldg, c0, flag0);
        }
        if (outs && lane == 0) outs[7] = 0.0;
        return;
    }
    const int solver_status = infeasible ? -3 : (fixed_violation ? -2 : 1);

    // ---- unpack (LOptimizer.hpp:305-347)
    double w[NZS], f[NZS];
    const double qnan = __builtin_nan("");
#pragma unroll
    for (int c = 0; c < CPZ; ++c) {
        const double2 vf = *reinterpret_cast<const double2 *>(fs + 128 * c + 2 * lane);
        f[2 * c] = vf.x; f[2 * c + 1] = vf.y;
    }
#pragma unroll
    for (int s = 0; s < NZS; ++s) w[s] = infeasible ? qnan : ((128 * (s >> 1) + 2 * lane + (s & 1) < nz) ? wv[s] : 0.0);
    double *stage = fs;                              // the solution replaces the linear term
    double cost;
    bool cost_pending = false;
    wave_sync();
#pragma unroll
    for (int c = 0; c < CPZ; ++c) *reinterpret_cast<double2 *>(stage + 128 * c + 2 * lane) = make_double2(w[2 * c], w[2 * c + 1]);
    wave_sync();
    if (infeasible) {
        cost = 1e30;
    } else if (!M.cost_direct) {
        // cost from the multipliers: (f'w - lambda'b_A)/2 + c0 (see solve_one)
        double j = 0;
#pragma unroll
        for (int s = 0; s < NZS; ++s) j = fma(f[s], w[s], j);
        if (lane < na_last) j = fma(-lam[lane], wsb[lane], j);
        cost = 0.5 * wave_sum(j) + c0;
    } else if (!FUSED && Bt.n_models <= 0) {
        // cost from its definition by lmpc_cost_mfma: the solution goes to the workspace in t0's place
#pragma unroll
        for (int c = 0; c < CPZ; ++c) {
            const int e = 128 * c + 2 * lane;
            if (e < ldz) st2(ws + ldz + e, w[2 * c], w[2 * c + 1]);
        }
        cost = 0.0;
        cost_pending = true;
    } else {
        double hw[NZS];
#pragma unroll
        for (int s = 0; s < NZS; ++s) hw[s] = 0;
        matvec_acc<CPZ>(GP(H), ldz, ldz, nz, stage, hw, lane);
        double j = 0;
#pragma unroll
        for (int s = 0; s < NZS; ++s) j += w[s] * (0.5 * hw[s] + f[s]);
        cost = wave_sum(j) + c0;
    }
#pragma unroll
    for (int s = 0; s < NZS; ++s) {
        const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
        if (e < nu) { if (outs) outs[8 + e] = w[s]; else glw(Bt.cmd)[(size_t)b * nu + e] = w[s]; }
    }
    if (outs) {
        if (lane == 0) {
            outs[0] = cost; outs[1] = solver_status == 1 ? 0 : (solver_status == -2 ? 1 : 2); outs[2] = solver_status;
            outs[3] = solver_status == -3 ? 0 : 1; outs[4] = 0; outs[5] = rounds_total; outs[6] = infeasible ? 0 : na_last; outs[7] = 2.0;
        }
    } else if (lane == 0) {
        if (Bt.cost && !cost_pending) glw(Bt.cost)[b] = cost;
        if (Bt.solver_status) glw(Bt.solver_status)[b] = solver_status;
        if (Bt.status) glw(Bt.status)[b] = solver_status == 1 ? 0 : (solver_status == -2 ? 1 : 2);     // LOptimizer.hpp:386-415
        if (Bt.is_feasible) glw(Bt.is_feasible)[b] = solver_status == -3 ? 0 : 1;
        if (Bt.iterations) glw(Bt.iterations)[b] = 0;
        if (Bt.polish_rounds) glw(Bt.polish_rounds)[b] = rounds_total;
        if (Bt.active_count) glw(Bt.active_count)[b] = infeasible ? 0 : na_last;
    }

    if (Bt.active_lower && Bt.active_upper) {
        // bits assembled in LDS (in t0's place: no longer needed), written out as whole words
        wave_sync();
        unsigned *bl = reinterpret_cast<unsigned *>(t0s);
        unsigned *bu = bl + M.active_words;
        for (int wd = lane; wd < 2 * M.active_words; wd += 64) bl[wd] = 0u;
        wave_sync();
        if (!infeasible) {
#pragma unroll
            for (int s = 0; s < NZS; ++s) {
                const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
                if (e >= nz || actb[s] == 0) continue;
                const double l = lam[posb[s]];
                if (!(fabs(l) > dtol_last)) continue;
                const int side = l < 0 ? -1 : 1;
                const double lo = lws[e], hi = uws[e];
                for (int p = GP(boxrow_ptr)[e]; p < GP(boxrow_ptr)[e + 1]; ++p) {
                    const int rr = GP(boxrow_ref)[p];
                    if (side < 0 && GP(boxrow_lo)[p] == lo) atomicOr(&bl[rr >> 5], 1u << (rr & 31));
                    if (side > 0 && GP(boxrow_hi)[p] == hi) atomicOr(&bu[rr >> 5], 1u << (rr & 31));
                }
            }
#pragma unroll
            for (int s = 0; s < NGS; ++s) {
                const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
                if (r >= mg || actg[s] == 0) continue;
                const double l = lam[posg[s]];
                if (!(fabs(l) > dtol_last)) continue;
                const int rr = GP(g_refrow)[r];
                if (l < 0) atomicOr(&bl[rr >> 5], 1u << (rr & 31));
                else atomicOr(&bu[rr >> 5], 1u << (rr & 31));
            }
        }
        wave_sync();
        for (int wd = lane; wd < M.active_words; wd += 64) {
            glw(Bt.active_lower)[(size_t)b * M.active_words + wd] = bl[wd];
            glw(Bt.active_upper)[(size_t)b * M.active_words + wd] = bu[wd];
        }
        wave_sync();
    }

    if (Bt.seq_state || Bt.seq_input || Bt.seq_output) {
        // OptSequence (LOptimizer.hpp:305-338): roll the model forward with the optimal inputs
        wave_sync();
        const gdp gA = GP(A), gB = GP(B), gC = GP(C), gBd = GP(Bd), gDd = GP(Dd), gdm = gl(Bt.dmeas ? Bt.dmeas : M.dmeas_s);
        const gip gblk = GP(blk);
        auto dm = [&](int k, int dd) -> double { return ref_at(gdm, Bt.dmeas_bs, Bt.dmeas_ks, b, k, dd); };
        double *xs0 = scratch, *xs1 = scratch + nx;      // ping-pong state
        if (lane < nx) xs0[lane] = infeasible ? qnan : gl(Bt.x0)[(size_t)b * nx + lane];
        wave_sync();
        for (int i = 0; i <= ph; ++i) {
            const double *xc = (i & 1) ? xs1 : xs0;
            double *xn = (i & 1) ? xs0 : xs1;
            const int k = i > 0 ? i - 1 : 0;
            if (Bt.seq_state && lane < nx) glw(Bt.seq_state)[((size_t)b * (ph + 1) + i) * nx + lane] = xc[lane];
            if (Bt.seq_input && lane < nu) {
                const int ii = (i + 1 <= ph) ? i + 1 : ph;
                glw(Bt.seq_input)[((size_t)b * (ph + 1) + i) * nu + lane] = stage[gblk[ii] * nu + lane];
            }
            if (Bt.seq_output && lane < ny) {
                double yv = 0;
                for (int c = 0; c < nx; ++c) yv = fma(gC[lane + c * ny], xc[c], yv);
                if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) yv = fma(gDd[lane + dd * ny], dm(k, dd), yv);
                glw(Bt.seq_output)[((size_t)b * (ph + 1) + i) * ny + lane] = yv;
            }
            if (i < ph && lane < nx) {
                double s = 0;
                for (int c = 0; c < nx; ++c) s = fma(gA[lane + c * nx], xc[c], s);
                for (int c = 0; c < nu; ++c) s = fma(gB[lane + c * nx], stage[gblk[i + 1] * nu + c], s);
                if (M.has_dist) for (int dd = 0; dd < ndu; ++dd) s = fma(gBd[lane + dd * nx], dm(i, dd), s);
                xn[lane] = s;
            }
            wave_sync();
        }
    }
    wave_sync();
    if (lane == 0 && !outs) ws[ldz + ldy + 2 * ldg + 1] = cost_pending ? 3.0 : 2.0;     // 2: done, the fallback kernel skips it; 3: lmpc_cost_mfma first
    stamp(ts3);   // 3: unpacked
    if (Bt.dbg_cycles && lane == 0)
    {
        long long *o = Bt.dbg_cycles + (size_t)b * 8;
#ifdef MPCX_PROFILE_ROUNDS
        o[0] = ts1 - ts0; o[1] = ts2 - ts1;                    // load, solve, six phases
        for (int k = 0; k < 6; ++k) o[2 + k] = pacc[k];
#else
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;        // (slots 4..6: lmpc_solve_group's own stamps)
#endif
    }
}

// waves per SIMD the lean kernels are compiled for: the one-chunk variant fits 128 VGPRs
#ifndef MPCX_FAST_WAVES1
#define MPCX_FAST_WAVES1 4
#endif
#ifndef MPCX_FAST_WAVES2
#define MPCX_FAST_WAVES2 2
#endif
template <int CPZ> constexpr int fast_waves() { return CPZ == 1 ? MPCX_FAST_WAVES1 : (CPZ == 2 ? MPCX_FAST_WAVES2 : 2); }     // (CPZ = 2 at three per SIMD: 7 % faster at config 4, but 20 spilled VGPRs = 74 MB of scratch writes per 32 768)

// LDS of the lean kernels: [lw | uw] of the workgroup (padded to whole lane pairs), then one slice per wavefront (M.fast_slice doubles)
template <int CPZ>
__device__ __forceinline__ void fast_load_box(const LmpcDev &M, double *lwuw)
{
    constexpr int ZP = 128 * CPZ;
    const double INF = __builtin_huge_val();
    for (int e = 2 * (int)threadIdx.x; e < ZP; e += 2 * (int)blockDim.x) {
        const bool ok = e < M.ldz;
        const d2 vl = ld2(GP(lw) + (ok ? e : 0)), vu = ld2(GP(uw) + (ok ? e : 0));
        *reinterpret_cast<double2 *>(lwuw + e) = ok ? make_double2(vl.x, vl.y) : make_double2(-INF, -INF);
        *reinterpret_cast<double2 *>(lwuw + ZP + e) = ok ? make_double2(vu.x, vu.y) : make_double2(INF, INF);
    }
    __syncthreads();
}

template <int CPZ, int CPG>
__global__ __launch_bounds__(kWavesPerBlock * 64, fast_waves<CPZ>()) void lmpc_solve(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt, double *wsbase)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LmpcDev &M = *Mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ZP = 128 * CPZ;
    const int wpb = blockDim.x >> 6;
    double *rec = smem + 2 * ZP + (size_t)wave * M.fast_slice;
    fast_load_box<CPZ>(M, smem);
    // one instance per wavefront, no loop over instances: the grid covers the batch (a loop makes the compiler hoist the per-lane
    // addresses of every input and output array out of it and keep them alive -- in scratch, i.e. in HBM -- across the whole solve)
    const int i = blockIdx.x * wpb + wave;
    if (i < Bt.batch) solve_fast<CPZ, CPG, 0>(M, Bt, i, lane, rec, smem, glw(wsbase) + (size_t)i * M.wsld);
}

// The same for a heterogeneous batch (mpcx_lmpc_hetero_*): every instance its own model struct, so every wavefront keeps its own
// copy of the box bounds (after the slices).  A kernel of its own: the shared-model kernel above keeps its register allocation.
template <int CPZ, int CPG>
__global__ __launch_bounds__(kWavesPerBlock * 64, fast_waves<CPZ>()) void lmpc_solve_hetero(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt, double *wsbase)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LmpcDev &M = *Mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ZP = 128 * CPZ;
    const int wpb = blockDim.x >> 6;
    double *rec = smem + 2 * ZP + (size_t)wave * M.fast_slice;
    double *box = smem + 2 * ZP + (size_t)wpb * M.fast_slice + (size_t)wave * 2 * ZP;
    const double INF = __builtin_huge_val();
    const int i = blockIdx.x * wpb + wave;
    if (i < Bt.batch) {
        const LmpcDev &Mi = Mp[lmpc_model_of(Bt, i)];
#pragma unroll
        for (int c = 0; c < CPZ; ++c) {
            const int e = 128 * c + 2 * lane;
            const bool ok = e < Mi.ldz;
            const d2 vl = ld2(gl(Mi.lw) + (ok ? e : 0)), vu = ld2(gl(Mi.uw) + (ok ? e : 0));
            *reinterpret_cast<double2 *>(box + e) = ok ? make_double2(vl.x, vl.y) : make_double2(-INF, -INF);
            *reinterpret_cast<double2 *>(box + ZP + e) = ok ? make_double2(vu.x, vu.y) : make_double2(INF, INF);
        }
        wave_sync();
        solve_fast<CPZ, CPG, 0>(Mi, Bt, i, lane, rec, box, glw(wsbase) + (size_t)i * M.wsld);
        wave_sync();
    }
}

// The same with the record computed in place (no assemble kernel, no workspace traffic): instances in batch order
template <int CPZ, int CPG>
__global__ __launch_bounds__(kWavesPerBlock * 64, fast_waves<CPZ>()) void lmpc_solve_fused(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt, double *wsbase)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LmpcDev &M = *Mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    fast_load_box<CPZ>(M, smem);
    double *rec = smem + 2 * 128 * CPZ + (size_t)wave * M.fast_slice;
    const int wpb = blockDim.x >> 6;
    const int b = blockIdx.x * wpb + wave;
    if (b < Bt.batch) solve_fast<CPZ, CPG, 1>(M, Bt, b, lane, rec, smem, glw(wsbase) + (size_t)b * M.wsld);
}

// The fused form as a persistent kernel: one workgroup of kPersistWaves wavefronts per CU loads the composed map into LDS once
// (87.5 KB at N = 20; with twelve 5.6 KB slices 157 of the 160 KB of a CU), then every wavefront pulls instances from a device
// counter until the batch is exhausted -- the record of an instance costs one pass over LDS instead of a round trip through
// HBM, nobody waits for a neighbour, and an early finisher simply takes the next instance.
constexpr int kPersistWaves = 12;
template <int CPZ, int CPG>
__global__ __launch_bounds__(kPersistWaves * 64) void lmpc_solve_persistent(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt, double *wsbase,
                                                                             int *counter)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LmpcDev &M = *Mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nmf = M.rowsF * M.kin;                  // even
    {
        const gdp src = gl(Bt.fused == 2 ? M.MF1 : M.MF0);
        for (int k = 2 * (int)threadIdx.x; k < nmf; k += 2 * (int)blockDim.x) {
            const d2 v = ld2(src + k);
            *reinterpret_cast<double2 *>(smem + k) = make_double2(v.x, v.y);
        }
    }
    double *lwuw = smem + nmf;
    fast_load_box<CPZ>(M, lwuw);
    double *rec = lwuw + 2 * 128 * CPZ + (size_t)wave * M.fast_slice;
    // The first instance of a wavefront is its own number; the rest of the batch is handed out by eight counters (one per
    // residue of the workgroup number, i.e. per XCD under the usual placement), each over every eighth instance: a single
    // device-scope counter serves about 88 pulls per microsecond, which thousands of wavefronts starting together would queue on.
    const int nwaves = gridDim.x * kPersistWaves, shard = blockIdx.x & 7;
    int b = blockIdx.x * kPersistWaves + wave;
    while (b < Bt.batch) {
        solve_fast<CPZ, CPG, 1>(M, Bt, b, lane, rec, lwuw, glw(wsbase) + (size_t)b * M.wsld, smem);
        int n = 0;
        if (lane == 0) n = atomicAdd(counter + shard, 1);
        n = __builtin_amdgcn_readfirstlane(n);
        b = nwaves + shard + 8 * n;
    }
}

// =====================================================================================
// assemble + solve in one workgroup: sixteen instances per workgroup of sixteen wavefronts
// =====================================================================================
// Phase 1 is lmpc_assemble_mfma (lmpc_kernels.hip) on sixteen wavefronts instead of four: ProblemBuilder::get for sixteen instances
// as two small GEMMs on v_mfma_f64_16x16x4_f64 (instances = the N dimension), every row tile written straight into the LDS slice
// of its instance.  Phase 2: wavefront w solves instance w from its slice (solve_fast, SRC = 2).  No workspace record, no second
// launch: the only HBM traffic of a solve is its inputs and outputs.  One workgroup per CU; at the benchmark batch the launch is
// one workgroup deep.  A workgroup moves on when its slowest instance is done, so long batches use the two-kernel path instead.
// Round 5: the two-chunk variant too (CPZ = CPG = 2: up to 256 condensed variables -- config 4's N = 50).  Its slices are 10.9 KB, so a workgroup
// holds EIGHT instances on eight wavefronts (eight of the sixteen MFMA columns carry an instance, the others repeat the last one and store
// nothing); such a controller's cost comes from its definition, one product with H per instance inside solve_fast (SRC = 2 is a fused form).
// The kernel arguments are a kilobyte here (the model struct by value) and the compiler fetches each field where it is first used: a dozen scalar
// loads on different cache lines, each a miss, one after the other.  Touch every 64-byte line of the segment at once instead (one dword each, all
// requested before the one wait), after which the compiler's own loads hit the scalar cache.
template <size_t BYTES>
__device__ __forceinline__ void kernarg_touch()
{
    static_assert(BYTES <= 16 * 64 && BYTES > 15 * 64, "kernarg_touch covers sixteen lines, the sixteenth must exist");
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    int t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;
    asm volatile("s_load_dword %0, %16, 0x0\n\ts_load_dword %1, %16, 0x40\n\ts_load_dword %2, %16, 0x80\n\ts_load_dword %3, %16, 0xc0\n\t"
                 "s_load_dword %4, %16, 0x100\n\ts_load_dword %5, %16, 0x140\n\ts_load_dword %6, %16, 0x180\n\ts_load_dword %7, %16, 0x1c0\n\t"
                 "s_load_dword %8, %16, 0x200\n\ts_load_dword %9, %16, 0x240\n\ts_load_dword %10, %16, 0x280\n\ts_load_dword %11, %16, 0x2c0\n\t"
                 "s_load_dword %12, %16, 0x300\n\ts_load_dword %13, %16, 0x340\n\ts_load_dword %14, %16, 0x380\n\ts_load_dword %15, %16, 0x3c0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10), "=&s"(t11),
                   "=&s"(t12), "=&s"(t13), "=&s"(t14), "=&s"(t15)
                 : "s"(ka)
                 : "memory");
}
#ifndef MPCX_GROUP_BALANCE
#define MPCX_GROUP_BALANCE 1
#endif

template <int CPZ> constexpr int kGroupWavesOf = CPZ == 1 ? 16 : 8;
constexpr int kGroupG2 = 5;               // groups of four MFMA k-steps of the second product whose A operands are in flight together (the whole of N = 20's)
constexpr int kGroupG1 = 2;               // ... of the first product (kin = 32: the whole of it for up to 12 states, 4 inputs and 12 outputs)
// Round 6: the phase's trips to memory side by side.  A launch starts with cold caches (the controller's factors come from the memory side once per
// XCD and launch) and the phase used to chain its first-touch loads: the model struct, the box bounds, the inputs, the first product's operands,
// the row bounds of its epilogue, the second product's operands, the bounds again at the head of the solve -- seven dependent trips of 1.5-2 us
// each before the first round.  Now the model struct travels in the kernel arguments and every wavefront requests ALL of what the phase will
// touch before it waits for any of it: one trip.  The products take their operands in the old order (the same bits), from copies of the maps laid
// out as the MFMA wants them (LmpcDev::MA0p / MA1p / Ymp: a lane's operands of four k-steps are 32 contiguous bytes, a wavefront's 2 KB): the phase
// was bound by the instruction rate of the vector memory pipe -- forty-odd 8-byte loads per lane, 64 lanes on four rows each -- not by its bytes.
// Measured by cutting the kernel short (tools/group_cut.py; quadrotor N = 20, 4096 instances, us per step with the idle fallback launch): empty
// 6.9, inputs staged 7.6, first product 10.6, second product 15.3, whole 49.0 before the packed copies.
template <int CPZ, int CPG>
__global__ __launch_bounds__(kGroupWavesOf<CPZ> * 64) void lmpc_solve_group(const LmpcDev Mv, const LmpcBatchDev Bt, double *wsbase, const int variant)
{
    constexpr int kGroupWaves = kGroupWavesOf<CPZ>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
#if defined(MPCX_GROUP_CUT) && MPCX_GROUP_CUT == 0
    { if (threadIdx.x < kGroupWavesOf<CPZ> && blockIdx.x * kGroupWavesOf<CPZ> + threadIdx.x < Bt.batch) Bt.done[blockIdx.x * kGroupWavesOf<CPZ> + threadIdx.x] = 2; return; }                                   // (timing experiment, tools/group_cut.sh: what a launch costs up to here)
#endif
    kernarg_touch<sizeof(LmpcDev) + sizeof(LmpcBatchDev) + sizeof(double *) + sizeof(int)>();
    const LmpcDev &M = Mv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int nx = M.nx, nu = M.nu, ny = M.ny;
    const int kin4 = M.kin >> 2, nz4 = M.nz16 >> 2;
    const int ldz = M.ldz, ldg = M.ldg, ldy = M.ldy;
    constexpr int ZP = 128 * CPZ, GPD = 128 * CPG;
    double *lwuw = smem;
    double *slices = lwuw + 2 * ZP;
    double *Bv = slices + (size_t)kGroupWaves * M.fast_slice;      // [kin4][64]   vin as MFMA B operands
    double *Bf = Bv + (size_t)kin4 * 64;                           // [nz4][64]    f as MFMA B operands
    double *c0s = Bf + (size_t)nz4 * 64;                           // [16 wavefronts][16 instances]
    unsigned *bad = reinterpret_cast<unsigned *>(c0s + kGroupWaves * 16);   // [16]
    const int outld = 8 + ((nu + 1) & ~1);
    double *outs = c0s + kGroupWaves * 16 + 16;                             // [16][8 + nu]: results of the sixteen instances
    double *mine = slices + (size_t)wave * M.fast_slice;
    const gdp MAp = gl(variant ? M.MA1p : M.MA0p), Ymp = GP(Ymp);
    const int ntile1 = M.rowsA >> 4, tg = M.nz16 >> 4, ts = tg + (M.mg16 >> 4), tq = ts + (M.ns16 >> 4);
    const int ntile2 = M.ldy16 >> 4;
    const int G1 = (kin4 + 3) >> 2, G2 = (nz4 + 3) >> 2;          // k-step groups per row tile of the packed maps
    auto ld4 = [](gdp p) -> v4d { return *reinterpret_cast<const v4d MPCX_GAS *>(p); };
    // the slice of instance j (this lane's MFMA column; a column beyond the workgroup's instances computes a copy and stores nothing)
    const bool jlive = j < kGroupWaves;
    double *sj = slices + (size_t)(jlive ? j : 0) * M.fast_slice;
    double *t0j = sj, *gt0j = sj + ZP, *lgj = gt0j + GPD, *ugj = lgj + GPD, *fj = ugj + GPD;

    const int b0 = blockIdx.x * kGroupWaves;    // one batch of sixteen (eight) per workgroup, no loop (see lmpc_solve)
    const int bj = b0 + (jlive ? j : kGroupWaves - 1);
    const int bc = bj < Bt.batch ? bj : Bt.batch - 1;
    // profiling aid: when the wavefront started, when the first product was done, when the records were complete
    auto gstamp = [&](int k) { if (Bt.dbg_cycles && lane == 0 && b0 + wave < Bt.batch) Bt.dbg_cycles[(size_t)(b0 + wave) * 8 + k] = (long long)__builtin_readcyclecounter(); };
    gstamp(4);

    // ---- every first-touch load of the phase, requested before anything waits
    // vin operands: k-step kb holds rows 4 kb + kq of instance j (one predicated load per lane: the branches only choose the address)
    auto vin_at = [&](int kb) -> double {
        const int k = 4 * kb + kq;
        gdp src = gl(Bt.x0);
        bool ld = false;
        if (k < M.nxp) { ld = k < nx; src = gl(Bt.x0) + ((size_t)bc * nx + k); }
        else if (k < M.nxp + M.nup) { const int c = k - M.nxp; ld = c < nu; src = gl(Bt.u0) + ((size_t)bc * nu + c); }
        else if (k < M.ione) { const int c = k - M.nxp - M.nup; ld = variant && c < ny; src = gl(Bt.yref) + ((size_t)bc * Bt.yref_bs + c); }
        double v = k == M.ione ? 1.0 : 0.0;
        if (ld) v = *src;
        return v;
    };
    const double vin0 = wave < kin4 ? vin_at(wave) : 0.0;
    // the box bounds (the workgroup's copy: the first ZP / 2 threads hold a pair each)
    const int ebx = 2 * (int)threadIdx.x;
    const bool boxok = ebx < ldz;
    d2 bxl = {0.0, 0.0}, bxu = {0.0, 0.0};
    if (ebx < ZP) { bxl = ld2(GP(lw) + (boxok ? ebx : 0)); bxu = ld2(GP(uw) + (boxok ? ebx : 0)); }
    // first product: the operands of this wavefront's first tile (a wavefront without one requests the last tile's: no divergence, nothing used)
    const int t1 = wave < ntile1 ? wave : ntile1 - 1;
    v4d a1[kGroupG1];
#pragma unroll
    for (int g = 0; g < kGroupG1; ++g) a1[g] = ld4(MAp + (((size_t)t1 * G1 + (g < G1 ? g : 0)) * 64 + lane) * 4);
    // ... and what its epilogue compares with or subtracts from: four rows of (lg0, ug0) or of (slo, shi)
    double eblo[4], ebhi[4];
    {
        const bool ing = t1 >= tg && t1 < ts, ins = t1 >= ts && t1 < tq;
        const gdp plo = ing ? GP(lg0) + 16 * (t1 - tg) : GP(slo) + 16 * (ins ? t1 - ts : 0);
        const gdp phi = ing ? GP(ug0) + 16 * (t1 - tg) : GP(shi) + 16 * (ins ? t1 - ts : 0);
        const int lim = ing ? ldg - 16 * (t1 - tg) : (ins ? M.ns - 16 * (t1 - ts) : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + kq;
            eblo[r] = 0.0; ebhi[r] = 0.0;
            if (row < lim) { eblo[r] = plo[row]; ebhi[r] = phi[row]; }
        }
    }
    // second product: the operands of this wavefront's first tile
    const int t2 = wave < ntile2 ? wave : ntile2 - 1;
    v4d a2[kGroupG2];
#pragma unroll
    for (int g = 0; g < kGroupG2; ++g) a2[g] = ld4(Ymp + (((size_t)t2 * G2 + (g < G2 ? g : 0)) * 64 + lane) * 4);
    // the solve's first look at the general rows -- which are equalities (lg0 == ug0) -- is the same for every instance: the last wavefront looks
    d2 eql[CPG], equ[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const int r = 128 * c + 2 * lane;
        eql[c] = d2{0.0, 0.0}; equ[c] = d2{1.0, 1.0};
        if (wave == kGroupWaves - 1) { eql[c] = ld2(GP(lg0) + (r < ldg ? r : 0)); equ[c] = ld2(GP(ug0) + (r < ldg ? r : 0)); }
    }

    fast_init_pads<CPZ, CPG>(mine, ldz, ldg, lane);      // (the previous instance's active-set bitmaps may have run over them)
    if (wave < kin4) Bv[wave * 64 + lane] = vin0;
    for (int kb = wave + kGroupWaves; kb < kin4; kb += kGroupWaves) Bv[kb * 64 + lane] = vin_at(kb);
    if (ebx < ZP) {
        const double INF = __builtin_huge_val();
        *reinterpret_cast<double2 *>(lwuw + ebx) = boxok ? make_double2(bxl.x, bxl.y) : make_double2(-INF, -INF);
        *reinterpret_cast<double2 *>(lwuw + ZP + ebx) = boxok ? make_double2(bxu.x, bxu.y) : make_double2(INF, INF);
    }
    if (threadIdx.x < 16) bad[threadIdx.x] = 0u;
    __syncthreads();
    gstamp(7);
#if defined(MPCX_GROUP_CUT) && MPCX_GROUP_CUT == 1
    { if (threadIdx.x < kGroupWavesOf<CPZ> && blockIdx.x * kGroupWavesOf<CPZ> + threadIdx.x < Bt.batch) Bt.done[blockIdx.x * kGroupWavesOf<CPZ> + threadIdx.x] = 2; return; }
#endif

    {
        double c0p = 0.0;
        bool badl = false;
        auto tile1 = [&]<bool PRE>(const int t) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int g0 = 0; g0 < G1; g0 += kGroupG1) {
                v4d a[kGroupG1];
#pragma unroll
                for (int g = 0; g < kGroupG1; ++g) a[g] = (PRE && g0 == 0) ? a1[g] : ld4(MAp + (((size_t)t * G1 + (g0 + g < G1 ? g0 + g : g0)) * 64 + lane) * 4);
#pragma unroll
                for (int g = 0; g < kGroupG1; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kb = 4 * (g0 + g) + e;
                        if (kb < kin4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][e], Bv[kb * 64 + lane], acc, 0, 0, 0);      // (kb is wave-uniform)
                    }
            }
            if (t < tg) {
                // linear term: operand of the second product, and the instance's f
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Bf[(4 * t + r) * 64 + lane] = acc[r];
                    const int row = 16 * t + 4 * r + kq;
                    if (jlive && row < ldz) fj[row] = acc[r];
                }
            } else if (t < ts) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * (t - tg) + 4 * r + kq;
                    if (jlive && row < ldg) {
                        const double l0 = PRE ? eblo[r] : GP(lg0)[row], u0 = PRE ? ebhi[r] : GP(ug0)[row];
                        lgj[row] = l0 - acc[r]; ugj[row] = u0 - acc[r];
                    }
                }
            } else if (t < tq) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * (t - ts) + 4 * r + kq;
                    if (row < M.ns) badl |= violates(acc[r], PRE ? eblo[r] : GP(slo)[row], PRE ? ebhi[r] : GP(shi)[row], M.eps_abs, M.eps_rel);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kb2 = 4 * (t - tq) + r;
                    if (kb2 < kin4) c0p = fma(0.5 * Bv[kb2 * 64 + lane], acc[r], c0p);
                }
            }
        };
        int *eqb = reinterpret_cast<int *>(outs + kGroupWaves * outld);      // [64] the equality flags of the general rows, for every wavefront's solve
        if (wave < ntile1) tile1.template operator()<true>(wave);
        for (int t = wave + kGroupWaves; t < ntile1; t += kGroupWaves) tile1.template operator()<false>(t);
        if (badl && jlive) atomicOr(&bad[j], 1u);
        // this wavefront's share of the cost constant of instance j (no atomics: the sum must not depend on arrival order)
        c0p += __shfl_xor(c0p, 16, 64);
        c0p += __shfl_xor(c0p, 32, 64);
        if (kq == 0) c0s[wave * 16 + j] = c0p;
        __syncthreads();
        gstamp(5);
#if defined(MPCX_GROUP_CUT) && MPCX_GROUP_CUT == 2
        { if (threadIdx.x < kGroupWavesOf<CPZ> && blockIdx.x * kGroupWavesOf<CPZ> + threadIdx.x < Bt.batch) Bt.done[blockIdx.x * kGroupWavesOf<CPZ> + threadIdx.x] = 2; return; }
#endif

        auto tile2 = [&]<bool PRE>(const int t) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int g0 = 0; g0 < G2; g0 += kGroupG2) {
                v4d a[kGroupG2];
#pragma unroll
                for (int g = 0; g < kGroupG2; ++g) a[g] = (PRE && g0 == 0) ? a2[g] : ld4(Ymp + (((size_t)t * G2 + (g0 + g < G2 ? g0 + g : g0)) * 64 + lane) * 4);
#pragma unroll
                for (int g = 0; g < kGroupG2; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kb = 4 * (g0 + g) + e;
                        if (kb < nz4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][e], Bf[kb * 64 + lane], acc, 0, 0, 0);
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + kq;
                if (!jlive) continue;
                if (row < ldz) t0j[row] = acc[r];
                else if (row < ldy) gt0j[row - ldz] = acc[r];
            }
        };
        if (wave < ntile2) tile2.template operator()<true>(wave);
        for (int t = wave + kGroupWaves; t < ntile2; t += kGroupWaves) tile2.template operator()<false>(t);
        if (wave == kGroupWaves - 1) {                          // (a wavefront without a tile of the second product at the benchmark's sizes)
            int bits = 0;
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                const bool ok = 128 * c + 2 * lane < ldg;
                bits |= ((ok && eql[c].x == equ[c].x) ? 1 : 0) << (2 * c) | ((ok && eql[c].y == equ[c].y) ? 1 : 0) << (2 * c + 1);
            }
            eqb[lane] = bits;
        }
        if (threadIdx.x < kGroupWaves) {
            double *tl = slices + (size_t)threadIdx.x * M.fast_slice + 2 * ZP + 3 * GPD;
            double c0 = 0.0;
            for (int w = 0; w < kGroupWaves; ++w) c0 += c0s[w * 16 + threadIdx.x];
            tl[0] = c0;
            tl[1] = bad[threadIdx.x] ? 1.0 : 0.0;
        }
        __syncthreads();
        const int eqbits = eqb[lane];
        // ---- which wavefront solves which instance.  A launch of the benchmark batch holds every instance at once, four wavefronts on every SIMD, and it
        // ends when the busiest SIMD has issued the rounds of its four: the time of the launch is that SIMD's work (tools/group_phases.py: solves of six
        // rounds took 45 k cycles at the median and 78 k where the three neighbours were long ones too -- 25 rounds on the worst SIMD of the batch against
        // 14 on average; asking the arbiter for precedence changes nothing, measured).  How many rounds an instance will need is not known, but how far its
        // unconstrained optimum lies outside the bounds says a good deal about it (the sum of the violations correlates 0.83 with the rounds over the
        // benchmark batch, their number 0.76: tools/activeset_sim.py), and the slices are in LDS: the instances are dealt to the wavefronts in the order of
        // that sum, back and forth over the four SIMDs (wavefront w runs on SIMD w mod 4) -- the worst SIMD's rounds 25 -> 22 in the simulation (19 with the
        // rounds known beforehand, 18.2 = a quarter of the worst workgroup's), the step 0.0460 -> 0.0408 ms on the GPU (dealt by the number: 0.0425).
        int inst = wave;
#if MPCX_GROUP_BALANCE
        if (b0 + kGroupWaves <= Bt.batch) {
            const double *t0m = mine, *gt0m = mine + ZP, *lgm = gt0m + GPD, *ugm = lgm + GPD;
            auto over = [](double v, double lo, double hi) { return fmax(fmax(lo - v, v - hi), 0.0); };
            double sv = 0.0;
#pragma unroll
            for (int c = 0; c < CPZ; ++c) {
                const int e = 128 * c + 2 * lane;
                const double2 l = *reinterpret_cast<const double2 *>(lwuw + e), u = *reinterpret_cast<const double2 *>(lwuw + ZP + e), t = *reinterpret_cast<const double2 *>(t0m + e);
                sv += over(t.x, l.x, u.x) + over(t.y, l.y, u.y);
            }
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                const int r = 128 * c + 2 * lane;
                const double2 l = *reinterpret_cast<const double2 *>(lgm + r), u = *reinterpret_cast<const double2 *>(ugm + r), t = *reinterpret_cast<const double2 *>(gt0m + r);
                sv += over(t.x, l.x, u.x) + over(t.y, l.y, u.y);
            }
            sv = wave_sum(sv);
            double *keys = reinterpret_cast<double *>(bad);      // (the flags were read before the barrier above; sixteen doubles are theirs)
            if (lane == 0) keys[wave] = sv;
            __syncthreads();
            // (the rank of instance `me`: the sixteen comparisons as four per lane, lane (me, part) against keys 4 part .. 4 part + 3, the parts added up by two
            // lane exchanges -- a quarter of the vector instructions of sixteen broadcasts per lane, in a kernel that is bound by them)
            const int me = lane & (kGroupWaves - 1), part = lane >> 4;
            const double ki = keys[me];
            int rank = 0;
#pragma unroll
            for (int q = 0; q < kGroupWaves / 4; ++q) {
                const int jj = (kGroupWaves / 4) * part + q;
                const double kj = keys[jj];
                rank += (kj > ki || (kj == ki && jj < me)) ? 1 : 0;
            }
            rank += __shfl_xor(rank, 16, 64);
            rank += __shfl_xor(rank, 32, 64);
            const int q = rank >> 2, sd = (q & 1) ? 3 - (rank & 3) : (rank & 3);
            inst = (int)__builtin_ctzll(__ballot(lane < kGroupWaves && 4 * q + sd == wave));      // (a permutation: exactly one lane answers)
        }
#endif
        double *const slice_i = slices + (size_t)inst * M.fast_slice;
        const int b = b0 + inst;
        gstamp(6);
#if defined(MPCX_GROUP_CUT) && MPCX_GROUP_CUT == 3
        { if (threadIdx.x < kGroupWavesOf<CPZ> && blockIdx.x * kGroupWavesOf<CPZ> + threadIdx.x < Bt.batch) Bt.done[blockIdx.x * kGroupWavesOf<CPZ> + threadIdx.x] = 2; return; }
#endif
        if (b < Bt.batch) solve_fast<CPZ, CPG, 2>(M, Bt, b, lane, slice_i, lwuw, glw(wsbase) + (size_t)b * M.wsld, nullptr, outs + inst * outld, eqbits);
        __syncthreads();
        // the sixteen instances' results, written by neighbouring lanes: one transaction per array and workgroup instead of sixteen
        {
            const int t = threadIdx.x;
            if (t < kGroupWaves && b0 + t < Bt.batch) {
                const double *o = outs + t * outld;
                const int bb = b0 + t;
                const bool done = o[7] == 2.0;
                glw(Bt.done)[bb] = done ? 2 : 0;
                if (done) {
                    if (Bt.cost) glw(Bt.cost)[bb] = o[0];
                    if (Bt.status) glw(Bt.status)[bb] = (int)o[1];
                    if (Bt.solver_status) glw(Bt.solver_status)[bb] = (int)o[2];
                    if (Bt.is_feasible) glw(Bt.is_feasible)[bb] = (int)o[3];
                    if (Bt.iterations) glw(Bt.iterations)[bb] = 0;
                    if (Bt.polish_rounds) glw(Bt.polish_rounds)[bb] = (int)o[5];
                    if (Bt.active_count) glw(Bt.active_count)[bb] = (int)o[6];
                }
            }
            for (int e = t; e < kGroupWaves * nu; e += blockDim.x) {
                const int ti = e / nu, jj = e - ti * nu;
                if (b0 + ti < Bt.batch && outs[ti * outld + 7] == 2.0) glw(Bt.cmd)[(size_t)(b0 + ti) * nu + jj] = outs[ti * outld + 8 + jj];
            }
        }
    }
}

}  // namespace
size_t lmpc_group_lds_bytes(const LmpcDev &m);
namespace {

template <int CPZ, int CPG>
int launch_fast_variant(const LmpcDev &m, const LmpcDev *m_dev, const LmpcBatchDev &b, double *ws, hipStream_t stream)
{
    size_t ldsf = ((size_t)2 * 128 * CPZ + (size_t)kWavesPerBlock * m.fast_slice + (b.n_models > 0 ? (size_t)kWavesPerBlock * 2 * 128 * CPZ : 0)) * sizeof(double);
    static const size_t dbg_pad = [] { const char *pad = getenv("MPCX_DBG_LDS_PAD"); return pad ? (size_t)atoi(pad) : (size_t)0; }();      // (occupancy experiments; read once)
    ldsf += dbg_pad;
    const size_t lds_max = lmpc_lds_limit();
    if (ldsf > lds_max) return -2;
    auto k2 = lmpc_solve<CPZ, CPG>;
    auto k2h = lmpc_solve_hetero<CPZ, CPG>;
    // the fused forms serve the one-chunk variant only (fused_record: up to 384 rows of the composed map)
    auto k4 = lmpc_solve_fused<1, 1>;
    auto k5 = lmpc_solve_persistent<1, 1>;
    static std::atomic<size_t> configured[64];
    int devid = 0;
    (void)hipGetDevice(&devid);
    devid &= 63;
    if (ldsf > configured[devid].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(k2h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf) != hipSuccess)
            return -3;
        size_t prev = configured[devid].load(std::memory_order_relaxed);
        while (prev < ldsf && !configured[devid].compare_exchange_weak(prev, ldsf, std::memory_order_release)) {}
    }
    int blocks = (b.batch + kWavesPerBlock - 1) / kWavesPerBlock;      // one wavefront per instance: the grid covers the batch
    if (blocks < 1) blocks = 1;
    const bool fused = b.fused != 0 && CPZ == 1 && CPG == 1;
    if constexpr (CPZ <= 2 && CPZ == CPG) {
        if (b.fused >= 3) {
            // assemble + solve in one workgroup of sixteen (two-chunk variant: eight) wavefronts, one per CU
            constexpr int NW = kGroupWavesOf<CPZ>;
            const size_t ldsg = lmpc_group_lds_bytes(m);
            if (ldsg == 0 || ldsg > lds_max) return -2;
            static std::atomic<int> gconf[64];
            if (!gconf[devid].load(std::memory_order_acquire)) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(lmpc_solve_group<CPZ, CPG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max) != hipSuccess) return -3;
                gconf[devid].store(1, std::memory_order_release);
            }
            const int wgs = (b.batch + NW - 1) / NW;
            hipLaunchKernelGGL((lmpc_solve_group<CPZ, CPG>), dim3(wgs), dim3(NW * 64), ldsg, stream, m, b, ws, b.fused - 3);      // (the model struct by value: see the kernel)
            return hipGetLastError() == hipSuccess ? 0 : -3;
        }
    }
    // persistent form: the composed map, the box bounds and kPersistWaves slices must fit one CU's LDS, and the batch must be worth
    // the prologue of every workgroup (the composed map: 87 KB at N = 20)
    const size_t ldsp = ((size_t)m.rowsF * m.kin + 2 * (size_t)128 + kPersistWaves * (size_t)m.fast_slice) * sizeof(double);
    if (fused && b.pcounter && ldsp <= lds_max && b.batch >= 1024) {
        static std::atomic<int> pconf[64];
        if (!pconf[devid].load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max) != hipSuccess) return -3;
            pconf[devid].store(1, std::memory_order_release);
        }
        (void)hipMemsetAsync(b.pcounter, 0, 8 * sizeof(int), stream);
        int wgs = (b.batch + kPersistWaves - 1) / kPersistWaves;
        if (wgs > 256) wgs = 256;
        hipLaunchKernelGGL(k5, dim3(wgs), dim3(kPersistWaves * 64), ldsp, stream, m_dev, b, ws, b.pcounter);
    } else if (fused) hipLaunchKernelGGL(k4, dim3(blocks), dim3(kWavesPerBlock * 64), ldsf, stream, m_dev, b, ws);
    else if (b.n_models > 0) hipLaunchKernelGGL(k2h, dim3(blocks), dim3(kWavesPerBlock * 64), ldsf, stream, m_dev, b, ws);
    else hipLaunchKernelGGL(k2, dim3(blocks), dim3(kWavesPerBlock * 64), ldsf, stream, m_dev, b, ws);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

int lmpc_fast_slice(const LmpcDev &m)
{
    const int cp = lmpc_kernel_variant(m.ldz, m.ldg);
    if (cp < 1) return 0;
    const int fixed = cp == 1 ? fast_slice_fixed<1, 1>() : (cp == 2 ? fast_slice_fixed<2, 2>() : fast_slice_fixed<4, 4>());
    const int scratch = m.kin > 2 * m.nx ? m.kin : 2 * m.nx;    // vin of the fused record / ping-pong state of the sequence roll-out
    int n = (fixed + scratch + 1) / 2 * 2;
    while (n % 16 != 2) n += 2;                                 // slices 2 doubles apart modulo the 32 LDS banks x 4 bytes
    return n;
}

// LDS block of lmpc_solve_group for this controller (0: no group form for its variant): two box-bound vectors, a slice per instance, the staged
// MFMA operands, the cost constants and flags, the result records, the equality flags of the general rows
size_t lmpc_group_lds_bytes(const LmpcDev &m)
{
    const int cp = lmpc_kernel_variant(m.ldz, m.ldg);
    if (cp != 1 && cp != 2) return 0;
    const int nw = cp == 1 ? 16 : 8;
    return ((size_t)2 * 128 * cp + (size_t)nw * m.fast_slice + (size_t)(m.kin / 4 + m.nz16 / 4) * 64 + (size_t)nw * 16 + 16 + (size_t)nw * (8 + ((m.nu + 1) & ~1)) + 32) * sizeof(double);
}

int lmpc_launch_fast(const LmpcDev &m, const LmpcDev *m_dev, const LmpcBatchDev &b, double *ws, void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (lmpc_kernel_variant(m.ldz, m.ldg)) {
    case 1: return launch_fast_variant<1, 1>(m, m_dev, b, ws, s);
#ifndef MPCX_FAST_ONLY_CP1
    case 2: return launch_fast_variant<2, 2>(m, m_dev, b, ws, s);
    case 4: return launch_fast_variant<4, 4>(m, m_dev, b, ws, s);
#endif
    default: return -2;
    }
}

}  // namespace mpcx
