// nlmpc_sqp_wg: the SQP of nlmpc_engine.hpp as ONE WORKGROUP PER INSTANCE with the reduced problem resident in LDS.
//
// Replaces, like nlmpc_sqp, NLOptimizer::run (reference include/mpc/NLMPC/NLOptimizer.hpp:412-638) with its four NLopt callbacks
// (:760-997 -> Objective.hpp:91-265, Constraints.hpp:211-316, 490-905, Mapping.hpp:174-211) for the built-in systems of
// mpcx/nlmpc_models.hpp (component-wise functors with declared structure).  Same algorithm -- condensing of the dynamics equalities,
// damped inverse BFGS, Goldfarb-Idnani dual active set in range-space form, l1 merit line search -- organised for the CU instead of for
// one wavefront:
//
//   * 1, 2 or 4 wavefronts work on one instance (WAVES; the launcher picks by problem size), every loop is spread over all their lanes;
//   * everything an iteration reads more than once lives in the workgroup's LDS for the whole solve: the iterate, the inverse BFGS matrix
//     (symmetric, packed), the reduced constraint rows, the working set's Cholesky factor, the sparse rows' first entries, the small
//     vectors; the folded dynamics blocks too where they fit (otherwise they stream through the workspace, written and read in
//     whole coalesced blocks, one step ahead of the chain that consumes them).  The dense user Jacobian, the sensitivities Phi, B^-1 N'
//     and the Schur complement of nlmpc_sqp's workspace do not exist any more:
//       - the user Jacobian is kept as its declared non-zero blocks (one NX-vector per (row, state row it reads) pair);
//       - Phi is never stored: the forward sweep that would form it (one column per lane, in registers) delivers the reduced gradient,
//         the reduced constraint rows and their offsets as it goes; the state step is one more sweep with the input step applied;
//       - the dynamics blocks are folded at evaluation time, [A_i | B_i | c_i] <- -E_i^-1 [A_i | B_i | c_i] (one Gauss-Jordan per step on
//         [E | A B c | I], a column per lane), so that every sweep step is one matrix-vector product;
//       - a row leaves the working set by a Givens down-date of the factor, so the Schur complement itself is not kept;
//   * the phases (evaluate / condense / BFGS / sub-problem / step / merit / line search / update) are separate non-inlined functions
//     whose only shared state is the LDS block: each gets its own register allocation and the loop around them carries a dozen scalars.
//
// Vector-valued user hooks (mpcx/nlmpc_hooks.hpp) keep going through nlmpc_sqp.
#pragma once

#include "nlmpc_engine.hpp"

namespace mpcx {
namespace engine {

// ---- the plan: LDS and workspace layout of one instance (host-computed, a kernel argument) ------------------------------------------
struct WgPlan {
    int waves;                  // wavefronts per instance: 1, 2, 4 or (wide systems alone on their CU) 8
    int per_cu;                 // workgroups of this plan a CU's LDS holds
    int hard, nq;               // sub-problem variables: ch nu (+ slack when soft)
    int kw;                     // working-set capacity
    int nd, ndld, nd_user, nsb; // dense sub-problem rows (columns of art): user rows that read a state (or promise no sparsity), then bounds on states
    int nsx;                    // (user row, state row) pairs with a non-zero Jacobian block
    int art_total;              // doubles of the dense rows' storage (every row as long as the last input block it can depend on: wg_row_len)
    int needs_phi;              // some sub-problem row reads a state (otherwise the tables that say which -- xmask, jxoff -- and the LDS copy of the multipliers, which only
                                // the multipliers of the dynamics need, do not exist: 7 KB at eight oscillators with input bounds, what lets that shape fit a CU's LDS)
    int f_lds;                  // the folded dynamics blocks live in LDS
    int minv;                   // the working set's Schur complement is kept as its inverse (large working sets), not as a Cholesky factor --
                                // per instance and per attempt: an instance that fails in the inverse form is solved again, from the start, in the factor form (ST_MINV)
    int cut;                    // the working set's capacity was cut below what the problem can need: a second launch (only_overflowed) takes the instances that outgrow it
    int only_overflowed;        // a second pass with the full working-set capacity: only the instances whose working set outgrew the first pass's
    int curv0;                  // the curvature estimate is set to the condensed Gauss-Newton Hessian of the cost (init_curvature) ...
    int inv_nb, inv_lds;        // init_curvature inverts its matrix in blocks of inv_nb pivots (16 from 97 variables on, 8 below; 0: pivot by pivot -- a debug choice), the panel
                                // buffers in the overlay behind Xs / Us where it has the room (inv_lds), in the workspace otherwise (w_phi)
    int curv_lds;               // init_curvature's two NX x nzu buffers fit the overlay behind Xs / Us (otherwise they are in the workspace: w_phi)
    int curv0_it;               // ... before iteration curv0_it (0: the solve starts from it; k > 0: the first k iterations run from the identity, as NLopt's SLSQP does)
    int carry_m;                // (minv, every row a short list with constant entries) the inverse is carried from one sub-problem to the next: w_msave
    int lds_total;              // doubles
    // LDS offsets (doubles)
    int o_red, o_st, o_z, o_c, o_gin, o_gu, o_gr, o_p, o_glold, o_sv, o_hinv, o_mu, o_flag, o_br, o_s1v, o_s1m, o_dcol, o_xmask, o_jxoff,
        o_slot, o_sbf, o_jx, o_art, o_wq, o_sgq, o_uq, o_tq, o_invd, o_xq, o_np, o_vv, o_wv, o_F, o_prm, o_cd, o_yd,
        o_bidx, o_bsign, o_bval, o_xrf, o_xre, o_drow, o_mbuf, o_aoff, o_alen;
    int o_Xs, o_Us, o_dXs, o_dUs, o_Jm, o_lam, o_dx;      // overlay, outside the sub-problem
    int o_L;                                              // overlay, inside the sub-problem: the packed factor
    // workspace offsets (doubles) of one instance
    int w_scal, w_F, w_art, w_einv, w_gx, w_hinv, w_sp, w_msave, w_phi;   // (w_scal: the controller's NlmpcWsLayout::scal, 16 doubles: cost, dual steps, cycles per phase)
    int ws_total;
};

// The kernel's one argument.  The phases read what they need from it where it lies -- the kernarg segment, constant address space:
// scalar loads, every dimension, offset and base pointer in SGPRs -- instead of carrying it through their calls.
struct WgArgs {
    NlmpcDev M;
    NlmpcSolveDev S;
    WgPlan P;
};
typedef const WgArgs __attribute__((address_space(4))) *WgArgsPtr;
constexpr int kWgCtxDoubles = 0;

// slots of the scalar block st[] through which the phases hand results to the loop
enum { ST_COST = 0, ST_FP, ST_FM, ST_ERR, ST_LAMDYN, ST_R0, ST_R1, ST_R2, ST_R3, ST_R4, ST_R5, ST_SHED = 12 /* two 64-bit words */, ST_NSHED = 14 /* rows shed at warm starts, whole solve */, ST_OVER = 15 /* the working set outgrew its capacity */, ST_ACC = 16,
       ST_CARRY = 24 /* the saved inverse: 0 none, 1 valid for the current B^-1, 2 valid up to the BFGS update in ST_BFRHO / ST_BFCC */, ST_BFRHO, ST_BFCC, ST_CARRYN /* sub-problems since the inverse was formed afresh */, ST_NCARRY /* warm starts that took the carried inverse, whole solve */, ST_CONSET /* eval_con has filed the rows once */,
       ST_MINV = 30 /* the sub-problems keep the Schur complement's inverse (WgPlan::minv, the first attempt, until one of them fails its check) */,
       ST_QNW /* rows in the working set when the last sub-problem ended, failed or not */, ST_SWITCHED /* 1: this attempt has left the inverse form */,
       ST_QSTAT = 40,
#ifdef MPCX_NL_STATS
       ST_TOTAL = 56
#else
       ST_TOTAL = 40
#endif
};
// -DMPCX_NL_STATS (libmpcx_stats.so): shader-clock cycles of the sub-problem's parts in st[ST_QSTAT ..]: unconstrained minimiser, warm start (the kept
// rows' Schur complement, its factor, the shedding rounds), and per dual step: scan, entering row, N_W v, solve, N_W' r, B^-1 w, the rest
#ifdef MPCX_NL_STATS
#define MPCX_QLAP(k) do { const long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) st[ST_QSTAT + (k)] += (double)(now_ - qt_); qt_ = now_; } while (0)
#else
#define MPCX_QLAP(k) do { } while (0)
#endif

#define MPCX_WG_PHASE __device__ __attribute__((noinline))
// a probe point of the host interpreter's tests (tests/emu/wg_probes.hpp defines it before this header is read); nothing in the product
#ifndef MPCX_WG_PROBE_CARRY
#define MPCX_WG_PROBE_CARRY(...) do { } while (0)
#endif
// -DMPCX_EMU_TRACE (the host interpreter of tests/emu only): a line per sub-problem on stderr
#ifdef MPCX_EMU_TRACE
#define MPCX_TRACE(...) do { if (threadIdx.x == 0 && blockIdx.x == 0) fprintf(stderr, __VA_ARGS__); } while (0)
#else
#define MPCX_TRACE(...) do { } while (0)
#endif

__device__ __forceinline__ double *wg_lds()
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    return smem;
}
// (Inside a function that is not the kernel the kernarg segment pointer itself is not an input; the pointer to the implicit arguments,
// which follow the explicit ones at the next multiple of eight bytes, is.)
__device__ __forceinline__ WgArgsPtr wg_args()
{
    typedef const char __attribute__((address_space(4))) *cptr;
    return (WgArgsPtr)((cptr)__builtin_amdgcn_implicitarg_ptr() - ((sizeof(WgArgs) + 7) & ~(size_t)7));
}

typedef double wg_v4d __attribute__((ext_vector_type(4)));      // the accumulator of v_mfma_f64_16x16x4_f64: element (lane >> 4) + 4 r, column lane & 15 in register r

template <int CTRL> __device__ __forceinline__ double row_share(double v) { return dpp_d<CTRL>(v); }      // 0x150 + n: lane n of every row of 16

template <int P> __device__ __forceinline__ double group_sum(double v)      // over P adjacent lanes (P = 1, 2, 4, 8, 16), every lane gets the sum
{
    if constexpr (P >= 2) v += dpp_d<0xB1>(v);
    if constexpr (P >= 4) v += dpp_d<0x4E>(v);
    if constexpr (P >= 8) v += dpp_d<0x141>(v);
    if constexpr (P >= 16) v += dpp_d<0x140>(v);
    return v;
}

template <int WAVES> struct Team {
    static constexpr int NT = 64 * WAVES;
    static __device__ __forceinline__ void sync()
    {
        if constexpr (WAVES == 1) nl_wave_sync(); else __syncthreads();
    }
};

// Reductions over the workgroup: a wave-level reduction, one LDS slot per wavefront, one barrier; the result is the same bits in every
// thread (fixed order).  Two sets of slots alternate, so that consecutive reductions need no second barrier; a phase ends with a barrier.
// The bodies are out of line: a solve passes through some forty reductions per iteration, and what bounds a wavefront that walks a long
// instruction stream once per iteration is the instruction cache (64 KB for two CUs) -- every helper below exists once per kernel.
#define MPCX_WG_CALL __device__ __attribute__((noinline))
template <int WAVES> MPCX_WG_CALL double wg_red_sum(double v, double *s)
{
    v = wave_sum(v);
    if constexpr (WAVES == 1) return v;
    else {
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
        __syncthreads();
        double r = s[0];
#pragma unroll
        for (int i = 1; i < WAVES; ++i) r += s[i];
        return r;
    }
}
template <int WAVES> MPCX_WG_CALL double wg_red_max(double v, double *s)
{
    v = wave_max(v);
    if constexpr (WAVES == 1) return v;
    else {
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
        __syncthreads();
        double r = s[0];
#pragma unroll
        for (int i = 1; i < WAVES; ++i) r = fmax(r, s[i]);
        return r;
    }
}
// two sums with one barrier
struct WgSum2 { double a, b; };
template <int WAVES> MPCX_WG_CALL WgSum2 wg_red_sum2(double a, double b, double *s)
{
    a = wave_sum(a); b = wave_sum(b);
    if constexpr (WAVES > 1) {
        if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = a; s[8 + (threadIdx.x >> 6)] = b; }
        __syncthreads();
        a = s[0]; b = s[8];
#pragma unroll
        for (int i = 1; i < WAVES; ++i) { a += s[i]; b += s[8 + i]; }
    }
    return WgSum2{a, b};
}
// four reductions with one barrier; bit i of maxmask: value i is a maximum, otherwise a sum.  Takes both parities of a slot set: the
// caller puts a barrier before its next reduction.
struct WgRed4 { double a, b, c, d; };
template <int WAVES> MPCX_WG_CALL WgRed4 wg_red_mix4(double a, double b, double c, double d, int maxmask, double *s)
{
    a = (maxmask & 1) ? wave_max(a) : wave_sum(a); b = (maxmask & 2) ? wave_max(b) : wave_sum(b);
    c = (maxmask & 4) ? wave_max(c) : wave_sum(c); d = (maxmask & 8) ? wave_max(d) : wave_sum(d);
    if constexpr (WAVES > 1) {
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { s[w] = a; s[8 + w] = b; s[16 + w] = c; s[24 + w] = d; }
        __syncthreads();
        a = s[0]; b = s[8]; c = s[16]; d = s[24];
#pragma unroll
        for (int i = 1; i < WAVES; ++i) {
            a = (maxmask & 1) ? fmax(a, s[i]) : a + s[i]; b = (maxmask & 2) ? fmax(b, s[8 + i]) : b + s[8 + i];
            c = (maxmask & 4) ? fmax(c, s[16 + i]) : c + s[16 + i]; d = (maxmask & 8) ? fmax(d, s[24 + i]) : d + s[24 + i];
        }
    }
    return WgRed4{a, b, c, d};
}
// largest value and the lowest index holding it
struct WgArgmax { double v; int idx; };
template <int WAVES> MPCX_WG_CALL WgArgmax wg_red_argmax(double v, int idx, double *s)
{
    wave_argmax(v, idx);
    if constexpr (WAVES > 1) {
        if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = v; s[8 + (threadIdx.x >> 6)] = (double)idx; }
        __syncthreads();
        v = s[0]; idx = (int)s[8];
#pragma unroll
        for (int i = 1; i < WAVES; ++i) {
            const double ov = s[i]; const int oi = (int)s[8 + i];
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
    }
    return WgArgmax{v, idx};
}
// Slot sets: set 0 is everybody's (a phase that reduces ends with a barrier, so the next phase may start over at its first parity); the scan and
// the entering row of the dual method, which follow each other with their results in registers, have a set each and no barrier behind them --
// a set is written again only after every thread has passed a later barrier of the step.
constexpr int kWgRedDoubles = 64;     // set 0: two parities of sixteen; sets 1 and 2: one reduction per use, sixteen each
template <int WAVES> struct Red {
    double *buf;
    int par;
    __device__ __forceinline__ explicit Red(double *b, int set = 0) : buf(b + (set == 0 ? 0 : 16 + 16 * set)), par(0) {}
    __device__ __forceinline__ double *slots() { double *s = buf + par * 16; par ^= 1; return s; }
    __device__ __forceinline__ double sum(double v) { return wg_red_sum<WAVES>(v, slots()); }
    __device__ __forceinline__ double max(double v) { return wg_red_max<WAVES>(v, slots()); }
    __device__ __forceinline__ void argmax(double &v, int &idx) { const WgArgmax r = wg_red_argmax<WAVES>(v, idx, slots()); v = r.v; idx = r.idx; }
    __device__ __forceinline__ WgRed4 mix4(double a, double b, double c, double d, int maxmask) { par = 0; return wg_red_mix4<WAVES>(a, b, c, d, maxmask, buf); }
    __device__ __forceinline__ void sum2(double &a, double &b) { const WgSum2 r = wg_red_sum2<WAVES>(a, b, slots()); a = r.a; b = r.b; }
};

// dst = scale * H src for the symmetric matrix H packed by rows of its lower triangle (row r at r (r + 1) / 2), all in LDS;
// P lanes share a row: the row's own part (entries left of the diagonal, contiguous) and the column part below it (entry (c, r) at
// c (c + 1) / 2 + r: the offset grows by c + 1 per step), four independent partial sums each.  Every thread of the workgroup calls it;
// the caller synchronises.
template <int P, int NT, bool STEP = false>
__device__ __forceinline__ void hmul_rows(const double *hp, const double *src, double *dst, int n, double scale, int tid, const double *extra = nullptr)
{
    const int part = tid & (P - 1);
    for (int r0 = 0; r0 < n; r0 += NT / P) {
        const int r = r0 + tid / P;
        const bool live = r < n;
        const int rr = live ? r : 0;
        const double *row = hp + rr * (rr + 1) / 2;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int c = part;
        for (; c + 3 * P <= rr; c += 4 * P) {
            a0 = fma(row[c], src[c], a0); a1 = fma(row[c + P], src[c + P], a1);
            a2 = fma(row[c + 2 * P], src[c + 2 * P], a2); a3 = fma(row[c + 3 * P], src[c + 3 * P], a3);
        }
        for (; c <= rr; c += P) a0 = fma(row[c], src[c], a0);
        // c is now the first column beyond the diagonal that this lane takes
        int off = c * (c + 1) / 2 + rr;
        for (; c + 3 * P < n; c += 4 * P) {
            const int o1 = off + P * c + P * (P + 1) / 2, o2 = o1 + P * (c + P) + P * (P + 1) / 2, o3 = o2 + P * (c + 2 * P) + P * (P + 1) / 2;
            a0 = fma(hp[off], src[c], a0); a1 = fma(hp[o1], src[c + P], a1);
            a2 = fma(hp[o2], src[c + 2 * P], a2); a3 = fma(hp[o3], src[c + 3 * P], a3);
            off = o3 + P * (c + 3 * P) + P * (P + 1) / 2;
        }
        for (; c < n; c += P) { a1 = fma(hp[off], src[c], a1); off += P * c + P * (P + 1) / 2; }
        const double acc = group_sum<P>((a0 + a1) + (a2 + a3));
        if (live && part == 0) {
            if constexpr (STEP) dst[r] -= scale * ((extra ? extra[r] : 0.0) - acc);      // the dual step's move: x -= t (v - B^-1 w)
            else dst[r] = scale * acc;
        }
    }
}
template <int NT, bool STEP = false>
__device__ __forceinline__ void hmul(const double *hp, const double *src, double *dst, int n, double scale, int tid, const double *extra = nullptr)
{
    if (4 * n <= NT) hmul_rows<4, NT, STEP>(hp, src, dst, n, scale, tid, extra);
    else if (2 * n <= NT) hmul_rows<2, NT, STEP>(hp, src, dst, n, scale, tid, extra);
    else hmul_rows<1, NT, STEP>(hp, src, dst, n, scale, tid, extra);
}
__device__ __forceinline__ double hsym(const double *hp, int r, int c) { return r >= c ? hp[r * (r + 1) / 2 + c] : hp[c * (c + 1) / 2 + r]; }

// (row, column) of element e of a packed lower triangle
__device__ __forceinline__ void tri_index(int e, int &r, int &c)
{
    r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while (r * (r + 1) / 2 > e) --r;
    while ((r + 1) * (r + 2) / 2 <= e) ++r;
    c = e - r * (r + 1) / 2;
}

// ---- the working set's factor, one wavefront (rows lane and lane + 64) ------------------------------------------------------------------
// S = L L' in place: Lp holds the lower triangle of S, packed by rows; right-looking, as chol_factor's LDS part
__device__ __forceinline__ bool chol_inplace(double *Lp, double *invd, int n, int lane)
{
    double dmax = 0.0;
    for (int r = lane; r < n; r += 64) dmax = fmax(dmax, Lp[r * (r + 1) / 2 + r]);
    dmax = wave_max(dmax);
    bool ok = true;
    const int o0 = lane * (lane + 1) / 2, o1 = (lane + 64) * (lane + 65) / 2;
    for (int k = 0; k < n; ++k) {
        const double dkk = Lp[k * (k + 1) / 2 + k];
        ok &= dkk > 1e-13 * dmax;
        const double dd = sqrt(dkk > 1e-13 * dmax ? dkk : 1e-13 * dmax + 1e-300), id = 1.0 / dd;
        double l0 = 0.0, l1 = 0.0;
        if (lane > k && lane < n) { l0 = Lp[o0 + k] * id; }
        if (lane + 64 > k && lane + 64 < n) { l1 = Lp[o1 + k] * id; }
        nl_wave_sync();                                           // everybody has read the pivot before it is overwritten
        if (lane > k && lane < n) Lp[o0 + k] = l0;
        if (lane + 64 > k && lane + 64 < n) Lp[o1 + k] = l1;
        if (lane == 0) { Lp[k * (k + 1) / 2 + k] = dd; invd[k] = id; }
        nl_wave_sync();
        for (int j = k + 1; j < n; ++j) {
            const double ljk = Lp[j * (j + 1) / 2 + k];
            if (lane >= j && lane < n) Lp[o0 + j] = fma(-l0, ljk, Lp[o0 + j]);
            if (lane + 64 >= j && lane + 64 < n) Lp[o1 + j] = fma(-l1, ljk, Lp[o1 + j]);
        }
        nl_wave_sync();
    }
    return ok;
}
// L y = t and L' x = y on the leading n rows of the packed factor (rows lane and, TWO, lane + 64), column by column: once an unknown is
// known every remaining row loses its term; an element is scaled by its reciprocal diagonal when it is handed on and, for the result,
// once at the end -- no per-step predicates on "my own element", the entries beyond a row's end are masked after the load.
template <bool TWO>
__device__ __forceinline__ void tri_forward(const double *Lp, const double *invd, int n, double &t0, double &t1, int lane)
{
    const int o0 = lane * (lane + 1) / 2, o1 = (lane + 64) * (lane + 65) / 2;
    const int h0 = lane < n ? lane : -1, h1 = (TWO && lane + 64 < n) ? lane + 64 : -1;      // a row takes part in step k while k < its number
    double a0 = 0 < h0 ? Lp[o0] : 0.0, a1 = TWO ? (0 < h1 ? Lp[o1] : 0.0) : 0.0, id = invd[0];
    const int n0 = TWO ? min(n, 64) : n;
    for (int k = 0; k < n0; ++k) {
        const int kn = k + 1;
        const double a0n = Lp[o0 + kn], idn = invd[min(kn, n - 1)];
        double a1n = 0.0;
        if constexpr (TWO) a1n = Lp[o1 + kn];
        const double yk = read_lane(t0, k) * id;
        t0 = fma(-a0, yk, t0);
        if constexpr (TWO) t1 = fma(-a1, yk, t1);
        a0 = kn < h0 ? a0n : 0.0; id = idn;
        if constexpr (TWO) a1 = kn < h1 ? a1n : 0.0;
    }
    if constexpr (TWO) {
        for (int k = 64; k < n; ++k) {
            const int kn = k + 1;
            const double a1n = Lp[o1 + kn], idn = invd[min(kn, n - 1)];
            const double yk = read_lane(t1, k - 64) * id;
            t1 = fma(-a1, yk, t1);
            a1 = kn < h1 ? a1n : 0.0; id = idn;
        }
    }
    t0 *= lane < n ? invd[lane] : 0.0;
    if constexpr (TWO) t1 *= lane + 64 < n ? invd[lane + 64] : 0.0;
}
template <bool TWO>
__device__ __forceinline__ void tri_backward(const double *Lp, const double *invd, int n, double &t0, double &t1, int lane)
{
    // column k of L' is row k of L: entries 0 .. k-1 contiguous at k (k + 1) / 2
    int k = n - 1;
    double id = invd[k];
    double a0 = lane < k ? Lp[k * (k + 1) / 2 + lane] : 0.0, a1 = (TWO && lane + 64 < k) ? Lp[k * (k + 1) / 2 + lane + 64] : 0.0;
    if constexpr (TWO) {
        for (; k >= 64; --k) {
            const int kn = k - 1, on = kn * (kn + 1) / 2;
            const double a0n = Lp[on + lane], a1n = Lp[on + lane + 64], idn = invd[kn];
            const double xk = read_lane(t1, k - 64) * id;
            t0 = fma(-a0, xk, t0); t1 = fma(-a1, xk, t1);
            a0 = lane < kn ? a0n : 0.0; a1 = lane + 64 < kn ? a1n : 0.0; id = idn;
        }
    }
    for (; k >= 0; --k) {
        const int kn = max(k - 1, 0), on = kn * (kn + 1) / 2;
        const double a0n = Lp[on + lane], idn = invd[kn];
        const double xk = read_lane(t0, k) * id;
        t0 = fma(-a0, xk, t0);
        a0 = lane < kn ? a0n : 0.0; id = idn;
    }
    t0 *= lane < n ? invd[lane] : 0.0;
    if constexpr (TWO) t1 *= lane + 64 < n ? invd[lane + 64] : 0.0;
}
// S = L L' in place by the whole workgroup: per column the pivot, the column's scaling (one row per thread), and the trailing update
// spread over the threads in two dimensions (four threads share a row) -- the one-wavefront form walks each row's update serially.
template <int WAVES>
__device__ __forceinline__ bool chol_inplace_wg(double *Lp, double *invd, double *flagslot, int n, int tid)
{
    constexpr int NT = 64 * WAVES;
    using T = Team<WAVES>;
    double dmax = 0.0;
    for (int r = tid; r < n; r += NT) dmax = fmax(dmax, Lp[r * (r + 1) / 2 + r]);
    dmax = wave_max(dmax);
    if constexpr (WAVES > 1) {
        if ((tid & 63) == 0) flagslot[1 + (tid >> 6)] = dmax;
        T::sync();
        dmax = flagslot[1];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) dmax = fmax(dmax, flagslot[1 + w]);
    }
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        T::sync();                                               // the previous column's trailing update is complete
        const double dkk = Lp[k * (k + 1) / 2 + k];
        ok &= dkk > 1e-13 * dmax;
        const double dd = sqrt(dkk > 1e-13 * dmax ? dkk : 1e-13 * dmax + 1e-300), id = 1.0 / dd;
        T::sync();                                               // everybody has read the pivot
        for (int r = k + 1 + tid; r < n; r += NT) Lp[r * (r + 1) / 2 + k] *= id;
        if (tid == 0) { Lp[k * (k + 1) / 2 + k] = dd; invd[k] = id; }
        T::sync();
        const int part = tid & 3;
        for (int r = k + 1 + (tid >> 2); r < n; r += NT / 4) {
            const int ro = r * (r + 1) / 2;
            const double lrk = Lp[ro + k];
            for (int j = k + 1 + part; j <= r; j += 4) Lp[ro + j] = fma(-lrk, Lp[j * (j + 1) / 2 + k], Lp[ro + j]);
        }
    }
    T::sync();
    return ok;
}
// Row and column j leave S: row j of L is deleted, the rows below move up, and Givens rotations on the column pairs (r, r + 1),
// r = j .. n - 2, restore the triangle.  Every lane streams along its own rows: it carries the rotated entry of column r + 1 into the
// next rotation; the rotation itself comes from the row whose diagonal it creates.
__device__ __forceinline__ void chol_delete(double *Lp, double *invd, int n, int j, int lane)
{
    const int r0 = lane, r1 = lane + 64;
    const bool m0 = r0 > j && r0 < n, m1 = r1 > j && r1 < n;
    const int s0 = r0 * (r0 + 1) / 2, s1 = r1 * (r1 + 1) / 2;                  // where the rows are
    const int d0 = m0 ? (r0 - 1) * r0 / 2 : 0, d1 = m1 ? (r1 - 1) * r1 / 2 : 0; // where they go
    double carry0 = m0 ? Lp[s0 + j] : 0.0, carry1 = m1 ? Lp[s1 + j] : 0.0;
    // columns 0 .. j-1 move up unchanged, eight at a time: everybody reads before anybody writes
    for (int c0 = 0; c0 < j; c0 += 8) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = min(c0 + u, j - 1);
            a[u] = m0 ? Lp[s0 + c] : 0.0; b[u] = m1 ? Lp[s1 + c] : 0.0;
        }
        nl_wave_sync();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c0 + u < j) { if (m0) Lp[d0 + c0 + u] = a[u]; if (m1) Lp[d1 + c0 + u] = b[u]; }
        }
        nl_wave_sync();
    }
    // (The entry of the next rotation is requested before this one's results are stored -- a row reads column r + 2 two rotations before
    // its neighbour overwrites it, the wavefront runs in lockstep -- and 1 / sqrt comes from v_rsq_f64 and two Newton steps instead of the
    // square root's and the division's expansions: the rotations are a serial chain, their latency is what a row that leaves costs.)
    double y0n = (r0 >= j + 1 && r0 < n) ? Lp[s0 + j + 1] : 0.0, y1n = (r1 >= j + 1 && r1 < n) ? Lp[s1 + j + 1] : 0.0;
    for (int r = j; r <= n - 2; ++r) {
        const bool u0 = r0 >= r + 1 && r0 < n, u1 = r1 >= r + 1 && r1 < n;
        const double y0 = y0n, y1 = y1n;
        y0n = (r0 >= r + 2 && r0 < n) ? Lp[s0 + r + 2] : 0.0; y1n = (r1 >= r + 2 && r1 < n) ? Lp[s1 + r + 2] : 0.0;
        const int pl = (r + 1) & 63;
        const double a = (r + 1) < 64 ? read_lane(carry0, pl) : read_lane(carry1, pl);
        const double b = (r + 1) < 64 ? read_lane(y0, pl) : read_lane(y1, pl);
        const double h2 = fma(a, a, b * b);
        double ir = __builtin_amdgcn_rsq(h2);
        ir = ir * fma(-0.5 * h2 * ir, ir, 1.5);
        ir = ir * fma(-0.5 * h2 * ir, ir, 1.5);
        if (!(h2 > 0.0)) ir = 0.0;                                 // (a dependent row: the factor's pivot test has the word)
        const double c = a * ir, s = b * ir;
        const double n0 = c * carry0 + s * y0, n1 = c * carry1 + s * y1;
        carry0 = c * y0 - s * carry0; carry1 = c * y1 - s * carry1;
        if (u0) Lp[d0 + r] = n0;
        if (u1) Lp[d1 + r] = n1;
        if (lane == 0) invd[r] = ir;
    }
    nl_wave_sync();
}

// The dense rows of the sub-problem are stored row by row, each as long as it can be non-zero: a row that reads states up to x_{s+1} depends on
// the input blocks 0 .. min(s, ch - 1) only -- on average half of the nq columns at config 3 (art: 15 KB instead of 30).  Returns the length of
// user row k in bits 0 .. 15 and bit 30 if nothing but the sweep writes it (no input read directly, no slack entry: the sweep then stores
// its entries instead of adding them to zeros).  Wide states (nx > 8: the forward sweep) keep whole rows.
template <class Mdl> __host__ __device__ inline int wg_row_len(int k, int ph, int ch, int nq, int nzu, int mi)
{
    constexpr int NU = Mdl::NU;
    if (Mdl::NX > 8) return nq;
    int top = -1, ublk = -1;
    for (int i = 1; i <= ph; ++i) if (k < mi ? Mdl::ineq_reads_x(k, i) : Mdl::eq_reads_x(k - mi, i)) top = i - 1;
    for (int i = 0; i < ph; ++i) if (k < mi ? Mdl::ineq_reads_u(k, i) : Mdl::eq_reads_u(k - mi, i)) ublk = i < ch - 1 ? i : ch - 1;
    if (Mdl::INEQ_USES_SLACK && k < mi && nq > nzu) return nq;
    const int lx = top >= 0 ? ((top < ch - 1 ? top : ch - 1) + 1) * NU : 0, lu = (ublk + 1) * NU;
    const int len = lx > lu ? lx : lu;
    return len | (ublk < 0 ? 1 << 30 : 0);
}
// (a bound on a state: the whole input part -- the host's plan does not see which state it bounds; zeroed at evaluation, the sweep adds)
__host__ __device__ inline int wg_bound_row_len(int nzu) { return nzu; }

// Register budget (inherited by every phase): two wavefronts per SIMD -- no phase of any built-in system spills at 256 registers, and the LDS
// block of a problem that takes several wavefronts leaves room for two or three workgroups per CU at most; four per SIMD (128 registers)
// for the smallest systems at one wavefront per instance, whose 4 KB blocks let sixteen instances share a CU, and for the four-wavefront
// variant of a small-state system that keeps its blocks and reduced rows in the workspace: its LDS block is a quarter of a CU's (config 3:
// 40 KB), and four workgroups per CU deliver more than three at 168 registers did (63 k against 56 k solves/s).
template <class Mdl, int WAVES, bool FL> struct kWgWavesPerSimdOf {
    static constexpr int value = (WAVES == 1 && Mdl::NX <= 2) ? 4 : (WAVES == 4 && !FL && Mdl::NX <= 4) ? 4 : 2;
};

// ---- the kernel's phases ----------------------------------------------------------------------------------------------------------------
template <class Mdl, int WAVES, bool FL>
struct WgSqp {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, FW = NX + NU + 1, NT = 64 * WAVES;
    static constexpr bool CT = Mdl::CONTINUOUS;
    static constexpr int GW = 3 * NX + NU + 1;                 // columns of the Gauss-Jordan tableau [E | A B c | I] of one step
    using T = Team<WAVES>;

    struct V {                                                 // what a phase needs of the arguments and of the LDS block (scalars throughout)
        WgArgsPtr A;
        double *sm;
        int b;                                                 // this workgroup's instance
        double *w;                                             // its workspace
        int ph, ch, nz, nxs, nzu, nr, mi, m, mt, nq, ndld, kw;
        __device__ __forceinline__ V()
            : A(wg_args()), sm(wg_lds()), b(blockIdx.x), w(A->S.ws + (size_t)b * A->M.ws.total), ph(A->M.ph), ch(A->M.ch), nz(A->M.nz),
              nxs(ph * NX), nzu(A->M.nzu), nr(nzu + 1), mi(A->M.nineq), m(mi + A->M.nue), mt(m + A->M.nbnd), nq(A->P.nq), ndld(A->P.ndld), kw(A->P.kw) {}
        __device__ __forceinline__ double *at(int off) const { return sm + off; }
        __device__ __forceinline__ int *iat(int off) const { return reinterpret_cast<int *>(sm + off); }
        __device__ __forceinline__ const double *x0() const { return A->S.x0 + (size_t)b * NX; }
        __device__ __forceinline__ const double *u0() const { return A->S.u0 + (size_t)b * NU; }
        __device__ __forceinline__ Scale scale() const { return Scale(A->M.su, A->M.ss, A->M.iss, A->M.scaled != 0); }
    };
    // the folded blocks: LDS or workspace, the pointer typed accordingly
    struct FP {
        typedef typename BlockPtr<FL>::type type;
        static __device__ __forceinline__ type get(const V &v) { return BlockPtr<FL>::make(FL ? v.sm + v.A->P.o_F : v.w + v.A->P.w_F); }
    };

    // the dense reduced rows art [nq x ndld]: where the folded blocks are -- in LDS (FL), or in the workspace, from where whole-matrix products
    // fetch them in bulk (contiguous, a dozen loads per lane in flight) and where a CU's L2 keeps the few resident instances' copies
    static __device__ __forceinline__ typename FP::type art_of(const V &v) { return BlockPtr<FL>::make(FL ? v.sm + v.A->P.o_art : v.w + v.A->P.w_art); }

    // the sparse form of sub-problem row k (as in nlmpc_sqp): first entry and count in LDS, entries 1 .. 3 in the workspace
    struct Sp {
        const double *s1v; const int *s1m; const double *spv; const int *spi;
        __device__ __forceinline__ explicit Sp(const V &v)
            : s1v(v.at(v.A->P.o_s1v)), s1m(v.iat(v.A->P.o_s1m)), spv(v.w + v.A->P.w_sp),
              spi(reinterpret_cast<const int *>(v.w + v.A->P.w_sp + (size_t)v.mt * kNlSparse)) {}
        __device__ __forceinline__ int count(int k) const { return s1m[k] >> 16; }
        __device__ __forceinline__ int index(int k, int j) const { return j == 0 ? (s1m[k] & 0xffff) : spi[k * kNlSparse + j]; }
        __device__ __forceinline__ double value(int k, int j) const { return j == 0 ? s1v[k] : spv[k * kNlSparse + j]; }
        __device__ __forceinline__ double dot(int k, const double *x) const
        {
            const int mw = s1m[k];
            if ((mw >> 16) == 0) return 0.0;
            double acc = s1v[k] * x[mw & 0xffff];
            for (int j = 1; j < (mw >> 16); ++j) acc = fma(spv[k * kNlSparse + j], x[spi[k * kNlSparse + j]], acc);
            return acc;
        }
    };

    // ------------------------------------------------------------------------------------------------------------------------------
    // start: initial guess (NLOptimizer.hpp:431-510), inverse Hessian estimate, structure tables
    // (attempt 1: the same instance again after the inverse form failed -- everything as at the first, the factor form)
    static MPCX_WG_PHASE void start(int attempt)
    {
        const V v; const auto &M = v.A->M; const auto &S = v.A->S; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ph = v.ph, ch = v.ch, nz = v.nz, nxs = v.nxs, nzu = v.nzu, nr = v.nr, mi = v.mi, m = v.m, mt = v.mt;
        const int nbnd = mt - m, nsb = P.nsb, nd_user = P.nd_user;
        double *z = v.at(P.o_z), *hinv = v.at(P.o_hinv), *mu = v.at(P.o_mu), *st = v.at(P.o_st);
        int *flag = v.iat(P.o_flag), *dcol = v.iat(P.o_dcol), *jxoff = v.iat(P.o_jxoff), *slot = v.iat(P.o_slot), *sbf = v.iat(P.o_sbf);
        unsigned long long *xmask = reinterpret_cast<unsigned long long *>(v.at(P.o_xmask));
        const double *x0 = v.x0(), *u0 = v.u0(), *zwarm = S.z_warm;
        const double *zlb = M.zlb, *zub = M.zub;
        const int *bnd_idx = M.bnd_idx;
        {
            // the model's parameters and the bound table into LDS: every model function reads parameters, and a read from HBM in the
            // middle of a dependent chain costs a memory latency each time
            const double *prm_g = S.params_b ? S.params_b + (size_t)v.b * S.nparams : M.params, *bs = M.bnd_sign, *bv = M.bnd_val;
            double *prm_l = v.at(P.o_prm), *bsl = v.at(P.o_bsign), *bvl = v.at(P.o_bval);
            int *bil = v.iat(P.o_bidx);
            for (int k = tid; k < Mdl::NPARAMS; k += NT) prm_l[k] = prm_g[k];
            for (int k = tid; k < nbnd; k += NT) { bil[k] = bnd_idx[k]; bsl[k] = bs[k]; bvl[k] = bv[k]; }
        }
        if (zwarm) {
            const double *zw = zwarm + (size_t)v.b * nz;
            for (int k = tid; k < nxs; k += NT) { const int i = k / NX; z[k] = zw[i == ph - 1 ? k : k + NX]; }
            for (int k = tid; k < nzu; k += NT) {
                const int bl = k / NU, j = k - bl * NU;
                const int step = min(bl + 1, ph - 1);
                z[nxs + k] = zw[nxs + min(step, ch - 1) * NU + j];
            }
            if (tid == 0) z[nz - 1] = zw[nz - 1];
        } else {
            for (int k = tid; k < nxs; k += NT) z[k] = x0[k % NX];
            for (int k = tid; k < nzu; k += NT) z[nxs + k] = u0[k % NU];
            if (tid == 0) z[nz - 1] = 0.0;
        }
        const int nh = nr * (nr + 1) / 2;
        if (S.keep_curvature) {                          // the estimate the previous tick's solve left in the workspace
            const double *hs = v.w + P.w_hinv;
            for (int e = tid; e < nh; e += NT) hinv[e] = hs[e];
        } else {
            for (int e = tid; e < nh; e += NT) { int r, c; tri_index(e, r, c); hinv[e] = r == c ? 1.0 : 0.0; }
        }
        const bool lean = !P.needs_phi;                          // no row reads a state: no xmask / jxoff / mu in LDS
        for (int k = tid; k < mt; k += NT) { if (!lean) mu[k] = 0.0; flag[k] = 0; }
        for (int k = tid; k < ST_TOTAL; k += NT) st[k] = (k == ST_MINV && P.minv && attempt == 0) ? 1.0 : 0.0;
        T::sync();
        // NLOptimizer::fixOptimalSolution (NLOptimizer.hpp:705-716): a start outside the bounds goes to (ub - lb) / 2 (sic)
        for (int k = tid; k < nz; k += NT) {
            const double lo = zlb[k], hi = zub[k];
            if (z[k] < lo || z[k] > hi) z[k] = (hi - lo) / 2.0;
        }
        // which state rows a user row reads (bit i-1: X row i, i = 1 .. ph; row 0 is x0, not a variable)
        if (!lean) for (int k = tid; k < m; k += NT) {
            unsigned long long mk = 0;
            for (int i = 1; i <= ph; ++i)
                if (k < mi ? Mdl::ineq_reads_x(k, i) : Mdl::eq_reads_x(k - mi, i)) mk |= 1ull << (i - 1);
            xmask[k] = mk;
        }
        T::sync();
        if (tid == 0) {                                       // prefix counts: Jacobian block slots and dense columns
            int ns = 0, ndc = 0;
            int *drow = v.iat(P.o_drow);                        // the user row behind dense column dc
            for (int k = 0; k < m; ++k) {
                if (!lean) jxoff[k] = ns;
                unsigned long long mk = lean ? 0ull : xmask[k];
                const bool dense = mk != 0ull || !Mdl::XFREE_ROWS_SPARSE;
                dcol[k] = dense ? ndc : -1;
                if (dense) drow[ndc++] = k;
                while (mk) { const int i = (int)__builtin_ctzll(mk) + 1; mk &= mk - 1; slot[ns++] = (k << 8) | i; }
            }
            if (!lean) jxoff[m] = ns;
            // the same pairs grouped by state row: what the sweep consumes as it passes state row i + 1 (entry: row << 12 | block slot;
            // bit 31: the pair is all there is to the row's reduced entries -- one state row, no input -- and is stored, not added)
            int *xrf = v.iat(P.o_xrf), *xre = v.iat(P.o_xre);
            int ne = 0;
            for (int i = 1; i <= ph; ++i) {
                xrf[i - 1] = ne;
                if (!lean) for (int k = 0; k < m; ++k)
                    if ((xmask[k] >> (i - 1)) & 1ull)
                        xre[ne++] = (k << 12) | (jxoff[k] + __builtin_popcountll(xmask[k] & ((1ull << (i - 1)) - 1ull))) | (sole_state_row(k, xmask[k], mi, ph) ? (int)0x80000000 : 0);
            }
            xrf[ph] = ne;
        }
        // bounds: those on states come first in the table (ascending index) and are dense rows after the user's
        for (int kb = tid; kb < nbnd; kb += NT) dcol[m + kb] = bnd_idx[kb] < nxs ? nd_user + kb : -1;
        T::sync();
        {
            // where the dense rows lie (wg_row_len): lengths by every thread, offsets by one
            int *aoff = v.iat(P.o_aoff), *alen = v.iat(P.o_alen);
            const int *drow = v.iat(P.o_drow);
            for (int dc = tid; dc < P.nd; dc += NT)
                alen[dc] = dc < nd_user ? wg_row_len<Mdl>(drow[dc], ph, ch, P.nq, nzu, mi) : (NX > 8 ? P.nq : wg_bound_row_len(nzu));
            T::sync();
            if (tid == 0) { int o = 0; for (int dc = 0; dc < P.nd; ++dc) { aoff[dc] = o; o += alen[dc] & 0xffff; } aoff[P.nd] = o; }
        }
        for (int i = tid; i <= ph; i += NT) {                 // first state bound of state row i + 1 (z entries i NX ..)
            int f = 0;
            while (f < nsb && bnd_idx[f] < i * NX) ++f;
            sbf[i] = f;
        }
        T::sync();
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // Evaluation at z, in three parts.  values_only: the last evaluation of a solve.
    // (1) Mapping::unwrapVector (Mapping.hpp:174-211), Objective::evaluate + computeGradient (Objective.hpp:91-265)
    static MPCX_WG_PHASE void eval_cost(int values_only)
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ph = v.ph, ch = v.ch, nz = v.nz, nxs = v.nxs, nzu = v.nzu;
        const double dv = kDv;
        const double *prm = v.at(P.o_prm), *x0 = v.x0();
        const Scale sc = v.scale();
        double *z = v.at(P.o_z), *Xs = v.at(P.o_Xs), *Us = v.at(P.o_Us), *Jm = v.at(P.o_Jm), *lam = v.at(P.o_lam), *st = v.at(P.o_st),
               *gu = v.at(P.o_gu);
        gwp gxg = (gwp)(v.w + P.w_gx);
        for (int k = tid; k < (ph + 1) * NX; k += NT) {
            const int i = k / NX, j = k - i * NX;
            Xs[k] = sc.over_ss(i == 0 ? x0[j] : z[(i - 1) * NX + j], j);
        }
        for (int k = tid; k < (ph + 1) * NU; k += NT) {
            const int i = k / NU, j = k - i * NU;
            Us[k] = sc.by_su(z[nxs + min(min(i, ph - 1), ch - 1) * NU + j], j);
        }
        T::sync();
        const double e = z[nz - 1];
        auto Xa = [&](int j) { const double a = fabs(Xs[(j % (ph + 1)) * NX + j / (ph + 1)]); return a > 1.0 ? a : 1.0; };
        auto Ua = [&](int j) { const double a = fabs(Us[(j % (ph + 1)) * NU + j / (ph + 1)]); return a > 1.0 ? a : 1.0; };
        const Pert X0{Xs, NX, -1, -1, -1, 0.0}, U0{Us, NU, -1, -1, -1, 0.0};
        double f0;
        if constexpr (Mdl::COST_STAGEWISE) {                    // the rows' shares over the lanes (every lane on the whole horizon: the longest part of this phase)
            double part = 0.0;
            for (int i = tid; i <= ph; i += NT) part += Mdl::stage(i, X0, U0, ph, prm);
            Red<WAVES> R(v.at(P.o_red));
            f0 = R.sum(part) + Mdl::slack_cost(e, prm);
        } else {
            f0 = Mdl::cost(X0, U0, e, ph, prm);
        }
        if (tid == 0) st[ST_COST] = f0;
        if (!values_only) {
            const int nxv = ph * NX, nuv = ph * NU, nall = nxv + nuv + 2;
            const double de = fmax(dv, fabs(e)) * dv;
            for (int idx = tid; idx < nall; idx += NT) {
                const bool isx = idx < nxv, isu = !isx && idx < nxv + nuv;
                const int kk = isx ? idx : idx - nxv;
                const int i = isx ? kk / NX : (isu ? kk / NU : 0), j = isx ? kk - i * NX : (isu ? kk - i * NU : 0);
                const double dx = dv * Xa(j), du = dv * Ua(j);
                const Pert Xp{Xs, NX, isx ? i + 1 : -1, -1, isx ? j : -1, isx ? dx : 0.0};       // no chain rule for the state scaling (Objective.hpp:107-144)
                const Pert Up{Us, NU, isu ? i : -1, (isu && i == ph - 1) ? ph : -1, isu ? j : -1, isu ? du : 0.0};   // the last row moves with its copy
                const double ee = idx == nall - 2 ? e + de : (idx == nall - 1 ? e - de : e);
                if constexpr (Mdl::COST_STAGEWISE) {
                    // the rows of the horizon that do not see the perturbation cancel in f(x + d e) - f(x): only the one that does is evaluated
                    if (isx || isu) {
                        const int r = isx ? i + 1 : i;
                        double fpl = Mdl::stage(r, Xp, Up, ph, prm), f0l = Mdl::stage(r, X0, U0, ph, prm);
                        if (isu && i == ph - 1) { fpl += Mdl::stage(ph, Xp, Up, ph, prm); f0l += Mdl::stage(ph, X0, U0, ph, prm); }
                        const double gk = (fpl - f0l) / (isx ? dx : du);
                        if (isx) { lam[kk] = gk; gxg[kk] = gk; } else Jm[kk] = gk;
                        continue;
                    }
                }
                const double fp = Mdl::COST_STAGEWISE ? f0 - Mdl::slack_cost(e, prm) + Mdl::slack_cost(ee, prm) : Mdl::cost(Xp, Up, ee, ph, prm);
                if (isx) { const double gk = (fp - f0) / dx; lam[kk] = gk; gxg[kk] = gk; }
                else if (isu) Jm[kk] = (fp - f0) / du;
                else st[idx == nall - 2 ? ST_FP : ST_FM] = fp;
            }
            T::sync();
            for (int k = tid; k < nzu; k += NT) {
                const int bl = k / NU, j = k - bl * NU;
                double s = 0;
                for (int i = 0; i < ph; ++i) if (min(i, ch - 1) == bl) s += Jm[i * NU + j];
                gu[k] = sc.by_su(s, j);                                  // Iz2u' * vec(Jmv)
            }
            if (tid == 0) gu[nzu] = (st[ST_FP] - st[ST_FM]) / (2 * de);
        }
        T::sync();
    }

    // (2) Constraints::getStateEqConstraints (Constraints.hpp:490-905): the defects, and the Jacobian blocks folded with E^-1
    static MPCX_WG_PHASE void eval_dyn(int values_only)
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int ph = v.ph;
        const double dv = kDv;
        const double *prm = v.at(P.o_prm);
        const Scale sc = v.scale();
        const double *Xs = v.at(P.o_Xs), *Us = v.at(P.o_Us);
        double *c = v.at(P.o_c);
        const double h = 0.5 * M.Ts;
        if (values_only) {
            for (int i = tid; i < ph; i += NT) {
                double xk[NX], xk1[NX], uk[NU], fa[NX], fb[NX];
                for (int a = 0; a < NX; ++a) { xk[a] = Xs[i * NX + a]; xk1[a] = Xs[(i + 1) * NX + a]; }
                for (int a = 0; a < NU; ++a) uk[a] = Us[i * NU + a];
                Mdl::f(fa, xk, uk, prm);
                if (CT) {
                    Mdl::f(fb, xk1, uk, prm);
                    for (int a = 0; a < NX; ++a) c[i * NX + a] = sc.over_ss(xk[a] + (h * (fa[a] + fb[a])) - xk1[a], a);
                } else {
                    for (int a = 0; a < NX; ++a) c[i * NX + a] = sc.over_ss(xk1[a] - fa[a], a);
                }
            }
        } else if constexpr (!CT) {
            // one-step models: E = I, the folded blocks are the negated Jacobian blocks; one lane per (step, column)
            typename FP::type F = FP::get(v);
            for (int k = tid; k < ph * FW; k += NT) {
                const int i = k / FW, cc = k - i * FW;
                double xk[NX], uk[NU], f1[NX], f2[NX];
                for (int a = 0; a < NX; ++a) xk[a] = Xs[i * NX + a];
                for (int a = 0; a < NU; ++a) uk[a] = Us[i * NU + a];
                const bool isv = cc == FW - 1, isu = cc >= NX;
                const int vv = isu ? cc - NX : cc;
                double base = 0.0;
                for (int a = 0; a < NX; ++a) if (!isv && !isu && a == vv) base = xk[a];
                for (int a = 0; a < NU; ++a) if (!isv && isu && a == vv) base = uk[a];
                const double d = isv ? 0.0 : dv * fmax(fabs(base), 1.0);
                for (int a = 0; a < NX; ++a) if (!isv && !isu && a == vv) xk[a] = base + d;
                for (int a = 0; a < NU; ++a) if (!isv && isu && a == vv) uk[a] = base + d;
                Mdl::f(f1, xk, uk, prm);
                for (int a = 0; a < NX; ++a) if (!isv && !isu && a == vv) xk[a] = base - d;
                for (int a = 0; a < NU; ++a) if (!isv && isu && a == vv) uk[a] = base - d;
                Mdl::f(f2, xk, uk, prm);
                const double i2d = isv ? 0.0 : 1.0 / (2 * d);
                for (int a = 0; a < NX; ++a) {
                    double out;
                    if (isv) {
                        const double cv = sc.over_ss(Xs[(i + 1) * NX + a] - f1[a], a);
                        c[i * NX + a] = cv; out = -cv;
                    } else {
                        const double dcl = (f1[a] - f2[a]) * i2d;
                        // Jacobian entries: -Sx A Tx (state columns), -Sx B su (input columns); folded: their negatives
                        out = isu ? sc.by_su(sc.over_ss(dcl, a), vv) : sc.over_ss(sc.by_ss(dcl, vv), a);
                    }
                    F[(size_t)(i * NX + a) * FW + cc] = out;
                }
            }
        } else {
            // collocation: per step a Gauss-Jordan on [E | A B c | I], one column per lane in registers; G steps per wavefront at a time
            typename FP::type F = FP::get(v);
            gwp einv = (gwp)(v.w + P.w_einv);
            constexpr int G = 64 / GW > 0 ? 64 / GW : 1;
            const int g = lane / GW, cidx = lane - g * GW, base = g * GW;
            // kind of column: 0 E (perturb x_{i+1}), 1 A (perturb x_i), 2 B (perturb u_i), 3 the defect, 4 identity
            const int kind = cidx < NX ? 0 : (cidx < 2 * NX ? 1 : (cidx < 2 * NX + NU ? 2 : (cidx == 2 * NX + NU ? 3 : 4)));
            const int vv = kind == 0 ? cidx : (kind == 1 ? cidx - NX : (kind == 2 ? cidx - 2 * NX : (kind == 4 ? cidx - (2 * NX + NU + 1) : 0)));
            for (int i0 = wave * G; i0 < ph; i0 += WAVES * G) {
                const int i = i0 + g;
                const bool live = g < G && i < ph;
                const int ii = live ? i : ph - 1;
                double col[NX];
                {
                    // TWO evaluations of the vector field per lane, the same call sites for every kind of column (round 6; four until then, half of them
                    // discarded): E and A columns the central difference at x_{i+1} / x_i; the defect column f(x_i, u_i) and f(x_{i+1}, u_i); an input column
                    // needs the central difference at BOTH points -- the one at x_i on its own lane, the one at x_{i+1} on the identity column's lane of
                    // the same number (idle otherwise), handed over by a lane exchange and added in the order the two passes added them.
                    // The point is read from LDS where it is needed: a column keeps one perturbed copy and two results at a time.
                    static_assert(NU <= NX, "the identity columns' lanes lend themselves to the input columns");
                    const double *xr = Xs + ii * NX, *ur = Us + ii * NU;
                    {
                        const bool helper = kind == 4 && vv < NU;
                        const bool isu = kind == 2 || kind == 4, pert = kind <= 2 || helper;
                        const bool n1 = kind == 0 || kind == 4, n2 = n1 || kind == 3;         // the first / second evaluation's point is x_{i+1}
                        double xp[NX], up[NU], o1[NX], o2[NX];
#pragma unroll
                        for (int a = 0; a < NX; ++a) xp[a] = xr[(n1 ? NX : 0) + a];
#pragma unroll
                        for (int a = 0; a < NU; ++a) up[a] = ur[a];
                        double bs = 0.0;
#pragma unroll
                        for (int a = 0; a < NX; ++a) if (!isu && a == vv) bs = xp[a];
#pragma unroll
                        for (int a = 0; a < NU; ++a) if (isu && a == vv) bs = up[a];
                        const double d = pert ? dv * fmax(fabs(bs), 1.0) : 1.0;
#pragma unroll
                        for (int a = 0; a < NX; ++a) if (pert && !isu && a == vv) xp[a] = bs + d;
#pragma unroll
                        for (int a = 0; a < NU; ++a) if (pert && isu && a == vv) up[a] = bs + d;
                        Mdl::f(o1, xp, up, prm);
#pragma unroll
                        for (int a = 0; a < NX; ++a) xp[a] = xr[(n2 ? NX : 0) + a];
#pragma unroll
                        for (int a = 0; a < NX; ++a) if (pert && !isu && a == vv) xp[a] = bs - d;
#pragma unroll
                        for (int a = 0; a < NU; ++a) if (pert && isu && a == vv) up[a] = bs - d;
                        Mdl::f(o2, xp, up, prm);
                        const double i2d = 1.0 / (2 * d);                 // (one division per column; the quotient's last bit is below the differences' noise)
#pragma unroll
                        for (int a = 0; a < NX; ++a) col[a] = kind == 3 ? o1[a] + o2[a] : (o1[a] - o2[a]) * i2d;
                        // the input columns take their second half from the lane that computed it
                        const int src = base + 2 * NX + NU + 1 + (kind == 2 ? vv : 0);
#pragma unroll
                        for (int a = 0; a < NX; ++a) { const double t = __shfl(col[a], src); if (kind == 2) col[a] += t; }
                    }
                    if (!M.scaled) {
                        // (no scalings -- the rule: the five kinds of column by selects, the same arithmetic; as branches on the lane's kind the
                        // sixteen entries were eighty little blocks under exec masks)
                        const double dg = kind == 0 ? -1.0 : 1.0;
#pragma unroll
                        for (int a = 0; a < NX; ++a) {
                            const double hd = h * col[a], x0a = xr[a], x1a = xr[NX + a];
                            const double unit = a == vv ? dg : 0.0;
                            col[a] = kind <= 1 ? unit + hd : (kind == 2 ? hd : (kind == 3 ? x0a + hd - x1a : unit));
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < NX; ++a) {
                            const double da = col[a];
                            double cv;
                            if (kind == 0) cv = (a == vv ? -1.0 : 0.0) + h * sc.over_ss(sc.by_ss(da, vv), a);
                            else if (kind == 1) cv = (a == vv ? 1.0 : 0.0) + h * sc.over_ss(sc.by_ss(da, vv), a);
                            else if (kind == 2) cv = sc.by_su(h * sc.over_ss(da, a), vv);
                            else if (kind == 3) cv = sc.over_ss(xr[a] + (h * da) - xr[NX + a], a);
                            else cv = a == vv ? 1.0 : 0.0;
                            col[a] = cv;
                        }
                    }
                    if (live && kind == 3) for (int a = 0; a < NX; ++a) c[i * NX + a] = col[a];
                }
                // Gauss-Jordan with partial pivoting.  The rows rotate by one per step, so that the pivot row is always register 0 and
                // the loop body is the same for every step (no unrolling over the steps); after NX steps they are back in place.
                // (Where a wavefront holds one tableau -- G == 1: the wide systems -- the pivot column's lane is the same for every lane: v_readlane,
                // four cycles, instead of the LDS crossbar's round trip of a __shfl; 32 of them per pivot.)
                auto bcast = [&](double x, int k) { if constexpr (G == 1) return read_lane(x, k); else return __shfl(x, base + k); };
                auto bcast_i = [&](int x, int k) { if constexpr (G == 1) return __builtin_amdgcn_readlane(x, k); else return __shfl(x, base + k); };
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    // threshold pivoting: the row in place (register 0: the diagonal) stays the pivot while it is within a factor ten of the
                    // column's largest entry -- the rule for E = -I + h df/dx -- and only otherwise the search for the largest runs (where a
                    // wavefront holds one tableau both decisions are scalar branches)
                    int pr = 0;
                    double best = fabs(col[0]), mx = best;
#pragma unroll
                    for (int a = 1; a < NX; ++a) mx = fmax(mx, a < NX - k ? fabs(col[a]) : 0.0);
                    const int search = bcast_i(best < 0.1 * mx ? 1 : 0, k);
                    if (G != 1 || search) {
#pragma unroll
                        for (int a = 1; a < NX; ++a) { const double av = fabs(col[a]); if (a < NX - k && av > best) { best = av; pr = a; } }
                        if (!search) pr = 0;
                        pr = bcast_i(pr, k);                             // the pivot column's choice
                    }
                    double cp = col[0];
                    // (one tableau per wavefront: the choice is the same in every lane -- a scalar branch around the exchange, which a system whose
                    // E is close to -I, h small, never takes)
                    if (G != 1 || pr != 0) {
#pragma unroll
                        for (int a = 1; a < NX; ++a) if (a == pr) { cp = col[a]; col[a] = col[0]; }
                    }
                    const double piv = bcast(cp, k);
                    const double cs = cp / piv;
                    double nxt[NX];
#pragma unroll
                    for (int a = 1; a < NX; ++a) { const double ml = bcast(col[a], k); nxt[a - 1] = fma(-ml, cs, col[a]); }
                    nxt[NX - 1] = cs;
#pragma unroll
                    for (int a = 0; a < NX; ++a) col[a] = nxt[a];
                }
                if (live) {
                    if (kind == 1) for (int a = 0; a < NX; ++a) F[(size_t)(i * NX + a) * FW + vv] = -col[a];
                    else if (kind == 2) for (int a = 0; a < NX; ++a) F[(size_t)(i * NX + a) * FW + NX + vv] = -col[a];
                    else if (kind == 3) for (int a = 0; a < NX; ++a) F[(size_t)(i * NX + a) * FW + FW - 1] = -col[a];
                    else if (kind == 4) for (int a = 0; a < NX; ++a) einv[(size_t)(i * NX + a) * NX + vv] = col[a];
                }
            }
        }
        T::sync();
    }

    // a user row whose reduced entries come from one state row alone (no input enters it): the condensing sweep stores them
    static __device__ __forceinline__ bool sole_state_row(int k, unsigned long long mk, int mi, int ph)
    {
        // (only where the reduced rows live in the workspace: there an add is a dependent trip to memory per entry; in LDS the plain form measured
        // slower -- 324 k instead of 305 k cycles per iteration at config 3 -- and the sixteen-lane sweep of wide states adds throughout)
        if (FL || NX > 8 || __builtin_popcountll(mk) != 1) return false;
        for (int i = 0; i < ph; ++i)
            if (k < mi ? Mdl::ineq_reads_u(k, i) : Mdl::eq_reads_u(k - mi, i)) return false;
        return true;
    }

    // (3) Constraints::evaluateIneq / evaluateEq (Constraints.hpp:211-442) and their Jacobians (computeIneqJacobian :641-721,
    // computeEqJacobian :731-832) as blocks: one NX-vector per (row, state row) pair, the input part straight into the sub-problem's rows
    static MPCX_WG_PHASE void eval_con(int values_only)
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ph = v.ph, ch = v.ch, nz = v.nz, nxs = v.nxs, nzu = v.nzu, mi = v.mi, m = v.m, mt = v.mt, nq = v.nq, ndld = v.ndld;
        const int nsx = P.nsx;
        const double dv = kDv;
        const double *prm = v.at(P.o_prm);
        const Scale sc = v.scale();
        const double *z = v.at(P.o_z), *Xs = v.at(P.o_Xs), *Us = v.at(P.o_Us);
        double *st = v.at(P.o_st), *gin = v.at(P.o_gin), *jx = v.at(P.o_jx), *br = v.at(P.o_br), *s1v = v.at(P.o_s1v);
        typename FP::type art = art_of(v);
        int *s1m = v.iat(P.o_s1m);
        const int *dcol = v.iat(P.o_dcol), *slot = v.iat(P.o_slot), *aoff = v.iat(P.o_aoff), *alen = v.iat(P.o_alen);
        const unsigned long long *xmask = reinterpret_cast<const unsigned long long *>(v.at(P.o_xmask));
        const int *bnd_idx = v.iat(P.o_bidx);
        const double *bnd_sign = v.at(P.o_bsign), *bnd_val = v.at(P.o_bval);
        const double e = z[nz - 1];
        auto Xa = [&](int j) { const double a = fabs(Xs[(j % (ph + 1)) * NX + j / (ph + 1)]); return a > 1.0 ? a : 1.0; };
        auto Ua = [&](int j) { const double a = fabs(Us[(j % (ph + 1)) * NU + j / (ph + 1)]); return a > 1.0 ? a : 1.0; };
        const Pert X0{Xs, NX, -1, -1, -1, 0.0}, U0{Us, NU, -1, -1, -1, 0.0};
        for (int k = tid; k < mi; k += NT) gin[k] = Mdl::ineq(k, X0, U0, e, ph, prm);
        for (int k = tid; k < m - mi; k += NT) gin[mi + k] = Mdl::eq(k, X0, U0, ph, prm);
        T::sync();                                               // (an equality's value is read below by the thread that owns its row)
        if (values_only) return;
        for (int t = tid; t < nsx * NX; t += NT) {
            const int sl = t / NX, j = t - sl * NX, k = slot[sl] >> 8, i = slot[sl] & 0xff;
            double val;
            if (k < mi) {
                const double dx = dv * Xa(j);
                const Pert Xp{Xs, NX, i, -1, j, dx}, Xm{Xs, NX, i, -1, j, -dx};
                val = (Mdl::ineq(k, Xp, U0, e, ph, prm) - Mdl::ineq(k, Xm, U0, e, ph, prm)) / (2 * dx);
            } else {
                const double dx = dv * fmax(fabs(Xs[i * NX + j]), 1.0);
                const Pert Xp{Xs, NX, i, -1, j, dx}, Xm{Xs, NX, i, -1, j, -dx};
                val = (Mdl::eq(k - mi, Xp, U0, ph, prm) - Mdl::eq(k - mi, Xm, U0, ph, prm)) / (2 * dx);
            }
            jx[t] = sc.by_ss(val, j);                            // the state columns are multiplied by the state scaling (Constraints.hpp:269-284)
        }
        // the input part, one lane per user row: into the row's column of art (dense rows) or its (index, value) list
        gwp spv = (gwp)(v.w + P.w_sp);
        int *spi = reinterpret_cast<int *>(v.w + P.w_sp + (size_t)mt * kNlSparse);
        // (Mdl::XFREE_ROWS_AFFINE: a short-list row's entries are the same at every iterate -- differenced once, at the first; what central
        // differences make of a constant afterwards is that constant with nine good digits, a different ninth each time)
        const bool frozen = Mdl::XFREE_ROWS_AFFINE && st[ST_CONSET] != 0.0;
        for (int k = tid; k < m; k += NT) {
            const int dc = dcol[k];
            const bool dense = dc >= 0;
            if (frozen && !dense) { br[k] = gin[k]; continue; }
            const int ro = dense ? aoff[dc] : 0, rl = dense ? alen[dc] : 0;
            if (dense && !(rl >> 30)) for (int q = 0; q < (rl & 0xffff); ++q) art[ro + q] = 0.0;      // (a row only the sweep writes is stored by it)
            int cnt = 0, ix[kNlSparse + 1];
            double ev[kNlSparse + 1];
#pragma unroll
            for (int u = 0; u <= kNlSparse; ++u) { ix[u] = 0; ev[u] = 0.0; }
            auto put = [&](int q, double val) {
                if (q >= nq) return;
                if (dense) { art[ro + q] += val; return; }
                if (val == 0.0) return;                          // the finite differences leave exact zeros outside the structure
                // (every subscript a constant after unrolling: the two little arrays stay in registers -- through run-time subscripts they
                // lived in scratch memory, a trip to HBM per look)
                bool found = false;
#pragma unroll
                for (int u = 0; u < kNlSparse + 1; ++u) {
                    if (!found && u < cnt && ix[u] == q) { ev[u] += val; found = true; }
                }
                if (found) return;
#pragma unroll
                for (int u = 0; u <= kNlSparse; ++u) if (u == cnt) { ix[u] = q; ev[u] = val; }
                ++cnt;
            };
            // (the input rows this constraint reads as a bit mask first, then a walk over its set bits: the lanes' first rows are handled
            // together, then their second ones -- a loop over all rows with a test inside ran the body once per DISTINCT row of a wavefront)
            unsigned long long um = 0;
            for (int i = 0; i < ph; ++i) if (k < mi ? Mdl::ineq_reads_u(k, i) : Mdl::eq_reads_u(k - mi, i)) um |= 1ull << i;
            while (um) {
                const int i = (int)__builtin_ctzll(um);
                um &= um - 1;
                for (int j = 0; j < NU; ++j) {
                    double val;
                    if (k < mi) {
                        const double du = dv * Ua(j);
                        const Pert Up{Us, NU, i, -1, j, du}, Um{Us, NU, i, -1, j, -du};      // every input row on its own (no pairing here)
                        val = (Mdl::ineq(k, X0, Up, e, ph, prm) - Mdl::ineq(k, X0, Um, e, ph, prm)) / (2 * du);
                    } else {
                        const double du = dv * fmax(fabs(Us[(ph - 1) * NU + j]), 1.0);      // row ph-1's magnitude for every step (Constraints.hpp:780,806)
                        const Pert Up{Us, NU, i, i == ph - 1 ? ph : -1, j, du}, Um{Us, NU, i, i == ph - 1 ? ph : -1, j, -du};
                        val = (Mdl::eq(k - mi, X0, Up, ph, prm) - Mdl::eq(k - mi, X0, Um, ph, prm)) / (2 * du);
                    }
                    put(min(i, ch - 1) * NU + j, sc.by_su(val, j));
                }
            }
            if (Mdl::INEQ_USES_SLACK && k < mi) {
                const double de = fmax(dv, fabs(e)) * dv;
                put(nzu, (Mdl::ineq(k, X0, U0, e + de, ph, prm) - Mdl::ineq(k, X0, U0, e - de, ph, prm)) / (2 * de));
            }
            br[k] = gin[k];
            if (!dense) {
                if (cnt > kNlSparse) st[ST_ERR] = 1.0;           // the model's promise (XFREE_ROWS_SPARSE) does not hold
                const int cn = min(cnt, kNlSparse);
                s1v[k] = ev[0]; s1m[k] = (cn << 16) | ix[0];
#pragma unroll
                for (int u = 0; u < kNlSparse; ++u) { spv[k * kNlSparse + u] = ev[u]; spi[k * kNlSparse + u] = ix[u]; }
            } else {
                s1v[k] = 0.0; s1m[k] = kSpDense;
            }
        }
        // rows of the bounds lb <= z + d <= ub (NLOptimizer::setStateBounds / setInputBounds): on an input one entry, on a state a row of Phi
        for (int kb = tid; kb < mt - m; kb += NT) {
            const int zi = bnd_idx[kb], k = m + kb;
            const double sg = bnd_sign[kb];
            br[k] = sg * (z[zi] - bnd_val[kb]);
            if (zi < nxs) {
                const int dc = dcol[k];
                s1v[k] = 0.0; s1m[k] = kSpDense;
                { const int ro = aoff[dc], rl = alen[dc] & 0xffff; for (int q = 0; q < rl; ++q) art[ro + q] = 0.0; }
            } else {
                s1v[k] = sg; s1m[k] = (1 << 16) | (zi - nxs);
                spv[k * kNlSparse] = sg; spi[k * kNlSparse] = zi - nxs;
                for (int u = 1; u < kNlSparse; ++u) { spv[k * kNlSparse + u] = 0.0; spi[k * kNlSparse + u] = 0; }
            }
        }
        T::sync();
        if (tid == 0) st[ST_CONSET] = 1.0;                       // (read at the top of the next call: barriers in between)
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // A chain over the horizon with one column: v_{i+1} = Abar_i v_i + rhs_i (forward) or l_i = -rhs_i + Abar_{i+1}' l_{i+1} (backward),
    // in place in `io` (rhs in, result out), run by wavefront 0.  The column never leaves the registers: lane a of a group of NXP lanes
    // (a quad for NX <= 4, a DPP row of sixteen beyond) holds entry a, the others' entries reach it by DPP broadcasts, its row (column, backward)
    // of the next block is requested before this step's is used; LDS only receives the results.  Every thread of the workgroup
    // calls it; the caller synchronises afterwards.
    static constexpr int NXP = NX <= 4 ? 4 : 16;
    template <int B> static __device__ __forceinline__ double chain_bcast(double x)
    {
        if constexpr (NXP == 4) return dpp_d<B * 0x55>(x);        // quad_perm [B, B, B, B]
        else return row_share<0x150 + B>(x);
    }
    template <int B> struct ChainDot {
        static __device__ __forceinline__ double run(const double (&f)[NX], double x, double acc)
        {
            if constexpr (B < NX) return ChainDot<B + 1>::run(f, x, fma(f[B], chain_bcast<B>(x), acc));
            else return acc;
        }
    };
    template <bool BACKWARD>
    static __device__ __forceinline__ void chain(const V &v, double *io, int tid)
    {
        static_assert(NX <= 16, "a chain's column lives in one DPP row");
        if (tid >= 64) return;
        const int ph = v.ph;
        typename FP::type F = FP::get(v);
        const int a = tid & (NXP - 1), aa = min(a, NX - 1);
        const bool alive = a < NX, writes = alive && tid < NXP;
        // forward: step i uses block i (from i = 1 on; v_0 = 0); backward: step i uses block i + 1 (up to i = ph - 2; l_ph = 0).
        // The blocks of the next D steps are in flight while one is used: one step ahead hides an LDS access, four a trip to the workspace
        // (where the blocks of a wide system live: a step's arithmetic is a few hundred cycles, the trip a thousand and more).
        constexpr int D = FL ? 1 : 4;
        double fq[D][NX], rq[D];
        auto blk_of = [&](int step) { return BACKWARD ? ph - step : step; };            // (step >= 1)
        auto row_of = [&](int step) { return BACKWARD ? ph - 1 - step : step; };
        auto fetch = [&](int d, int step) {                     // this lane's row of the step's block (its column, backward) and right-hand side
            const int blk = blk_of(step);
#pragma unroll
            for (int bb = 0; bb < NX; ++bb)
                fq[d][bb] = (!alive || step == 0) ? 0.0 : (BACKWARD ? F[(size_t)(blk * NX + bb) * FW + aa] : F[(size_t)(blk * NX + aa) * FW + bb]);
            rq[d] = alive ? io[row_of(step) * NX + aa] : 0.0;
        };
#pragma unroll
        for (int d = 0; d < D; ++d) if (d < ph) fetch(d, d);
        double x = 0.0;
        for (int step0 = 0; step0 < ph; step0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int step = step0 + d;
                if (step < ph) {
                    double fc[NX];
#pragma unroll
                    for (int bb = 0; bb < NX; ++bb) fc[bb] = fq[d][bb];
                    const double rc = rq[d];
                    if (step + D < ph) fetch(d, step + D);
                    const double s = ChainDot<0>::run(fc, x, 0.0);       // (zero on the first step: fc = 0)
                    x = alive ? (BACKWARD ? s - rc : s + rc) : 0.0;
                    if (writes) io[row_of(step) * NX + aa] = x;
                }
            }
        }
        nl_wave_sync();
    }

    // the multipliers of the defects themselves from the chain's values, lam_i = E_i^-T l_i: this thread's share of max |lam|
    static __device__ __forceinline__ double defect_multiplier_max(const V &v, const double *lam, int tid)
    {
        double lmax = 0.0;
        if (CT) {
            gwp einv = (gwp)(v.w + v.A->P.w_einv);
            for (int k = tid; k < v.nxs; k += NT) {
                const int i = k / NX, a = k - i * NX;
                double s = 0.0;
#pragma unroll
                for (int bb = 0; bb < NX; ++bb) s = fma(einv[(size_t)(i * NX + bb) * NX + a], lam[i * NX + bb], s);
                lmax = fmax(lmax, fabs(s));
            }
        } else {
            for (int k = tid; k < v.nxs; k += NT) lmax = fmax(lmax, fabs(lam[k]));
        }
        return lmax;
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // condense + reduce where a row of the sub-problem reads a state: one column of [Phi | r] per lane (NX <= 8: in registers) or per
    // sixteen lanes (wider states: one entry per lane, the row products by DPP row_share), dx_{i+1} = Abar_i dx_i + (Bbar_i e_q | cbar_i).
    // What the column is needed for is taken as the sweep passes: Phi' g_x, the rows of the user Jacobian and of the state bounds
    // that read state row i + 1.
    template <int BB> struct RowShare {
        template <class FT> static __device__ __forceinline__ double dot(FT frow, double x, double acc)
        {
            if constexpr (BB < NX) return RowShare<BB + 1>::dot(frow, x, fma(frow[BB], row_share<0x150 + BB>(x), acc));
            else return acc;
        }
    };
    static MPCX_WG_PHASE void condense_phi()
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ph = v.ph, ch = v.ch, nzu = v.nzu, mi = v.mi, m = v.m, ndld = v.ndld;
        const double *lam = v.at(P.o_lam), *gu = v.at(P.o_gu), *jx = v.at(P.o_jx);
        double *gr = v.at(P.o_gr), *br = v.at(P.o_br);
        typename FP::type art = art_of(v);
        const int *dcol = v.iat(P.o_dcol), *jxoff = v.iat(P.o_jxoff), *sbf = v.iat(P.o_sbf), *aoff = v.iat(P.o_aoff), *alen = v.iat(P.o_alen);
        (void)lam; (void)dcol; (void)sbf; (void)alen;
        const unsigned long long *xmask = reinterpret_cast<const unsigned long long *>(v.at(P.o_xmask));
        const int *bnd_idx = v.iat(P.o_bidx);
        const double *bnd_sign = v.at(P.o_bsign);
        typename FP::type F = FP::get(v);
        if constexpr (NX <= 8) {
            // Row by row, backwards (the adjoint of the sweep): a dense row of the sub-problem is a covector w on the state rows it reads, and
            //     l_top = w_top,   l_{s-1} = w_{s-1} + Abar_s' l_s,   entries of the row on input block b(s) += Bbar_s' l_s,   offset += cbar_s' l_s
            // -- one lane per row, the multiplier l in its registers, every lane on the same step s (the block's entries are broadcast reads), the
            // row's results filed as the sweep passes; nothing is looked up along the way (the forward form walked, per column and state row, the
            // list of rows that read it: five dependent LDS reads per entry).  Wavefront 0 meanwhile runs the one chain of the reduced gradient
            // (lam: g_x in, the chain's multipliers out -- as where no row reads a state), the others share the rows.
            double *lamw = v.at(P.o_lam);
            chain<true>(v, lamw, tid);
            const int nd = P.nd, nd_user = P.nd_user, nq = v.nq;
            const int *drow = v.iat(P.o_drow);
            constexpr bool PF = NX <= 4;                         // the step's block in registers, requested in one batch (small states)
            constexpr int RW = WAVES > 1 ? NT - 64 : 64;         // threads that share the rows
            for (int dc = WAVES > 1 ? tid - 64 : tid; dc >= 0 && dc < nd; dc += RW) {
                const bool user = dc < nd_user;
                const int k = user ? drow[dc] : m + (dc - nd_user);
                const unsigned long long mk = user ? xmask[k] : 0ull;
                const int jo = user ? jxoff[k] : 0;
                const int zi = user ? 0 : bnd_idx[dc - nd_user];
                const int bs = user ? -1 : zi / NX, ba = user ? 0 : zi - (zi / NX) * NX;     // a bound: state row and entry
                const double bsg = user ? 0.0 : bnd_sign[dc - nd_user];
                const int top = user ? (mk ? 63 - __builtin_clzll(mk) : -1) : bs;           // the last state row the row reads
                double l[NX], acc[NU], cst = 0.0, old[NU];
#pragma unroll
                for (int a = 0; a < NX; ++a) l[a] = 0.0;
#pragma unroll
                for (int jq = 0; jq < NU; ++jq) acc[jq] = 0.0;
                double fb[PF ? NX * FW : 1];
                auto fetch = [&](int sI) {
#pragma unroll
                    for (int e = 0; e < NX * FW; ++e) fb[e] = F[(size_t)sI * NX * FW + e];
                };
                const int ro = aoff[dc];
                const bool stores = (alen[dc] >> 30) != 0;      // nothing else writes this row: its entries are stored, not added
                auto fetch_old = [&](int blk) {
#pragma unroll
                    for (int jq = 0; jq < NU; ++jq) old[jq] = stores ? 0.0 : art[ro + blk * NU + jq];
                };
                // (every lane walks the whole horizon, idle above its own top: the same step everywhere, so that the block's entries are one broadcast)
                if constexpr (PF) fetch(ph - 1);
                if (top >= 0) fetch_old(min(top, ch - 1));
                for (int sI = ph - 1; sI >= 0; --sI) {
                    if (sI <= top) {
                        // w_s joins the multiplier
                        if (user) {
                            if ((mk >> sI) & 1ull) {
                                const int sl = jo + __builtin_popcountll(mk & ((1ull << sI) - 1ull));
#pragma unroll
                                for (int a = 0; a < NX; ++a) l[a] += jx[sl * NX + a];
                            }
                        } else if (sI == bs) {
#pragma unroll
                            for (int a = 0; a < NX; ++a) if (a == ba) l[a] += bsg;
                        }
                        auto Fe = [&](int a, int c) -> double { if constexpr (PF) return fb[a * FW + c]; else return F[(size_t)(sI * NX + a) * FW + c]; };
                        // inputs and offset
#pragma unroll
                        for (int jq = 0; jq < NU; ++jq) {
                            double sacc = acc[jq];
#pragma unroll
                            for (int a = 0; a < NX; ++a) sacc = fma(Fe(a, NX + jq), l[a], sacc);
                            acc[jq] = sacc;
                        }
#pragma unroll
                        for (int a = 0; a < NX; ++a) cst = fma(Fe(a, FW - 1), l[a], cst);
                        if (sI <= ch - 1) {                        // the last step of this input block: file it, ask for the next block's entries
#pragma unroll
                            for (int jq = 0; jq < NU; ++jq) { art[ro + sI * NU + jq] = old[jq] + acc[jq]; acc[jq] = 0.0; }
                            if (sI > 0) fetch_old(sI - 1);
                        }
                        // l_{s-1} = Abar_s' l_s (w_{s-1} joins at the top of the next step)
                        if (sI > 0) {
                            double ln[NX];
#pragma unroll
                            for (int b2 = 0; b2 < NX; ++b2) {
                                double sacc = 0.0;
#pragma unroll
                                for (int a = 0; a < NX; ++a) sacc = fma(Fe(a, b2), l[a], sacc);
                                ln[b2] = sacc;
                            }
#pragma unroll
                            for (int a = 0; a < NX; ++a) l[a] = ln[a];
                        }
                    }
                    if constexpr (PF) { if (sI > 0) fetch(sI - 1); }     // (all of the next block's entries in one batch of broadcast reads)
                }
                if (top >= 0) br[k] += cst;
            }
            T::sync();
            // the reduced gradient from the chain's multipliers (lam = -l: Jx' lam = -g_x)
            for (int q = tid; q < nzu; q += NT) {
                const int bq = q / NU, jq = q - bq * NU;
                double sacc = gu[q];
                for (int i = bq; i < (bq == ch - 1 ? ph : bq + 1); ++i) {
#pragma unroll
                    for (int a = 0; a < NX; ++a) sacc = fma(-F[(size_t)(i * NX + a) * FW + NX + jq], lamw[i * NX + a], sacc);
                }
                gr[q] = sacc;
            }
            (void)nq; (void)mi;
        } else {
            static_assert(NX <= 16, "the workgroup form spreads a column over one DPP row");
            const int a = tid & 15, aa = min(a, NX - 1);
            const bool alive = a < NX;
            for (int q0 = 0; q0 <= nzu; q0 += NT / 16) {
                const int q = q0 + (tid >> 4);
                const bool qlive = q <= nzu, isr = q == nzu;
                const int bq = q / NU, jq = q - bq * NU;
                double x = 0.0, gacc = 0.0;
                for (int i = 0; i < ph; ++i) {
                    const bool drives = qlive && !isr && min(i, ch - 1) == bq;
                    const double rh = !alive ? 0.0 : (isr ? F[(size_t)(i * NX + aa) * FW + FW - 1] : (drives ? F[(size_t)(i * NX + aa) * FW + NX + jq] : 0.0));
                    const double t = RowShare<0>::dot(F + (size_t)(i * NX + aa) * FW, x, rh);
                    x = alive ? t : 0.0;
                    gacc = fma(alive ? lam[i * NX + aa] : 0.0, x, gacc);
                    auto row = [&](int k) {
                        const int sl = jxoff[k] + __builtin_popcountll(xmask[k] & ((1ull << i) - 1ull));
                        const double s = group_sum<16>(alive ? jx[sl * NX + aa] * x : 0.0);
                        if (qlive && a == 0) { if (isr) br[k] += s; else art[aoff[dcol[k]] + q] += s; }
                    };
                    int first, count;
                    Mdl::ineq_rows_of_x(i + 1, first, count);
                    for (int k = first; k < first + count; ++k) if ((xmask[k] >> i) & 1ull) row(k);
                    for (int k = mi; k < m; ++k) if ((xmask[k] >> i) & 1ull) row(k);
                    for (int kb = sbf[i]; kb < sbf[i + 1]; ++kb) {
                        if (qlive && bnd_idx[kb] - i * NX == a) {
                            const double sg = bnd_sign[kb];
                            if (isr) br[m + kb] += sg * x; else art[aoff[dcol[m + kb]] + q] = sg * x;
                        }
                    }
                }
                gacc = group_sum<16>(gacc);
                if (qlive && !isr && a == 0) gr[q] = gu[q] + gacc;
            }
        }
        if (tid == 0) gr[nzu] = gu[nzu];
        T::sync();
    }
    // ... and where none does: the reduced gradient through one backward chain (Jx' lam = -g_x; lam holds g_x on entry, the chain's
    // multipliers afterwards).  Leaves the largest dynamics multiplier, which the merit weight has to dominate, in st[ST_LAMDYN].
    static MPCX_WG_PHASE void condense_chain()
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ph = v.ph, ch = v.ch, nzu = v.nzu;
        double *lam = v.at(P.o_lam), *gu = v.at(P.o_gu), *gr = v.at(P.o_gr), *st = v.at(P.o_st);
        typename FP::type F = FP::get(v);
        chain<true>(v, lam, tid);
        T::sync();
        for (int q = tid; q <= nzu; q += NT) {
            if (q == nzu) { gr[q] = gu[q]; continue; }
            const int bq = q / NU, jq = q - bq * NU;
            double s = gu[q];
            for (int i = bq; i < (bq == ch - 1 ? ph : bq + 1); ++i) {
#pragma unroll
                for (int a = 0; a < NX; ++a) s = fma(-F[(size_t)(i * NX + a) * FW + NX + jq], lam[i * NX + a], s);
            }
            gr[q] = s;
        }
        Red<WAVES> R(v.at(P.o_red));
        const double lmax = R.max(defect_multiplier_max(v, lam, tid));
        if (tid == 0) st[ST_LAMDYN] = lmax;
        T::sync();
    }

    // The dense rows of the sub-problem, row by row (wg_row_len).  yd[dc] = (row dc)' x for every dense row: sixteen lanes share a row (its
    // entries are contiguous: one 128-byte line per group and pass).  Every thread calls; the caller synchronises.
    static __device__ __forceinline__ void art_tmul(const V &v, const double *x, double *yd, int tid)
    {
        const auto &P = v.A->P;
        const int nd = P.nd, l = tid & 15;
        const int *aoff = v.iat(P.o_aoff), *alen = v.iat(P.o_alen);
        typename FP::type art = art_of(v);
        for (int d0 = 0; d0 < nd; d0 += NT / 16) {
            const int dc = d0 + (tid >> 4);
            const bool live = dc < nd;
            const int ro = live ? aoff[dc] : 0, len = live ? alen[dc] & 0xffff : 0;
            double a0 = 0.0, a1 = 0.0;
            int q = l;
            for (; q + 16 < len; q += 32) { a0 = fma(art[ro + q], x[q], a0); a1 = fma(art[ro + q + 16], x[q + 16], a1); }
            for (; q < len; q += 16) a0 = fma(art[ro + q], x[q], a0);
            const double acc = group_sum<16>(a0 + a1);
            if (live && l == 0) yd[dc] = acc;
        }
    }
    // yd[dc] = (row dc)' x for the dense rows of the WORKING set only (what t = N_W B^-1 n and the warm start's multipliers need: a dozen rows of
    // sixty -- the scan is the one product that needs them all).  Sixteen lanes a row; every thread calls; the caller synchronises.
    static __device__ __forceinline__ void art_ws_tmul(const V &v, int nw, const double *x, double *yd, int tid)
    {
        const auto &P = v.A->P;
        const int l = tid & 15;
        const int *aoff = v.iat(P.o_aoff), *alen = v.iat(P.o_alen), *wq = v.iat(P.o_wq), *dcol = v.iat(P.o_dcol);
        typename FP::type art = art_of(v);
        for (int t0 = 0; t0 < nw; t0 += NT / 16) {
            const int t = t0 + (tid >> 4);
            const int dc = t < nw ? dcol[wq[t]] : -1;
            const int ro = dc >= 0 ? aoff[dc] : 0, len = dc >= 0 ? alen[dc] & 0xffff : 0;
            double a0 = 0.0, a1 = 0.0;
            int q = l;
            for (; q + 16 < len; q += 32) { a0 = fma(art[ro + q], x[q], a0); a1 = fma(art[ro + q + 16], x[q + 16], a1); }
            for (; q < len; q += 16) a0 = fma(art[ro + q], x[q], a0);
            const double acc = group_sum<16>(a0 + a1);
            if (dc >= 0 && l == 0) yd[dc] = acc;
        }
    }
    // out[q] += sum over the dense rows of the working set of cd[row] * (row)[q]: a loop over the WORKING rows (a dozen, not all sixty: the
    // product with the whole matrix read every row to multiply most of them by zero), a lane per entry.  Every thread calls; the caller synchronises.
    static __device__ __forceinline__ void art_ws_mul(const V &v, int nw, const double *cd, double *out, int tid)
    {
        const auto &P = v.A->P;
        const int nq = v.nq;
        const int *aoff = v.iat(P.o_aoff), *alen = v.iat(P.o_alen), *wq = v.iat(P.o_wq), *dcol = v.iat(P.o_dcol);
        typename FP::type art = art_of(v);
        for (int q = tid; q < nq; q += NT) {
            double acc = 0.0;
            // four rows at a time: their look-ups (row -> dense index -> offset, length, coefficient) and their entries are requested together --
            // one row at a time every step was four dependent trips to LDS and one to the rows
            for (int t0 = 0; t0 < nw; t0 += 4) {
                int dc[4], ro[4], rl[4];
                double cf[4], av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) dc[u] = t0 + u < nw ? dcol[wq[t0 + u]] : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int d = dc[u] >= 0 ? dc[u] : 0; ro[u] = aoff[d]; rl[u] = dc[u] >= 0 ? alen[d] & 0xffff : 0; cf[u] = cd[d]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) av[u] = q < rl[u] ? art[ro[u] + q] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = fma(av[u], cf[u], acc);
            }
            out[q] += acc;
        }
    }
    // the working set as it stands in LDS: row numbers, orientations, and where the rows' entries are (derived once per phase,
    // outside any loop over lanes)
    struct Ws {
        const int *wq, *dcol;
        const double *sgq;
        double *cd, *yd;
        int nd;
        __device__ __forceinline__ explicit Ws(const V &v)
            : wq(v.iat(v.A->P.o_wq)), dcol(v.iat(v.A->P.o_dcol)), sgq(v.at(v.A->P.o_sgq)), cd(v.at(v.A->P.o_cd)), yd(v.at(v.A->P.o_yd)), nd(v.A->P.nd) {}
    };
    // N_W' r over the SPARSE working rows, two ways (kOneEntry, a property of the model):
    //   * every short-list row has one entry (bounds, the examples' u_j <= 0.5): two such rows on one variable are parallel, the dual method never
    //     holds two parallel rows, so at most one working row touches a variable -- the rows' owners scatter by LDS atomic adds that cannot meet;
    //   * otherwise (rows with several entries, several rows on one variable): every variable gathers its contributions in the working set's
    //     order, ml[t] the rows' oriented coefficients in LDS -- the same bits on every run, whatever the scheduling.
    static constexpr bool kOneEntry = Mdl::SPARSE_ROWS_ONE_ENTRY || !Mdl::XFREE_ROWS_SPARSE;
    // (gather form) out[q] = sum over the sparse working rows t, in order, of ml[t] * (row t's entry on q), q = id, id + nth, ..
    static __device__ __forceinline__ void sparse_gather(const V &v, const Sp &sp, const Ws &W, int nw, const double *ml, double *out, int id, int nth)
    {
        for (int q = id; q < v.nq; q += nth) {
            double acc = 0.0;
            for (int t = 0; t < nw; ++t) {
                const int k = W.wq[t];
                if (W.dcol[k] >= 0) continue;
                const int cn = sp.count(k);
                for (int e = 0; e < cn; ++e) if (sp.index(k, e) == q) acc = fma(sp.value(k, e), ml[t], acc);
            }
            out[q] = acc;
        }
    }
    // out[q] = sum_t coef[t] * (oriented normal of working row t)[q] for q < nq: the sparse rows as above, the dense rows go through art.  Arrays
    // by their LDS offsets; every thread calls; synchronised on return.  (Out of line, like everything the sub-problem uses more than once.)
    static MPCX_WG_CALL void ws_nt_mul(int nw, int coef_off, int out_off)
    {
        const V v; const Sp sp(v); const Ws W(v);
        const int tid = threadIdx.x, nq = v.nq;
        const double *coef = v.at(coef_off);
        double *out = v.at(out_off);
        for (int q = tid; q < nq; q += NT) out[q] = 0.0;
        for (int dc = tid; dc < W.nd; dc += NT) W.cd[dc] = 0.0;
        T::sync();
        if constexpr (kOneEntry) {
            for (int t = tid; t < nw; t += NT) {
                const int k = W.wq[t], dc = W.dcol[k];
                const double ml = W.sgq[t] * coef[t];
                if (dc >= 0) W.cd[dc] = ml;
                else {
                    const int cn = sp.count(k);
                    for (int j = 0; j < cn; ++j) atomicAdd(out + sp.index(k, j), sp.value(k, j) * ml);
                }
            }
        } else {
            double *ml = v.at(v.A->P.o_tq);                      // (the dual step's scratch: free between steps)
            for (int t = tid; t < nw; t += NT) {
                const int dc = W.dcol[W.wq[t]];
                ml[t] = W.sgq[t] * coef[t];
                if (dc >= 0) W.cd[dc] = ml[t];
            }
            T::sync();
            sparse_gather(v, sp, W, nw, ml, out, tid, NT);
        }
        T::sync();
        if (W.nd > 0) { art_ws_mul(v, nw, W.cd, out, tid); T::sync(); }
    }
    // out[t] = (oriented normal of working row t)' x for t < nw
    static MPCX_WG_CALL void ws_n_mul(int nw, int x_off, int out_off)
    {
        const V v; const Sp sp(v); const Ws W(v);
        const int tid = threadIdx.x;
        const double *x = v.at(x_off);
        double *out = v.at(out_off);
        if (W.nd > 0) { art_tmul(v, x, W.yd, tid); T::sync(); }
        for (int t = tid; t < nw; t += NT) {
            const int k = W.wq[t], dc = W.dcol[k];
            out[t] = W.sgq[t] * (dc >= 0 ? W.yd[dc] : sp.dot(k, x));
        }
        T::sync();
    }
    // dst = scale * B^-1 src
    static MPCX_WG_CALL void hmul_call(int src_off, int dst_off, double scale)
    {
        const V v;
        hmul<NT>(v.at(v.A->P.o_hinv), v.at(src_off), v.at(dst_off), v.nq, scale, threadIdx.x);
        T::sync();
    }
    // wv += art cd: the dense rows' part of N_W' rr (cd: their coefficients, scattered by the dual part of the step)
    static MPCX_WG_CALL void art_mul_call(int nw)
    {
        const V v; const auto &P = v.A->P;
        art_ws_mul(v, nw, v.at(P.o_cd), v.at(P.o_wv), threadIdx.x);
        T::sync();
    }
    // xq -= t (vv - B^-1 wv): the primal part of a dual step
    // (with_v false: xq += t B^-1 wv -- the warm start's minimiser on the kept rows with t = -1)
    static MPCX_WG_CALL void hmul_step_call(double t, bool with_v = true)
    {
        const V v; const auto &P = v.A->P;
        hmul<NT, true>(v.at(P.o_hinv), v.at(P.o_wv), v.at(P.o_xq), v.nq, t, threadIdx.x, with_v ? v.at(P.o_vv) : nullptr);
        T::sync();
    }
    // the oriented normal n = sgn * (row k) of a sub-problem row into np and vv = B^-1 n.  For the dual method (sums): returns n' B^-1 n and
    // n'n, and leaves (row)' vv of the nw_t working rows' dense ones in yd -- the dense rows' part of N_W B^-1 n, for whoever gathers the
    // working set's entries; no barrier behind the reduction (slot set 2).  For the warm start's row-by-row Schur complement: np and vv only.
    static MPCX_WG_CALL WgSum2 normal_call(int k, double sgn, bool sums = true, int nw_t = 0)
    {
        const V v; const auto &P = v.A->P; const Sp sp(v);
        const int tid = threadIdx.x, nq = v.nq, dc = v.iat(P.o_dcol)[k];
        const double *hinv = v.at(P.o_hinv);
        typename FP::type art = art_of(v);
        double *np_ = v.at(P.o_np), *vv = v.at(P.o_vv);
        if (dc >= 0) {
            const int ro = v.iat(P.o_aoff)[dc], rl = v.iat(P.o_alen)[dc] & 0xffff;
            for (int q = tid; q < nq; q += NT) np_[q] = q < rl ? sgn * art[ro + q] : 0.0;
            T::sync();
            hmul<NT>(hinv, np_, vv, nq, 1.0, tid);
        } else {
            const int cn = sp.count(k);
            for (int q = tid; q < nq; q += NT) {
                double nvl = 0, hv = 0;
                for (int j = 0; j < cn; ++j) {
                    const int ix = sp.index(k, j);
                    const double val = sgn * sp.value(k, j);
                    if (ix == q) nvl += val;
                    hv = fma(hsym(hinv, ix, q), val, hv);
                }
                np_[q] = nvl; vv[q] = hv;
            }
        }
        T::sync();
        if (!sums) return WgSum2{0.0, 0.0};
        double snn = 0, npn = 0;
        for (int q = tid; q < nq; q += NT) { snn += np_[q] * vv[q]; npn += np_[q] * np_[q]; }
        if (nw_t > 0 && P.nd > 0) art_ws_tmul(v, nw_t, vv, v.at(P.o_yd), tid);
        if (P.minv && P.nd == 0 && v.at(P.o_st)[ST_MINV] != 0.0) {
            // (the inverse form, every row a short list: t = N_W B^-1 n and the zeros N_W' r starts from, under the reduction's barrier --
            // what ws_dual_step_m would otherwise spend a phase on)
            const Ws W(v);
            double *tq = v.at(P.o_tq), *wv = v.at(P.o_wv);
            for (int t = tid; t < nw_t; t += NT) tq[t] = W.sgq[t] * sp.dot(W.wq[t], vv);
            for (int q = tid; q < nq; q += NT) wv[q] = 0.0;
        }
        Red<WAVES> R(v.at(P.o_red), 2);
        R.sum2(snn, npn);
        return WgSum2{snn, npn};
    }
    // the most violated row outside the working set at xq (an equality is violated on either side); yd keeps art' xq
    static MPCX_WG_CALL WgArgmax scan_call()
    {
        const V v; const auto &P = v.A->P; const Sp sp(v); const Ws W(v);
        const int tid = threadIdx.x, mi = v.mi, m = v.m, mt = v.mt;
        const double *xq = v.at(P.o_xq), *br = v.at(P.o_br);
        const int *flag = v.iat(P.o_flag);
        if (W.nd > 0) { art_tmul(v, xq, W.yd, tid); T::sync(); }
        double vmax = -1e300; int pidx = 0x7fffffff;
        for (int k = tid; k < mt; k += NT) {
            const int dc = W.dcol[k];
            double s = br[k] + (dc >= 0 ? W.yd[dc] : sp.dot(k, xq));
            if (k >= mi && k < m) s = fabs(s);
            if (flag[k] == 0 && s > vmax) { vmax = s; pidx = k; }
        }
        Red<WAVES> R(v.at(P.o_red), 1);
        R.argmax(vmax, pidx);
        return WgArgmax{vmax, pidx};
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // damped BFGS update of the inverse Hessian estimate (Powell): s = a p, y = change of the reduced Lagrangian gradient
    static MPCX_WG_PHASE void bfgs(double a_prev, int nw_keep)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int nq = v.nq;
        const Sp sp(v);
        const Ws W(v);
        double *gr = v.at(P.o_gr), *glold = v.at(P.o_glold), *sv = v.at(P.o_sv), *hinv = v.at(P.o_hinv), *uq = v.at(P.o_uq), *st = v.at(P.o_st);
        double *v0 = v.at(P.o_xq), *v1 = v.at(P.o_np), *v2 = v.at(P.o_vv);
        Red<WAVES> R(v.at(P.o_red));
        double sBs = 0, sy = 0;
        ws_nt_mul(nw_keep, P.o_uq, P.o_vv);                      // N' u with the previous multipliers and this point's rows
        for (int q = tid; q < nq; q += NT) {
            const double gl = gr[q] + v2[q];
            const double y = gl - glold[q], Bs = -a_prev * glold[q];
            v0[q] = y; v1[q] = Bs;
            sBs += sv[q] * Bs; sy += sv[q] * y;
        }
        R.sum2(sBs, sy);
        if (sy < 0.2 * sBs) {
            const double th = 0.8 * sBs / (sBs - sy);
            for (int q = tid; q < nq; q += NT) v0[q] = th * v0[q] + (1 - th) * v1[q];
            sy = th * sy + (1 - th) * sBs;
        }
        T::sync();
        if (sy > 1e-300) {
            const double rho = 1.0 / sy;
            hmul_call(P.o_xq, P.o_vv, 1.0);
            double yHy = 0;
            for (int q = tid; q < nq; q += NT) yHy += v2[q] * v0[q];
            yHy = R.sum(yHy);
            const double cc = rho * rho * yHy + rho;
            for (int e = tid; e < nq * (nq + 1) / 2; e += NT) {
                int r, c;
                tri_index(e, r, c);
                hinv[e] += -rho * (sv[r] * v2[c] + v2[r] * sv[c]) + cc * sv[r] * sv[c];
            }
            // (the change of B^-1 is [s v2] [cc, -rho; -rho, 0] [s v2]': what a carried inverse of the Schur complement needs -- ws_warm; sv and
            // v2 stay where they are until then)
            if (tid == 0 && st[ST_CARRY] == 1.0) { st[ST_CARRY] = 2.0; st[ST_BFRHO] = rho; st[ST_BFCC] = cc; }
        }
        T::sync();
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // row t leaves the working set of nw rows: the factor is down-dated, the lists close up
    static MPCX_WG_PHASE void ws_drop(int kdrop, int nw)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x, lane = tid & 63;
        double *sgq = v.at(P.o_sgq), *uq = v.at(P.o_uq);
        int *wq = v.iat(P.o_wq), *flag = v.iat(P.o_flag);
        if (tid < 64) chol_delete(v.at(P.o_L), v.at(P.o_invd), nw, kdrop, lane);
        int kq = 0, kq2 = 0;
        double sg = 0, u = 0, sg2 = 0, u2 = 0;
        const bool mv = tid > kdrop && tid < nw;                 // (working sets hold at most 128 rows: one or two per thread)
        const bool mv2 = tid + NT > kdrop && tid + NT < nw;
        if (mv) { kq = wq[tid]; sg = sgq[tid]; u = uq[tid]; }
        if (mv2) { kq2 = wq[tid + NT]; sg2 = sgq[tid + NT]; u2 = uq[tid + NT]; }
        if (tid == 0) flag[wq[kdrop]] = 0;
        T::sync();
        if (mv) { wq[tid - 1] = kq; sgq[tid - 1] = sg; uq[tid - 1] = u; }
        if (mv2) { wq[tid + NT - 1] = kq2; sgq[tid + NT - 1] = sg2; uq[tid + NT - 1] = u2; }
        T::sync();
    }
    // The dual part of a step of the dual method, by wavefront 0 with the working set's vectors in its registers: t = N_W B^-1 n gathered
    // (yd holds art' B^-1 n), rr = S^-1 t, N_W' rr set up for the primal part (sparse rows scattered into wv, the dense rows' coefficients
    // into cd), the curvature along the step z'n = n'B^-1 n - |L^-1 t|^2 (what the factor's
    // next diagonal is the square root of), the ratio test over the multipliers, the step length, the multipliers' update and -- on a full
    // step -- the entering row's place in the factor and in the lists.  Results for everybody in st[R0 ..]: step length (1e300: none),
    // z'n, what happens (0 no step: the row depends on the working set; 1 the row joins; 2 row st[R3] leaves first), that row.
    static MPCX_WG_PHASE void ws_dual_step(int nw, int pidx, double sgn, double snn, double npn, double spv, double up)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x, lane = tid & 63, mi = v.mi, m = v.m;
        double *tq = v.at(P.o_tq), *uq = v.at(P.o_uq), *sgq = v.at(P.o_sgq), *st = v.at(P.o_st);
        int *wq = v.iat(P.o_wq), *flag = v.iat(P.o_flag);
#ifdef MPCX_NL_STATS
        long long qt_ = __builtin_readcyclecounter();          // (the parts of this phase: st[ST_QSTAT + 11 ..], as the inverse form's)
#endif
        if (tid < 64) {
            const Sp sp(v);
            double *Lp = v.at(P.o_L), *invd = v.at(P.o_invd), *cd = v.at(P.o_cd), *wv = v.at(P.o_wv);
            const double *yd = v.at(P.o_yd), *vv = v.at(P.o_vv);
            const int *dcol = v.iat(P.o_dcol);
            const int nd = P.nd, nq = v.nq;
            const bool h0 = lane < nw, h1 = lane + 64 < nw;
            const int k0 = h0 ? wq[lane] : 0, k1 = h1 ? wq[lane + 64] : 0;
            const int d0 = h0 ? dcol[k0] : -1, d1 = h1 ? dcol[k1] : -1;
            const double s0 = h0 ? sgq[lane] : 0.0, s1 = h1 ? sgq[lane + 64] : 0.0;
            // (requested before the substitutions, used after them)
            const double u0 = h0 ? uq[lane] : 0.0, u1 = h1 ? uq[lane + 64] : 0.0;
            // t = N_W v: a dense row's entry is in yd (the entering row's phase left art' v there), a sparse row's a few products
            double t0 = !h0 ? 0.0 : s0 * (d0 >= 0 ? yd[d0] : sp.dot(k0, vv)), t1 = !h1 ? 0.0 : s1 * (d1 >= 0 ? yd[d1] : sp.dot(k1, vv));
            double y0 = 0.0, y1 = 0.0;
            // N_W' rr starts from zero: its sparse rows are scattered below, its dense rows' coefficients go to cd
            for (int q = lane; q < nq; q += 64) wv[q] = 0.0;
            for (int dc = lane; dc < nd; dc += 64) cd[dc] = 0.0;
            MPCX_QLAP(11);
            if (nw > 0) {
                if (v.kw <= 64) { tri_forward<false>(Lp, invd, nw, t0, t1, lane); y0 = t0; tri_backward<false>(Lp, invd, nw, t0, t1, lane); }
                else { tri_forward<true>(Lp, invd, nw, t0, t1, lane); y0 = t0; y1 = t1; tri_backward<true>(Lp, invd, nw, t0, t1, lane); }
            }
            MPCX_QLAP(12);
            const double zn = snn - wave_sum((h0 ? y0 * y0 : 0.0) + (h1 ? y1 * y1 : 0.0));
            // dual ratio test: the smallest ratio, lowest slot on ties (equalities never leave)
            double tneg = -1e300; int tidx = 0x7fffffff;
            if (h0 && t0 > 1e-14 && !(k0 >= mi && k0 < m)) { tneg = -(u0 / t0); tidx = lane; }
            if (h1 && t1 > 1e-14 && !(k1 >= mi && k1 < m)) { const double tj = -(u1 / t1); if (tj > tneg) { tneg = tj; tidx = lane + 64; } }
            wave_argmax(tneg, tidx);
            const double tl = tneg > -1e300 ? -tneg : 1e300;
            const bool can_move = zn > 1e-13 * fmax(1.0, npn);
            const double t2 = can_move ? spv / zn : 1e300;
            const double tt = fmin(tl, t2);
            const int what = tt >= 1e300 ? 0 : (t2 <= tl ? 1 : 2);
            MPCX_QLAP(13);
            if (what != 0) {
                if (h0) uq[lane] = u0 - tt * t0;
                if (h1) uq[lane + 64] = u1 - tt * t1;
                if (can_move) {
                    nl_wave_sync();                                  // (the zeros above are in place)
                    if constexpr (kOneEntry) {
                        auto scatter = [&](bool h, int k, int dc, double ml) {
                            if (!h) return;
                            if (dc >= 0) cd[dc] = ml;
                            else { const int cn = sp.count(k); for (int e = 0; e < cn; ++e) atomicAdd(wv + sp.index(k, e), sp.value(k, e) * ml); }
                        };
                        scatter(h0, k0, d0, s0 * t0);
                        scatter(h1, k1, d1, s1 * t1);
                    } else {
                        const Ws W(v);
                        if (h0) { tq[lane] = s0 * t0; if (d0 >= 0) cd[d0] = s0 * t0; }
                        if (h1) { tq[lane + 64] = s1 * t1; if (d1 >= 0) cd[d1] = s1 * t1; }
                        nl_wave_sync();
                        sparse_gather(v, sp, W, nw, tq, wv, lane, 64);
                    }
                }
            }
            if (what == 1) {                                     // row nw of the factor: y and the square root of z'n (chol_append's guard)
                const bool ok = zn > 1e-13 * snn;
                const double dd = sqrt(ok ? zn : 1e-13 * snn + 1e-300);
                const int ro = nw * (nw + 1) / 2;
                if (h0) Lp[ro + lane] = y0;
                if (h1) Lp[ro + lane + 64] = y1;
                if (lane == 0) { Lp[ro + nw] = dd; invd[nw] = 1.0 / dd; uq[nw] = up + tt; wq[nw] = pidx; sgq[nw] = sgn; flag[pidx] = 1; }
            }
            if (lane == 0) { st[ST_R0] = tt; st[ST_R1] = zn; st[ST_R2] = (double)what; st[ST_R3] = (double)tidx; }
        }
        T::sync();
        MPCX_QLAP(14);
    }

    // S = N_W B^-1 N_W' of the kept rows where some of them are dense (rows through the sensitivities: config 3), on the matrix pipe:
    // the one GEMM-shaped product of the sub-problem (the reference's counterpart is the chain rule through the dense Jacobians,
    // Constraints.hpp:455-482).  V = B^-1 N_W' is formed sixteen rows at a time -- a wavefront takes row blocks rb = wave, wave + WAVES, ..
    // of B^-1 (v_mfma_f64_16x16x4_f64, B^-1 read through its packed triangle) and keeps its block of V in the accumulators; those
    // registers ARE the B operands of the second product (register r of the accumulator holds rows 4 r .. 4 r + 3 of the block in the
    // layout a k-step wants), so N_W[:, block] V[block, :] follows without V ever leaving the registers.  The wavefronts' partial sums meet
    // in the factor's storage in a fixed order (wavefront 0 stores, 1 .. add in turn): the same bits on every run.  Up to 64 kept rows.
    static constexpr int kSchurTiles = kWgWavesPerSimdOf<Mdl, WAVES, FL>::value >= 3 ? 2 : 4;    // (the accumulators of ten tiles do not fit a 128-register budget)
    static MPCX_WG_PHASE void ws_schur_mfma(int nw)
    {
        const V v; const auto &P = v.A->P; const Sp sp(v);
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nq = v.nq, ndld = v.ndld, j = lane & 15, kq = lane >> 4;
        const double *hinv = v.at(P.o_hinv), *sgq = v.at(P.o_sgq);
        const int *wq = v.iat(P.o_wq), *dcol = v.iat(P.o_dcol);
        double *Lp = v.at(P.o_L);
        typename FP::type art = art_of(v);
        const int nt = (nw + 15) >> 4, nkb = (nq + 15) >> 4;
        // this lane's kept row in each tile of sixteen: where its entries are
        int rk[kSchurTiles], rdc[kSchurTiles], rix[kSchurTiles], rcn[kSchurTiles], rro[kSchurTiles], rrl[kSchurTiles];
        const int *aoff = v.iat(P.o_aoff), *alen = v.iat(P.o_alen);
        double rsg[kSchurTiles], rv0[kSchurTiles];
#pragma unroll
        for (int ti = 0; ti < kSchurTiles; ++ti) {
            const int t = 16 * ti + j;
            const bool live = t < nw;
            const int k = live ? wq[t] : 0;
            rk[ti] = k; rdc[ti] = live ? dcol[k] : -1; rsg[ti] = live ? sgq[t] : 0.0;
            rro[ti] = rdc[ti] >= 0 ? aoff[rdc[ti]] : 0; rrl[ti] = rdc[ti] >= 0 ? alen[rdc[ti]] & 0xffff : 0;
            rcn[ti] = (live && rdc[ti] < 0) ? sp.count(k) : 0;
            rix[ti] = sp.index(k, 0); rv0[ti] = sp.value(k, 0);
        }
        // entry kk of the oriented normal of this lane's row in tile ti (zero beyond the row's or the matrix's end)
        auto nrm = [&](int ti, int kk) -> double {
            if (kk >= nq) return 0.0;
            if (rdc[ti] >= 0) return kk < rrl[ti] ? rsg[ti] * art[rro[ti] + kk] : 0.0;
            double val = (rcn[ti] > 0 && rix[ti] == kk) ? rv0[ti] : 0.0;
            for (int e = 1; e < rcn[ti]; ++e) if (sp.index(rk[ti], e) == kk) val += sp.value(rk[ti], e);
            return rsg[ti] * val;
        };
        wg_v4d S[kSchurTiles * (kSchurTiles + 1) / 2];
#pragma unroll
        for (int e = 0; e < kSchurTiles * (kSchurTiles + 1) / 2; ++e) S[e] = wg_v4d{0.0, 0.0, 0.0, 0.0};
        for (int rb = wave; rb < nkb; rb += WAVES) {
            wg_v4d Vb[kSchurTiles];
#pragma unroll
            for (int ti = 0; ti < kSchurTiles; ++ti) Vb[ti] = wg_v4d{0.0, 0.0, 0.0, 0.0};
            const int hr = 16 * rb + j;                          // this lane's row of B^-1
            for (int k0 = 0; k0 < nq; k0 += 4) {
                const int kk = k0 + kq;
                const double a = (hr < nq && kk < nq) ? hsym(hinv, hr, kk) : 0.0;
#pragma unroll
                for (int ti = 0; ti < kSchurTiles; ++ti)
                    if (ti < nt) Vb[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, nrm(ti, kk), Vb[ti], 0, 0, 0);
            }
#pragma unroll
            for (int mi = 0; mi < kSchurTiles; ++mi) {
                if (mi >= nt) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double a = nrm(mi, 16 * rb + 4 * r + kq);
#pragma unroll
                    for (int ni = 0; ni <= mi; ++ni) S[mi * (mi + 1) / 2 + ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Vb[ni][r], S[mi * (mi + 1) / 2 + ni], 0, 0, 0);
                }
            }
        }
        for (int turn = 0; turn < (WAVES < nkb ? WAVES : nkb); ++turn) {
            if (wave == turn) {
#pragma unroll
                for (int mi = 0; mi < kSchurTiles; ++mi) {
#pragma unroll
                    for (int ni = 0; ni <= mi; ++ni) {
                        if (mi >= nt) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * mi + kq + 4 * r, col = 16 * ni + j;
                            if (row < nw && col <= row) {
                                const int e = row * (row + 1) / 2 + col;
                                const double x = S[mi * (mi + 1) / 2 + ni][r];
                                Lp[e] = turn == 0 ? x : Lp[e] + x;
                            }
                        }
                    }
                }
            }
            T::sync();
        }
    }

    // One round of the warm start, by wavefront 0: the multipliers of the kept rows at the minimiser on them, u = S^-1 (N_W x0 + b) with
    // yd = art' x0 (x0 = -B^-1 gr does not change while rows are shed); the rows whose multiplier is negative as two 64-bit words in
    // st[ST_SHED] (equalities stay).  When none is: u is filed as the multipliers, and N_W' u is set up for the minimiser (sparse rows
    // scattered into wv, dense coefficients in cd), as in a dual step.
    static MPCX_WG_PHASE void ws_shed_round(int nw)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x, lane = tid & 63, mi = v.mi, m = v.m;
        if (tid < 64) {
            const Sp sp(v);
            double *Lp = v.at(P.o_L), *invd = v.at(P.o_invd), *cd = v.at(P.o_cd), *wv = v.at(P.o_wv), *uq = v.at(P.o_uq), *st = v.at(P.o_st);
            const double *yd = v.at(P.o_yd), *xq = v.at(P.o_xq), *br = v.at(P.o_br), *sgq = v.at(P.o_sgq);
            const int *dcol = v.iat(P.o_dcol), *wq = v.iat(P.o_wq);
            const int nd = P.nd, nq = v.nq;
            const bool h0 = lane < nw, h1 = lane + 64 < nw;
            const int k0 = h0 ? wq[lane] : 0, k1 = h1 ? wq[lane + 64] : 0;
            const int d0 = h0 ? dcol[k0] : -1, d1 = h1 ? dcol[k1] : -1;
            const double s0 = h0 ? sgq[lane] : 0.0, s1 = h1 ? sgq[lane + 64] : 0.0;
            double t0 = !h0 ? 0.0 : s0 * ((d0 >= 0 ? yd[d0] : sp.dot(k0, xq)) + br[k0]), t1 = !h1 ? 0.0 : s1 * ((d1 >= 0 ? yd[d1] : sp.dot(k1, xq)) + br[k1]);
            if (v.kw <= 64) { tri_forward<false>(Lp, invd, nw, t0, t1, lane); tri_backward<false>(Lp, invd, nw, t0, t1, lane); }
            else { tri_forward<true>(Lp, invd, nw, t0, t1, lane); tri_backward<true>(Lp, invd, nw, t0, t1, lane); }
            const unsigned long long b0 = __ballot(h0 && t0 < 0.0 && !(k0 >= mi && k0 < m)), b1 = __ballot(h1 && t1 < 0.0 && !(k1 >= mi && k1 < m));
            if (!(b0 | b1)) {
                if (h0) uq[lane] = t0;
                if (h1) uq[lane + 64] = t1;
                for (int q = lane; q < nq; q += 64) wv[q] = 0.0;
                for (int dc = lane; dc < nd; dc += 64) cd[dc] = 0.0;
                nl_wave_sync();
                if constexpr (kOneEntry) {
                    auto scatter = [&](bool h, int k, int dc, double ml) {
                        if (!h) return;
                        if (dc >= 0) cd[dc] = ml;
                        else { const int cn = sp.count(k); for (int e = 0; e < cn; ++e) atomicAdd(wv + sp.index(k, e), sp.value(k, e) * ml); }
                    };
                    scatter(h0, k0, d0, s0 * t0);
                    scatter(h1, k1, d1, s1 * t1);
                } else {
                    const Ws W(v);
                    double *ml = v.at(P.o_tq);
                    if (h0) { ml[lane] = s0 * t0; if (d0 >= 0) cd[d0] = s0 * t0; }
                    if (h1) { ml[lane + 64] = s1 * t1; if (d1 >= 0) cd[d1] = s1 * t1; }
                    nl_wave_sync();
                    sparse_gather(v, sp, W, nw, ml, wv, lane, 64);
                }
            }
            if (lane == 0) { unsigned long long *shw = reinterpret_cast<unsigned long long *>(st + ST_SHED); shw[0] = b0; shw[1] = b1; }
        }
        T::sync();
    }

    // ---- the working set's Schur complement kept as its INVERSE M = S^-1 (WgPlan::minv: large working sets) ----------------------------------
    // With the Cholesky factor every use of S is a pair of substitutions and every row that leaves a chain of rotations -- serial in the
    // number of working rows, on one wavefront, while the others wait (config 5: 50 to 90 rows, thirteen dual steps and thirteen rows shed
    // per iteration: half of the sub-problem's time).  With the inverse, S^-1 t is one symmetric product over the whole workgroup, a row
    // that joins is the bordering formula  M <- [M + r r'/d, -r/d; -r'/d, 1/d]  (r = M t, d = n'B^-1 n - t'r), a row that leaves the
    // rank-one correction  M <- M - m_j m_j'/M_jj  with the last row moved into its place: element-wise updates, no chain.  The price
    // is conditioning (errors grow with cond(S), not its root); the inverse is formed afresh at every warm start (a symmetric
    // Gauss-Jordan sweep of S, which also tells a dependent row by its pivot), so that an error lives for one sub-problem.
    // Packed like the factor (row r at r (r + 1) / 2), in the factor's storage.  P lanes share a row in the element-wise passes.
    // (an element-wise pass over the packed triangle: rows r and n - 1 - r together have n + 1 elements, so a group of lanes per PAIR of rows
    // gives every thread the same share -- by single rows the last ones are n times the first)
    template <class FN> static __device__ __forceinline__ void tri_rows(double *Mp, int n, int tid, FN fn)
    {
        const int np = (n + 1) >> 1;
        const int PL = np > 0 && NT / np > 0 ? NT / np : 1, groups = NT / PL, g = tid / PL, l = tid - g * PL;
        for (int p0 = 0; p0 < np; p0 += groups) {
            const int p = p0 + g;
            if (p < np && g < groups) {
                const int ra = p, rb = n - 1 - p, len = ra == rb ? ra + 1 : n + 1;
                for (int j = l; j < len; j += PL) {
                    if (j <= ra) fn(ra, j, Mp[ra * (ra + 1) / 2 + j]);
                    else fn(rb, j - ra - 1, Mp[rb * (rb + 1) / 2 + (j - ra - 1)]);
                }
            }
        }
    }
    // M <- S^-1 in place (S packed in Mp); false: a pivot vanished (dependent rows)
    static MPCX_WG_PHASE int ws_invert_m(int n)
    {
        const V v; const auto &P = v.A->P;
        return invert_packed(P.o_L, P.o_mbuf, n);
    }
    // a symmetric positive definite matrix, packed by rows of its lower triangle in LDS, replaced by its inverse (the sweep operator, pivot by
    // pivot); buf: n doubles of LDS.  0: a pivot fell below 1e-13 of the largest diagonal entry (the matrix is then garbage)
    // (the two arrays by their LDS offsets: a pointer handed through an out-of-line call is a generic one, and flat accesses wait on both memory
    // counters and take the long way to LDS -- measured on this very function: 10.7 k cycles per pivot instead of 0.7 k)
    static MPCX_WG_CALL int invert_packed(int mp_off, int buf_off, int n)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        double *Mp = v.at(mp_off), *buf = v.at(buf_off);
        double dmax = 0.0;
        for (int r = tid; r < n; r += NT) dmax = fmax(dmax, Mp[r * (r + 1) / 2 + r]);
        Red<WAVES> R(v.at(P.o_red));
        dmax = R.max(dmax);
        T::sync();
        bool ok = true;
        // Which elements of the triangle a thread owns does not depend on the pivot (rows r and n - 1 - r to a group of PL lanes, as in tri_rows):
        // where they are, and which column they belong to, is worked out ONCE, before the sweep -- the pass over the triangle is bound by the
        // instructions it issues (a wave64 instruction takes four cycles, two wavefronts share a SIMD: 800 instructions per pivot were 6.5 k
        // cycles), and most of them were this arithmetic.  Up to kOwn elements per thread in registers; a matrix too large for that many
        // (n = 121 on four wavefronts) takes the pass that recomputes them.
        constexpr int kOwn = 16;
        const int np = (n + 1) >> 1;
        const int PL = np > 0 && NT / np > 0 ? NT / np : 1, groups = NT / PL, g = tid / PL, l = tid - g * PL;
        const bool fits = np <= groups && (n + 1 + PL - 1) / PL <= kOwn;
        const bool mine = g < np && g < groups;
        const int ra = mine ? g : 0, rb = n - 1 - ra, len = !mine ? 0 : (ra == rb ? ra + 1 : n + 1);
        const int oa = ra * (ra + 1) / 2, ob = rb * (rb + 1) / 2 - ra - 1;
        int ix[kOwn], cc[kOwn];
#pragma unroll
        for (int u = 0; u < kOwn; ++u) {
            const int j = l + u * PL;
            const bool live = j < len, fa = j <= ra;
            ix[u] = live ? (fa ? oa : ob) + j : -1;
            cc[u] = (fa ? j : j - ra - 1) | (fa ? 0 : 1 << 30);      // bit 30: the element lies in row rb
        }
        for (int k = 0; k < n && ok; ++k) {                       // the sweep operator on pivot k: afterwards Mp = -(S^-1) on the swept part
            for (int t = tid; t < n; t += NT) buf[t] = hsym(Mp, t, k);
            const double d = Mp[k * (k + 1) / 2 + k];
            T::sync();
            ok = d > 1e-13 * dmax;
            const double id = 1.0 / d;
            if (fits) {
                const double ba = buf[ra] * id, bb = buf[rb] * id;      // buf[r] / d of this thread's two rows
                const bool ka = ra == k, kb = rb == k;
#pragma unroll
                for (int u0 = 0; u0 < kOwn; u0 += 4) {
                    double a[4], bc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = ix[u0 + u] >= 0 ? ix[u0 + u] : 0;
                        a[u] = Mp[e]; bc[u] = buf[cc[u0 + u] & 0xffff];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool inb = (cc[u0 + u] >> 30) != 0;
                        const int c = cc[u0 + u] & 0xffff;
                        const double br = inb ? bb : ba;
                        const bool rk = inb ? kb : ka;
                        const double x = rk ? (c == k ? -id : bc[u] * id) : (c == k ? br : fma(-br, bc[u], a[u]));
                        if (ix[u0 + u] >= 0) Mp[ix[u0 + u]] = x;
                    }
                }
            } else {
                for (int p0 = 0; p0 < np; p0 += groups) {
                    const int p = p0 + g;
                    if (p < np && g < groups) {
                        const int qa = p, qb = n - 1 - p, qlen = qa == qb ? qa + 1 : n + 1;
                        const double ba = buf[qa] * id, bb = buf[qb] * id;
                        const int pa = qa * (qa + 1) / 2, pb = qb * (qb + 1) / 2 - qa - 1;
                        for (int j0 = l; j0 < qlen; j0 += 4 * PL) {
                            double a[4], bc[4];
                            int jx[4], jc[4];
                            bool fa[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int j = j0 + u * PL < qlen ? j0 + u * PL : l;
                                fa[u] = j <= qa;
                                jc[u] = fa[u] ? j : j - qa - 1;
                                jx[u] = (fa[u] ? pa : pb) + j;
                                a[u] = Mp[jx[u]]; bc[u] = buf[jc[u]];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int r = fa[u] ? qa : qb;
                                const double br = fa[u] ? ba : bb;
                                const double x = r == k ? (jc[u] == k ? -id : bc[u] * id) : (jc[u] == k ? br : fma(-br, bc[u], a[u]));
                                if (j0 + u * PL < qlen) Mp[jx[u]] = x;
                            }
                        }
                    }
                }
            }
            T::sync();
        }
        if (!ok) return 0;
        tri_rows(Mp, n, tid, [&](int, int, double &a) { a = -a; });
        T::sync();
        return 1;
    }
    // The same inversion BLOCKED, for init_curvature's nq x nq matrix (n = 61 .. 121: the pivot-by-pivot sweep issues ~600 instructions per pivot and
    // wavefront, and a phase like that is bound by exactly that -- four cycles per wave64 instruction, the SIMD's wavefronts taking turns).  NB pivots
    // at a time (16, or 8 where the scratch is short): the sweep operator on a block K of pivots is
    //     A_KK <- -D,  D = A_KK^-1;     A_iK <- W_i = A_iK D;     A_ij <- A_ij - W_i A_jK'        (i, j outside K)
    // -- the panel product W = P D and the rank-NB update of the whole triangle as 16 x 16 tiles on the matrix pipe (v_mfma_f64_16x16x4_f64), the
    // NB x NB block inverted by wavefront 0 with the scalar sweep.  Scratch (LDS, by offset): NB^2 + 2 NB N16 doubles, N16 = n rounded up to 16.
    // Rows and pivots beyond n are identity / zero padding and are never stored.
    // (SLDS: the scratch lies in LDS at that offset; otherwise in the instance's workspace at that offset -- the same arithmetic either way: which
    // block size a controller takes depends on its size alone, never on a plan's LDS layout, so that every plan of a handle returns the same bits)
    template <int NB, bool SLDS>
    static MPCX_WG_CALL int invert_packed_blocked(int mp_off, int scratch_off, int n)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int j = lane & 15, kq = lane >> 4;
        double *Mp = v.at(mp_off);
        typedef typename BlockPtr<SLDS>::type ScrPtr;
        ScrPtr Sb = BlockPtr<SLDS>::make(SLDS ? v.at(scratch_off) : v.w + scratch_off);
        const int nt = (n + 15) >> 4, N16 = 16 * nt;
        ScrPtr Pn = Sb + NB * NB, Wn = Pn + NB * N16;
        double *flagslot = v.at(P.o_red) + 48;                  // (one word: the block's pivots were all positive)
        double dmax = 0.0;
        for (int r = tid; r < n; r += NT) dmax = fmax(dmax, Mp[r * (r + 1) / 2 + r]);
        Red<WAVES> R(v.at(P.o_red));
        dmax = R.max(dmax);
        T::sync();
        for (int k0 = 0; k0 < n; k0 += NB) {
            // the panel A[:, K] (rows beyond n: zero) and the block A_KK (pivots beyond n: identity)
            for (int e = tid; e < N16 * NB; e += NT) {
                const int t = e / NB, c = e - t * NB, k = k0 + c;
                Pn[e] = (t < n && k < n) ? hsym(Mp, t, k) : 0.0;
            }
            for (int e = tid; e < NB * NB; e += NT) {
                const int c1 = e / NB, c2 = e - c1 * NB;
                Sb[e] = (k0 + c1 < n && k0 + c2 < n) ? hsym(Mp, k0 + c1, k0 + c2) : (c1 == c2 ? 1.0 : 0.0);
            }
            if (tid == 0) flagslot[0] = 1.0;
            T::sync();
            // D = A_KK^-1 in place: the scalar sweep by wavefront 0 (afterwards Sb = -D)
            if (wave == 0) {
                bool okb = true;
                for (int c = 0; c < NB; ++c) {
                    const double d = Sb[c * NB + c];
                    double col[(NB * NB + 63) / 64], row[(NB * NB + 63) / 64], a[(NB * NB + 63) / 64];
#pragma unroll
                    for (int u = 0; u < (NB * NB + 63) / 64; ++u) {
                        const int e = lane + 64 * u, r1 = e / NB, c1 = e - r1 * NB;
                        const bool live = e < NB * NB;
                        col[u] = live ? Sb[r1 * NB + c] : 0.0; row[u] = live ? Sb[c * NB + c1] : 0.0; a[u] = live ? Sb[e] : 0.0;
                    }
                    okb = okb && (d > 1e-13 * dmax || k0 + c >= n);
                    const double id = 1.0 / d;
                    nl_wave_sync();
#pragma unroll
                    for (int u = 0; u < (NB * NB + 63) / 64; ++u) {
                        const int e = lane + 64 * u, r1 = e / NB, c1 = e - r1 * NB;
                        if (e < NB * NB) Sb[e] = r1 == c ? (c1 == c ? -id : row[u] * id) : (c1 == c ? col[u] * id : fma(-col[u] * id, row[u], a[u]));
                    }
                    nl_wave_sync();
                }
                if (lane == 0 && !okb) flagslot[0] = 0.0;
            }
            T::sync();
            if (flagslot[0] == 0.0) return 0;
            // W = P D (D = -Sb): sixteen rows of the panel to a wavefront at a time
            for (int ti = wave; ti < nt; ti += WAVES) {
                wg_v4d w = {0.0, 0.0, 0.0, 0.0};
                for (int q0 = 0; q0 < NB; q0 += 4) {
                    const double av = Pn[(16 * ti + j) * NB + q0 + kq], bv = j < NB ? -Sb[(q0 + kq) * NB + j] : 0.0;
                    w = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, w, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) if (j < NB) Wn[(16 * ti + 4 * r + kq) * NB + j] = w[r];
            }
            T::sync();
            // the whole triangle: A_tile -= W_ti P_tj' (the rows and columns of K get garbage here and their values below)
            const int ntl = nt * (nt + 1) / 2;
            for (int tn = wave; tn < ntl; tn += WAVES) {
                int ti, tj;
                tri_index(tn, ti, tj);
                wg_v4d c4;
                int ex[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + kq, col = 16 * tj + j;
                    ex[r] = (row < n && col <= row) ? row * (row + 1) / 2 + col : -1;
                    c4[r] = ex[r] >= 0 ? Mp[ex[r]] : 0.0;
                }
                for (int q0 = 0; q0 < NB; q0 += 4) {
                    const double av = -Wn[(16 * ti + j) * NB + q0 + kq], bv = Pn[(16 * tj + j) * NB + q0 + kq];
                    c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c4, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) if (ex[r] >= 0) Mp[ex[r]] = c4[r];
            }
            T::sync();
            // the rows and columns of K: A_iK = W_i, A_KK = -D
            for (int e = tid; e < n * NB; e += NT) {
                const int t = e / NB, c = e - t * NB, k = k0 + c;
                if (k >= n) continue;
                const bool tk = t >= k0 && t < k0 + NB;
                const double x = tk ? Sb[(t - k0) * NB + c] : Wn[t * NB + c];
                if (tk && t < k) continue;                      // (both in K: the pair (t, k) with t >= k writes it)
                const int r1 = t > k ? t : k, c1 = t > k ? k : t;
                Mp[r1 * (r1 + 1) / 2 + c1] = x;
            }
            T::sync();
        }
        tri_rows(Mp, n, tid, [&](int, int, double &a) { a = -a; });
        T::sync();
        return 1;
    }
    // row j leaves: M <- M - m_j m_j' / M_jj on the others, the last row takes slot j (in M and in the lists).  The vector rq = M t of the
    // warm start (the multipliers on the kept set, t fixed while rows are shed) follows: without row j it is rq_i - M_ij rq_j / M_jj -- the
    // next round needs no product (in the dual method's loop rq is recomputed anyway).
    static MPCX_WG_PHASE void ws_drop_m(int j, int nw)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x, last = nw - 1;
        double *Mp = v.at(P.o_L), *buf = v.at(P.o_mbuf), *sgq = v.at(P.o_sgq), *uq = v.at(P.o_uq), *rq = v.at(P.o_invd);
        int *wq = v.iat(P.o_wq), *flag = v.iat(P.o_flag);
        for (int t = tid; t < nw; t += NT) buf[t] = hsym(Mp, t, j);
        const double uj = rq[j];
        T::sync();
        const double id = 1.0 / buf[j];
        for (int t = tid; t < nw; t += NT) if (t != j) rq[t] = fma(-buf[t] * id, uj, rq[t]);
        // (the last row's entries go to row / column j as they are corrected: nobody reads or writes those places in this pass)
        tri_rows(Mp, nw, tid, [&](int r, int c, double &a) {
            if (r == j || c == j) return;
            const double x = fma(-buf[r] * id, buf[c], a);
            if (r == last && j != last) {
                if (c == last) Mp[j * (j + 1) / 2 + j] = x;
                else if (c < j) Mp[j * (j + 1) / 2 + c] = x;
                else Mp[c * (c + 1) / 2 + j] = x;
            } else a = x;
        });
        T::sync();
        if (tid == 0) {
            flag[wq[j]] = 0;
            if (j != last) { wq[j] = wq[last]; sgq[j] = sgq[last]; uq[j] = uq[last]; rq[j] = rq[last]; }
        }
        T::sync();
    }
    // one round of the warm start with the inverse: u = M (N_W x0 + b) (product: the first round; afterwards ws_drop_m keeps u current); the
    // rows with a negative multiplier in st[ST_SHED]; when there is none, u filed and N_W' u set up for the minimiser (as ws_shed_round)
    static MPCX_WG_PHASE void ws_shed_round_m(int nw, bool product)
    {
        const V v; const auto &P = v.A->P; const Sp sp(v);
        const int tid = threadIdx.x, lane = tid & 63, mi = v.mi, m = v.m, nq = v.nq, nd = P.nd;
        double *Mp = v.at(P.o_L), *tq = v.at(P.o_tq), *rq = v.at(P.o_invd), *cd = v.at(P.o_cd), *wv = v.at(P.o_wv), *uq = v.at(P.o_uq), *st = v.at(P.o_st);
        const double *yd = v.at(P.o_yd), *xq = v.at(P.o_xq), *br = v.at(P.o_br), *sgq = v.at(P.o_sgq);
        const int *dcol = v.iat(P.o_dcol), *wq = v.iat(P.o_wq);
        for (int q = tid; q < nq; q += NT) wv[q] = 0.0;
        for (int dc = tid; dc < nd; dc += NT) cd[dc] = 0.0;
        if (product) {
            for (int t = tid; t < nw; t += NT) {
                const int k = wq[t], dc = dcol[k];
                tq[t] = sgq[t] * ((dc >= 0 ? yd[dc] : sp.dot(k, xq)) + br[k]);
            }
            T::sync();
            hmul<NT>(Mp, tq, rq, nw, 1.0, tid);
            T::sync();
        }
        unsigned long long *shw = reinterpret_cast<unsigned long long *>(st + ST_SHED);
        if (tid < 64) {
            const bool h0 = lane < nw, h1 = lane + 64 < nw;
            const int k0 = h0 ? wq[lane] : 0, k1 = h1 ? wq[lane + 64] : 0;
            const double u0 = h0 ? rq[lane] : 0.0, u1 = h1 ? rq[lane + 64] : 0.0;
            // ONE row leaves per round here, the one with the most negative multiplier: with the inverse a round costs a product and a rank-one
            // correction, a dual step ten times that -- and of the rows that all have a negative multiplier on the full kept set most turn
            // positive again once the worst has left (config 5: 387 rows shed and 478 dual steps per solve this way, 639 and 673 when every
            // negative row leaves at once)
            double vneg = 0.0; int tsel = 0x7fffffff;
            if (h0 && u0 < 0.0 && !(k0 >= mi && k0 < m)) { vneg = -u0; tsel = lane; }
            if (h1 && u1 < 0.0 && !(k1 >= mi && k1 < m) && -u1 > vneg) { vneg = -u1; tsel = lane + 64; }
            wave_argmax(vneg, tsel);
            if (lane == 0) { shw[0] = tsel < 64 ? 1ull << tsel : 0ull; shw[1] = (tsel >= 64 && tsel < 128) ? 1ull << (tsel - 64) : 0ull; }
        }
        T::sync();
        if (!(shw[0] | shw[1])) {
            for (int t = tid; t < nw; t += NT) {
                const int k = wq[t], dc = dcol[k];
                const double u = rq[t], ml = sgq[t] * u;
                uq[t] = u;
                if (dc >= 0) cd[dc] = ml;
                else if constexpr (kOneEntry) { const int cn = sp.count(k); for (int e = 0; e < cn; ++e) atomicAdd(wv + sp.index(k, e), sp.value(k, e) * ml); }
                else tq[t] = ml;
            }
            if constexpr (!kOneEntry) { const Ws W(v); T::sync(); sparse_gather(v, sp, W, nw, tq, wv, tid, NT); }
            T::sync();
        }
    }
    // the dual part of a step with the inverse (see ws_dual_step for what it delivers in st[R0 ..]): t = N_W B^-1 n gathered, r = M t by the whole
    // workgroup, z'n = n'B^-1 n - t'r and the ratio test in one reduction, then the multipliers, N_W' r for the primal part and -- on a full
    // step -- the bordering of M and the lists
    // (gathered: normal_call has left t in tq and the zeros in wv -- the first attempt of a row where every row is a short list)
    static MPCX_WG_PHASE void ws_dual_step_m(int nw, int pidx, double sgn, double snn, double npn, double spv, double up, bool gathered)
    {
        const V v; const auto &P = v.A->P; const Sp sp(v);
        const int tid = threadIdx.x, mi = v.mi, m = v.m, nq = v.nq, nd = P.nd;
        double *Mp = v.at(P.o_L), *tq = v.at(P.o_tq), *rq = v.at(P.o_invd), *cd = v.at(P.o_cd), *wv = v.at(P.o_wv), *uq = v.at(P.o_uq),
               *sgq = v.at(P.o_sgq), *st = v.at(P.o_st);
        const double *yd = v.at(P.o_yd), *vv = v.at(P.o_vv);
        const int *dcol = v.iat(P.o_dcol);
        int *wq = v.iat(P.o_wq), *flag = v.iat(P.o_flag);
#ifdef MPCX_NL_STATS
        long long qt_ = __builtin_readcyclecounter();
#endif
        if (!gathered) {
            for (int t = tid; t < nw; t += NT) {
                const int k = wq[t], dc = dcol[k];
                tq[t] = sgq[t] * (dc >= 0 ? yd[dc] : sp.dot(k, vv));
            }
            for (int q = tid; q < nq; q += NT) wv[q] = 0.0;
            for (int dc = tid; dc < nd; dc += NT) cd[dc] = 0.0;
            T::sync();
        }
        MPCX_QLAP(11);
        if (nw > 0) { hmul<NT>(Mp, tq, rq, nw, 1.0, tid); T::sync(); }
        MPCX_QLAP(12);
        // z'n = n'B^-1 n - t'r and the dual ratio test: by wavefront 0 alone, the working set's vectors two entries a lane (a reduction over
        // the whole workgroup walked eight wavefronts through two wave reductions for the sake of the first two: 4.2 k cycles a step; the
        // test in the product's epilogue with one sum-and-argmax reduction over the workgroup: 24 k -> 40 k cycles per iteration at config 5)
        if (tid < 64) {
            const int lane = tid;
            const bool h0 = lane < nw, h1 = lane + 64 < nw;
            const double r0 = h0 ? rq[lane] : 0.0, r1 = h1 ? rq[lane + 64] : 0.0;
            const double tr = wave_sum((h0 ? tq[lane] * r0 : 0.0) + (h1 ? tq[lane + 64] * r1 : 0.0));
            double tneg = -1e300; int tidx = 0x7fffffff;
            if (h0 && r0 > 1e-14) { const int k0 = wq[lane]; if (!(k0 >= mi && k0 < m)) { tneg = -(uq[lane] / r0); tidx = lane; } }
            if (h1 && r1 > 1e-14) { const int k1 = wq[lane + 64]; if (!(k1 >= mi && k1 < m)) { const double tj = -(uq[lane + 64] / r1); if (tj > tneg) { tneg = tj; tidx = lane + 64; } } }
            wave_argmax(tneg, tidx);
            const double zn_ = snn - tr;
            const double tl_ = tneg > -1e300 ? -tneg : 1e300;
            const double t2_ = zn_ > 1e-13 * fmax(1.0, npn) ? spv / zn_ : 1e300;
            const double tt_ = fmin(tl_, t2_);
            if (lane == 0) { st[ST_R0] = tt_; st[ST_R1] = zn_; st[ST_R2] = tt_ >= 1e300 ? 0.0 : (t2_ <= tl_ ? 1.0 : 2.0); st[ST_R3] = (double)tidx; }
        }
        T::sync();
        MPCX_QLAP(13);
        const double tt = st[ST_R0], zn = st[ST_R1];
        const int what = (int)st[ST_R2];
        const bool can_move = zn > 1e-13 * fmax(1.0, npn);
        if (what != 0) {
            for (int t = tid; t < nw; t += NT) {
                const double r = rq[t];
                uq[t] -= tt * r;
                if (can_move) {
                    const int k = wq[t], dc = dcol[k];
                    const double ml = sgq[t] * r;
                    if (dc >= 0) cd[dc] = ml;
                    else if constexpr (kOneEntry) { const int cn = sp.count(k); for (int e = 0; e < cn; ++e) atomicAdd(wv + sp.index(k, e), sp.value(k, e) * ml); }
                    else tq[t] = ml;
                }
            }
            if constexpr (!kOneEntry) { if (can_move) { const Ws W(v); T::sync(); sparse_gather(v, sp, W, nw, tq, wv, tid, NT); } }
            if (what == 1) {                                     // M <- [M + r r'/d, -r/d; -r'/d, 1/d] with d = z'n (guarded as the factor's pivot is)
                const double d = zn > 1e-13 * snn ? zn : 1e-13 * snn + 1e-300, id = 1.0 / d;
                tri_rows(Mp, nw, tid, [&](int r, int c, double &a) { a = fma(rq[r] * id, rq[c], a); });
                const int ro = nw * (nw + 1) / 2;
                for (int t = tid; t < nw; t += NT) Mp[ro + t] = -rq[t] * id;
                if (tid == 0) { Mp[ro + nw] = id; uq[nw] = up + tt; wq[nw] = pidx; sgq[nw] = sgn; flag[pidx] = 1; }
            }
        }
        T::sync();
        MPCX_QLAP(14);
    }

    // warm start of the sub-problem: the rows active in the previous one (wq, sgq) as long as their multipliers stay non-negative -- the
    // minimiser on that set with u >= 0 is a valid state of the dual method.  xq holds -B^-1 gr on entry, the minimiser on the kept set
    // on return; returns the number of rows kept.
    static MPCX_WG_PHASE int ws_warm(int nw_keep)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x, lane = tid & 63;
        const int mi = v.mi, m = v.m, nq = v.nq;
        const Sp sp(v);
        const Ws W(v);
        double *hinv = v.at(P.o_hinv), *br = v.at(P.o_br), *sgq = v.at(P.o_sgq), *uq = v.at(P.o_uq), *tq = v.at(P.o_tq), *invd = v.at(P.o_invd),
               *Lp = v.at(P.o_L), *xq = v.at(P.o_xq), *np_ = v.at(P.o_np), *vv = v.at(P.o_vv), *wv = v.at(P.o_wv), *st = v.at(P.o_st);
        int *wq = v.iat(P.o_wq), *flag = v.iat(P.o_flag);
        const int *dcol = v.iat(P.o_dcol);
        Red<WAVES> R(v.at(P.o_red));
        auto is_eq = [&](int k) { return k >= mi && k < m; };
        int nw = nw_keep;
#ifdef MPCX_NL_STATS
        long long qt_ = __builtin_readcyclecounter();
#endif
        const bool minv = st[ST_MINV] != 0.0;
        // The inverse carried over from the previous sub-problem (WgPlan::carry_m: every row a short list with constant entries, so N_W has
        // not changed; the kept rows are that sub-problem's final working set, in its order).  B^-1 has changed by the BFGS update
        // [s v2] C [s v2]', C = [cc, -rho; -rho, 0], so S by U C U' with U = N_W [s v2], and by Woodbury
        //     M <- M - W (C^-1 + U'W)^-1 W',   W = M U,   C^-1 = [0, -1/rho; -1/rho, -cc/rho^2]:
        // two products with M and one element-wise pass instead of the sweep's n pivots (config 5: 88 k cycles per iteration for 60 to 80
        // rows).  Errors would add up along the iterations: after sixteen carried sub-problems, or when the 2 x 2 system is near singular, the
        // inverse is formed afresh.
        bool have_m = false;
        if (P.carry_m && minv && st[ST_CARRY] >= 1.0 && st[ST_CARRYN] < 16.0) {
            const bool upd = st[ST_CARRY] == 2.0;
            const double rho = st[ST_BFRHO], cc = st[ST_BFCC];
            const double *keep = v.w + P.w_msave, *sv = v.at(P.o_sv);
            double *u0 = tq, *u1 = np_, *w0 = invd, *w1 = wv;
            for (int e = tid; e < nw * (nw + 1) / 2; e += NT) Lp[e] = keep[e];
            if (upd)
                for (int t = tid; t < nw; t += NT) { const int k = wq[t]; u0[t] = sgq[t] * sp.dot(k, sv); u1[t] = sgq[t] * sp.dot(k, vv); }
            T::sync();
            have_m = true;
            if (upd) {
                hmul<NT>(Lp, u0, w0, nw, 1.0, tid);
                hmul<NT>(Lp, u1, w1, nw, 1.0, tid);
                T::sync();
                double g00 = 0.0, g01 = 0.0, g11 = 0.0, gsc = 0.0;
                for (int t = tid; t < nw; t += NT) { g00 += u0[t] * w0[t]; g01 += u0[t] * w1[t]; g11 += u1[t] * w1[t]; }
                const WgRed4 g = R.mix4(g00, g01, g11, gsc, 0);
                g00 = g.a; g01 = g.b - 1.0 / rho; g11 = g.c - cc / (rho * rho);
                const double det = g00 * g11 - g01 * g01, big = fmax(fabs(g00 * g11), g01 * g01);
                T::sync();
                if (fabs(det) > 1e-10 * big && big > 0.0) {
                    const double i00 = g11 / det, i01 = -g01 / det, i11 = g00 / det;
                    tri_rows(Lp, nw, tid, [&](int r, int c, double &a) {
                        a -= i00 * w0[r] * w0[c] + i01 * (w0[r] * w1[c] + w1[r] * w0[c]) + i11 * w1[r] * w1[c];
                    });
                } else have_m = false;
                T::sync();
            }
            MPCX_WG_PROBE_CARRY(have_m, nw, wq, sgq, sp, hinv, Lp, upd, st[ST_CARRYN]);
            if (tid == 0) { if (have_m) { st[ST_CARRYN] += 1.0; st[ST_NCARRY] += 1.0; st[ST_R4] = 1.0; } }
        }
        if (have_m) { T::sync(); }
        else {
        // S = N B^-1 N' of the kept rows, straight into the factor's storage
        int anyd = 0;
        for (int t = tid; t < nw; t += NT) anyd |= dcol[wq[t]] >= 0 ? 1 : 0;
        const bool any_dense = R.max((double)anyd) > 0.0;
        if (!any_dense) {
            for (int e = tid; e < nw * (nw + 1) / 2; e += NT) {
                int a, b2;
                tri_index(e, a, b2);
                const int ka = wq[a], kb = wq[b2];
                double s = 0.0;
                for (int ja = 0; ja < sp.count(ka); ++ja)
                    for (int jb = 0; jb < sp.count(kb); ++jb)
                        s = fma(sp.value(ka, ja) * sp.value(kb, jb), hsym(hinv, sp.index(ka, ja), sp.index(kb, jb)), s);
                Lp[e] = sgq[a] * sgq[b2] * s;
            }
        } else if (nw <= 16 * kSchurTiles) {
            ws_schur_mfma(nw);
        } else {
            for (int b2 = 0; b2 < nw; ++b2) {
                normal_call(wq[b2], sgq[b2], false);
                ws_n_mul(b2 + 1, P.o_vv, P.o_L + b2 * (b2 + 1) / 2);        // row b2 of S: entries 0 .. b2
            }
        }
        T::sync();
        MPCX_QLAP(1);
        if (tid == 0) st[ST_CARRYN] = 0.0;                      // (behind a barrier: everybody has read it)
        if (minv) {
            const int ok = ws_invert_m(nw);
            if (tid == 0) st[ST_R4] = ok ? 1.0 : 0.0;
        } else if (nw <= 32) {                                   // a small set: one wavefront, no workgroup barriers
            if (tid < 64) { const bool ok = chol_inplace(Lp, invd, nw, lane); if (lane == 0) st[ST_R4] = ok ? 1.0 : 0.0; }
        } else {
            const bool ok = chol_inplace_wg<WAVES>(Lp, invd, v.at(P.o_red), nw, tid);
            if (tid == 0) st[ST_R4] = ok ? 1.0 : 0.0;
        }
        T::sync();
        }
        MPCX_QLAP(2);
        if (st[ST_R4] == 0.0) {                                  // dependent rows: start cold
            for (int t = tid; t < nw; t += NT) flag[wq[t]] = 0;
            T::sync();
            return 0;
        }
        if (P.nd > 0) { art_ws_tmul(v, nw, xq, W.yd, tid); T::sync(); }       // (the kept rows' products with x0: fixed while rows are shed)
        bool first_round = true;
        while (nw > 0) {
            if (minv) ws_shed_round_m(nw, first_round); else ws_shed_round(nw);
            first_round = false;
            // every row with a negative multiplier leaves at once; equalities stay
            const unsigned long long *shw = reinterpret_cast<const unsigned long long *>(st + ST_SHED);
            unsigned long long m0 = shw[0], m1 = shw[1];
            if (!(m0 | m1)) break;
            while (m0 | m1) {                                    // from the last row down: the rows below a leaving one keep their numbers
                int t;
                if (m1) { const int bit = 63 - __builtin_clzll(m1); m1 &= ~(1ull << bit); t = 64 + bit; }
                else { const int bit = 63 - __builtin_clzll(m0); m0 &= ~(1ull << bit); t = bit; }
                if (minv) ws_drop_m(t, nw); else ws_drop(t, nw);
                --nw;
                if (tid == 0) st[ST_NSHED] += 1.0;
            }
        }
        if (nw > 0) {
            if (P.nd > 0) art_mul_call(nw);
            hmul_step_call(-1.0, false);                        // x = x0 - B^-1 N_W' u
        }
        MPCX_QLAP(3);
        return nw;
    }

    // sub-problem: min 1/2 p'Bp + gr'p  s.t.  art' p + br <= 0 (equalities: = 0)   (Goldfarb-Idnani, range-space form on B^-1)
    // returns the size of the final working set (>= 0) or a failure code (< 0)
    static MPCX_WG_PHASE int qp(int nw_keep)
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int nr = v.nr, mi = v.mi, m = v.m, mt = v.mt, nq = v.nq, KW = v.kw;
        const Sp sp(v);
        const Ws W(v);
        double *gr = v.at(P.o_gr), *hinv = v.at(P.o_hinv), *br = v.at(P.o_br), *mu = v.at(P.o_mu), *p = v.at(P.o_p),
               *sgq = v.at(P.o_sgq), *uq = v.at(P.o_uq), *tq = v.at(P.o_tq),
               *xq = v.at(P.o_xq), *np_ = v.at(P.o_np), *vv = v.at(P.o_vv), *wv = v.at(P.o_wv), *st = v.at(P.o_st);
        int *wq = v.iat(P.o_wq), *flag = v.iat(P.o_flag);
        const int *dcol = v.iat(P.o_dcol);
        Red<WAVES> R(v.at(P.o_red));
        auto is_eq = [&](int k) { return k >= mi && k < m; };

#ifdef MPCX_NL_STATS
        long long qt_ = __builtin_readcyclecounter();
#endif
        const bool minv = st[ST_MINV] != 0.0;                   // (constant over a solve attempt: set by start)
        for (int k = tid; k < mt; k += NT) flag[k] = 0;
        hmul_call(P.o_gr, P.o_xq, -1.0);                        // the unconstrained minimiser x = -B^-1 gr
        for (int t = tid; t < nw_keep; t += NT) flag[wq[t]] = 1;
        T::sync();
        MPCX_QLAP(0);
        int nw = nw_keep > 0 ? ws_warm(nw_keep) : 0;
        MPCX_TRACE("qp: kept %d of %d;", nw, nw_keep);
#ifdef MPCX_NL_STATS
        qt_ = __builtin_readcyclecounter();
#endif

        // ---- the dual method
        int fail = 0, nsteps = 0;
        bool done = false;
        for (int qit = 0; qit < 8 * (mt + nq) + 16 && !done && !fail; ++qit) {
            // the most violated row outside the working set
            const WgArgmax worst = scan_call();
            MPCX_QLAP(4);
            const double vmax = worst.v;
            const int pidx = worst.idx;
            if (mt == 0 || vmax <= 1e-12) { done = true; break; }        // primal feasible: optimal
            ++nsteps;
            if (nw >= KW) { fail = -3; if (tid == 0) st[ST_OVER] = 1.0; break; }      // working set full
            // an equality enters oriented so that it reads "n'p + b <= 0, violated"; it is never shed afterwards
            const bool p_is_eq = is_eq(pidx);
            double sgn = 1.0;
            if (p_is_eq) {
                const int dc = dcol[pidx];
                sgn = br[pidx] + (dc >= 0 ? W.yd[dc] : sp.dot(pidx, xq)) < 0.0 ? -1.0 : 1.0;
            }
            const WgSum2 nn = normal_call(pidx, sgn, true, nw);
            MPCX_QLAP(5);
            const double snn = nn.a, npn = nn.b;
            double up = 0.0, spv_ = vmax;
            bool added = false;
            for (int inner = 0; inner <= KW + 1 && !added && !fail; ++inner) {
                // t = N_W v (the new column of S); wavefront 0: rr = S^-1 t, the step length, the multipliers; then x -= t B^-1 (n - N_W' rr)
                if (minv) ws_dual_step_m(nw, pidx, sgn, snn, npn, spv_, up, inner == 0 && P.nd == 0); else ws_dual_step(nw, pidx, sgn, snn, npn, spv_, up);
                MPCX_QLAP(7);
                const double tt = st[ST_R0], zn = st[ST_R1];
                const int what = (int)st[ST_R2], kdrop = (int)st[ST_R3];
                const bool can_move = zn > 1e-13 * fmax(1.0, npn);
                if (what == 0) {
                    // no step: the row is a combination of working rows.  Violated by round-off only (a copy of an active row): set it
                    // aside; violated for real: the linearised constraints are inconsistent.
                    if (spv_ <= 1e-7 && !p_is_eq) { if (tid == 0) flag[pidx] = 2; T::sync(); added = true; break; }
                    fail = -1; break;
                }
                if (can_move) {
                    if (nw > 0) {
                        if (P.nd > 0) art_mul_call(nw);
                        MPCX_QLAP(8);
                        hmul_step_call(tt);
                        MPCX_QLAP(9);
                    } else {
                        for (int q = tid; q < nq; q += NT) xq[q] -= tt * vv[q];
                        T::sync();
                    }
                    spv_ -= tt * zn;
                }
                up += tt;
#ifdef MPCX_EMU_TRACE
                if (what == 1 && tid == 0 && blockIdx.x == 0) {
                    double r = 0.0;
                    for (int q = 0; q < nq; ++q) r += np_[q] * xq[q];
                    r += sgn * br[pidx];
                    if (fabs(r) > 1e-11) fprintf(stderr, "  [row %d joined with residual %.3e (violation before %.3e, nw %d, zn %.3e snn %.3e)]\n", pidx, r, vmax, nw, zn, snn);
                }
#endif
                if (what == 1) { ++nw; added = true; }                  // full step: the row has joined the working set
                else { if (minv) ws_drop_m(kdrop, nw); else ws_drop(kdrop, nw); --nw; }       // a multiplier hit zero: that row leaves, try again
            }
            if (!fail && !added) fail = -1;
            MPCX_QLAP(10);
        }
        if (!fail && !done) fail = -1;
        if (!fail && minv && nw > 0) {
            // The inverse form is trusted only as far as it can be checked.  Whatever r = M t was, the steps keep B x + g + N_W' u = 0 (x and u move
            // together) and u >= 0, and the scan has found every row outside the working set satisfied: x is the sub-problem's solution exactly when
            // the WORKING rows hold with equality -- which they do as far as M is S^-1 (a step keeps them at zero through t - S r = 0).  Their
            // residuals at x measure the inverse's error; beyond 1e-9 the sub-problem counts as failed (-5) and the attempt loop takes the
            // instance again with the factor.  (Found with tight input bounds on six oscillators: 2 of 256 instances "converged" at points
            // whose cost was 2e-4 and 5e-4 above the optimum, status SUCCESS.)
            ws_n_mul(nw, P.o_xq, P.o_tq);
            double rmax = 0.0;
            for (int t = tid; t < nw; t += NT) rmax = fmax(rmax, fabs(tq[t] + sgq[t] * br[wq[t]]));
            rmax = R.max(rmax);
            T::sync();
            if (rmax > 1e-9) fail = -5;
        }
        MPCX_TRACE(" %d dual steps, %d rows at the end, fail %d\n", nsteps, nw, fail);
        if (tid == 0) { st[ST_R5] += (double)nsteps; st[ST_R5 + 1] = fmax(st[ST_R5 + 1], (double)nw); st[ST_QNW] = (double)nw; }
        if (fail) { T::sync(); return fail; }
        if (P.needs_phi) {                                       // (the LDS copy of the multipliers: what merit's chain for the dynamics multipliers reads)
            for (int k = tid; k < mt; k += NT) mu[k] = 0.0;
            T::sync();
            for (int t = tid; t < nw; t += NT) mu[wq[t]] = sgq[t] * uq[t];
        }
        if (tid == 0 && nq < nr) xq[nq] = 0.0;                  // (p is xq: without a slack variable its last entry stays zero)
        if (P.carry_m && minv) {
            // the inverse of this working set's Schur complement is the next sub-problem's, up to the rank-two change of B^-1 in between
            // (the factor's storage is an overlay: the evaluation phases write over it)
            double *keep = v.w + P.w_msave;
            const double *Mp = v.at(P.o_L);
            for (int e = tid; e < nw * (nw + 1) / 2; e += NT) keep[e] = Mp[e];
            if (tid == 0) st[ST_CARRY] = nw > 0 ? 1.0 : 0.0;
        }
        T::sync();
        MPCX_QLAP(10);
        return nw;
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // the full-space step d = [dx ; p] (dx by one forward chain with p applied) and the numbers of the convergence test
    // st[R0..R3] = max |d|, max |defect|, g'd, max |z|
    static MPCX_WG_PHASE void step()
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ch = v.ch, nz = v.nz, nxs = v.nxs, nr = v.nr, mi = v.mi, m = v.m;
        double *p = v.at(P.o_p), *dx = v.at(P.o_dx), *c = v.at(P.o_c), *z = v.at(P.o_z), *gu = v.at(P.o_gu), *gin = v.at(P.o_gin), *st = v.at(P.o_st);
        typename FP::type F = FP::get(v);
        gwp gxg = (gwp)(v.w + P.w_gx);
        for (int k = tid; k < nxs; k += NT) {
            const int i = k / NX;
            const double *pb = p + min(i, ch - 1) * NU;
            double s = F[(size_t)k * FW + FW - 1];
#pragma unroll
            for (int j = 0; j < NU; ++j) s = fma(F[(size_t)k * FW + NX + j], pb[j], s);
            dx[k] = s;
        }
        T::sync();
        chain<false>(v, dx, tid);
        T::sync();
        Red<WAVES> R(v.at(P.o_red));
        double dmax = 0, cmax = 0, gd = 0, zmax = 0;
        for (int k = tid; k < nxs; k += NT) { dmax = fmax(dmax, fabs(dx[k])); gd += gxg[k] * dx[k]; cmax = fmax(cmax, fabs(c[k])); }
        for (int q = tid; q < nr; q += NT) { dmax = fmax(dmax, fabs(p[q])); gd += gu[q] * p[q]; }
        for (int k = tid; k < nz; k += NT) zmax = fmax(zmax, fabs(z[k]));
        for (int k = mi + tid; k < m; k += NT) cmax = fmax(cmax, fabs(gin[k]));       // user equalities count as defects
        const WgRed4 r4 = R.mix4(dmax, cmax, gd, zmax, 0xB);
        if (tid == 0) { st[ST_R0] = r4.a; st[ST_R1] = r4.b; st[ST_R2] = r4.c; st[ST_R3] = r4.d; }
        T::sync();
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // after the sub-problem: the BFGS memory, the largest multiplier (st[R0]) and the l1 violation at z (st[R1])
    static MPCX_WG_PHASE void merit(int nw)
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int nxs = v.nxs, mi = v.mi, m = v.m, nq = v.nq;
        const Sp sp(v);
        const Ws W(v);
        double *gr = v.at(P.o_gr), *glold = v.at(P.o_glold), *uq = v.at(P.o_uq), *lam = v.at(P.o_lam), *jx = v.at(P.o_jx), *c = v.at(P.o_c),
               *gin = v.at(P.o_gin), *st = v.at(P.o_st), *sgq = v.at(P.o_sgq);
        const int *wq = v.iat(P.o_wq), *jxoff = v.iat(P.o_jxoff);
        const unsigned long long *xmask = reinterpret_cast<const unsigned long long *>(v.at(P.o_xmask));
        Red<WAVES> R(v.at(P.o_red));
        // reduced Lagrangian gradient at this point with the new multipliers: the BFGS memory
        ws_nt_mul(nw, P.o_uq, P.o_glold);
        for (int q = tid; q < nq; q += NT) glold[q] += gr[q];
        double lam_max;
        if (P.needs_phi) {
            // multipliers of the dynamics equalities: Jx' lam = -(g_x + Jin_x' mu), a backward chain over the blocks
            gwp gxg = (gwp)(v.w + P.w_gx);
            const int *bnd_idx = v.iat(P.o_bidx);
            const double *bnd_sign = v.at(P.o_bsign);
            const double *mu = v.at(P.o_mu);                                // (the sub-problem left mu[k] = orientation * multiplier, zero off the working set)
            const int *sbf = v.iat(P.o_sbf);
            const int ph = v.ph;
            for (int row = tid; row < nxs; row += NT) {
                const int i = row / NX, a = row - i * NX;                   // entry a of state row i + 1
                double s2 = gxg[row];
                auto add = [&](int k) {
                    if (((xmask[k] >> i) & 1ull) && mu[k] != 0.0) {
                        const int sl = jxoff[k] + __builtin_popcountll(xmask[k] & ((1ull << i) - 1ull));
                        s2 = fma(jx[sl * NX + a], mu[k], s2);
                    }
                };
                int first, count;
                Mdl::ineq_rows_of_x(i + 1, first, count);
                for (int k = first; k < first + count; ++k) add(k);
                for (int k = mi; k < m; ++k) add(k);
                for (int kb = sbf[i]; kb < sbf[i + 1]; ++kb) if (bnd_idx[kb] == row) s2 = fma(bnd_sign[kb], mu[m + kb], s2);
                lam[row] = s2;
            }
            (void)ph;
            T::sync();
            chain<true>(v, lam, tid);
            T::sync();
            lam_max = defect_multiplier_max(v, lam, tid);
        } else {
            lam_max = tid == 0 ? st[ST_LAMDYN] : 0.0;
        }
        for (int t = tid; t < nw; t += NT) lam_max = fmax(lam_max, fabs(uq[t]));
        double viol = 0;
        for (int k = tid; k < nxs; k += NT) viol += fabs(c[k]);
        for (int k = tid; k < m; k += NT) viol += k < mi ? fmax(gin[k], 0.0) : fabs(gin[k]);
        const WgRed4 r4 = R.mix4(lam_max, viol, 0.0, 0.0, 0x1);
        if (tid == 0) { st[ST_R0] = r4.a; st[ST_R1] = r4.b; }
        T::sync();
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // line search on the l1 merit function: eight step lengths a = 2^-(g + 8 round) at a time, NT / 8 lanes each.
    // The three parts of a trial point's merit value, each this lane's share:
    static __device__ __attribute__((noinline)) double ls_user_rows(double al, int part, int stride)
    {
        const V v; const auto &P = v.A->P;
        const int ph = v.ph, mi = v.mi, m = v.m;
        const double *prm = v.at(P.o_prm);
        const double *z = v.at(P.o_z), *p = v.at(P.o_p);
        const Lin XL{v.at(P.o_Xs), v.at(P.o_dXs), NX, al}, UL{v.at(P.o_Us), v.at(P.o_dUs), NU, al};
        const double et = z[v.nz - 1] + al * p[v.nzu];
        double vio = 0.0;
        for (int k = part; k < mi; k += stride) vio += fmax(Mdl::ineq(k, XL, UL, et, ph, prm), 0.0);
        for (int k = part; k < m - mi; k += stride) vio += fabs(Mdl::eq(k, XL, UL, ph, prm));
        return vio;
    }
    // (this lane's share of the trial point's cost: all of it on lane 0 of the group, or -- a cost that is a sum over the horizon's rows -- the
    // rows part, part + stride, ..)
    static __device__ __attribute__((noinline)) double ls_cost(double al, int part, int stride)
    {
        const V v; const auto &P = v.A->P;
        const double *z = v.at(P.o_z), *p = v.at(P.o_p);
        const Lin XL{v.at(P.o_Xs), v.at(P.o_dXs), NX, al}, UL{v.at(P.o_Us), v.at(P.o_dUs), NU, al};
        const double et = z[v.nz - 1] + al * p[v.nzu];
        if constexpr (Mdl::COST_STAGEWISE) {
            double s = part == 0 ? Mdl::slack_cost(et, v.at(P.o_prm)) : 0.0;
            for (int i = part; i <= v.ph; i += stride) s += Mdl::stage(i, XL, UL, v.ph, v.at(P.o_prm));
            return s;
        } else {
            const double c = Mdl::cost(XL, UL, et, v.ph, v.at(P.o_prm));     // (every lane evaluates it: no divergence around the call)
            return part == 0 ? c : 0.0;
        }
    }
    static __device__ __attribute__((noinline)) double ls_defects(double al, int part, int stride)
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int ph = v.ph;
        const double *prm = v.at(P.o_prm);
        const Scale sc = v.scale();
        const Lin XL{v.at(P.o_Xs), v.at(P.o_dXs), NX, al}, UL{v.at(P.o_Us), v.at(P.o_dUs), NU, al};
        const double h = 0.5 * M.Ts;
        double vio = 0.0;
        for (int i = part; i < ph; i += stride) {
            double xk[NX], xk1[NX], uk[NU], fa[NX], s = 0;
            for (int a = 0; a < NX; ++a) { xk[a] = XL(i, a); xk1[a] = XL(i + 1, a); }
            for (int a = 0; a < NU; ++a) uk[a] = UL(i, a);
            Mdl::f(fa, xk, uk, prm);
            if (CT) {
                double fb[NX];
                Mdl::f(fb, xk1, uk, prm);
                for (int a = 0; a < NX; ++a) s += fabs(sc.over_ss(xk[a] + (h * (fa[a] + fb[a])) - xk1[a], a));
            } else {
                for (int a = 0; a < NX; ++a) s += fabs(sc.over_ss(xk1[a] - fa[a], a));
            }
            vio += s;
        }
        return vio;
    }
    // returns the accepted length, or -1 if none down to 2^-40
    static MPCX_WG_PHASE double linesearch(double nu_pen, double phi0, double dphi)
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ph = v.ph, ch = v.ch, nxs = v.nxs;
        const Scale sc = v.scale();
        const double *x0 = v.x0();
        double *z = v.at(P.o_z), *Xs = v.at(P.o_Xs), *Us = v.at(P.o_Us), *dXs = v.at(P.o_dXs), *dUs = v.at(P.o_dUs), *dx = v.at(P.o_dx),
               *p = v.at(P.o_p), *st = v.at(P.o_st);
        for (int k = tid; k < (ph + 1) * NX; k += NT) {
            const int i = k / NX, j = k - i * NX;
            Xs[k] = sc.over_ss(i == 0 ? x0[j] : z[(i - 1) * NX + j], j);
            dXs[k] = i == 0 ? 0.0 : sc.over_ss(dx[k - NX], j);
        }
        for (int k = tid; k < (ph + 1) * NU; k += NT) {
            const int i = k / NU, j = k - i * NU, q = min(min(i, ph - 1), ch - 1) * NU + j;
            Us[k] = sc.by_su(z[nxs + q], j);
            dUs[k] = sc.by_su(p[q], j);
        }
        T::sync();
        constexpr int GS = NT / kNlTrials;                       // lanes per trial point: 8, 16, 32 or 64
        const int grp = tid / GS, part = tid % GS;
        double a_step = -1.0;
        for (int round = 0; round < 5 && a_step < 0.0; ++round) {
            const double al = ldexp(1.0, -(grp + 8 * round));
            const double cst = ls_cost(al, part, GS);
            const double vio = ls_user_rows(al, part, GS) + ls_defects(al, part, GS);
            double mer = cst + nu_pen * vio;
            if constexpr (GS == 8) mer = group_sum<8>(mer);
            else if constexpr (GS == 16) mer = group_sum<16>(mer);
            else if constexpr (GS == 32) { mer = group_sum<16>(mer); mer += __shfl_xor(mer, 16); }
            else mer = wave_sum(mer);
            if (part == 0) st[ST_ACC + grp] = mer <= phi0 + 1e-4 * al * dphi ? 1.0 : 0.0;
            T::sync();
            for (int g = kNlTrials - 1; g >= 0; --g) if (st[ST_ACC + g] != 0.0) a_step = ldexp(1.0, -(g + 8 * round));
            T::sync();
        }
        return a_step;
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // z += a d; s = a p for the next BFGS update; the step's norms for nlopt's stopping rules: st[R0..R2] = |step|_1, |z|_1, max |step|
    static MPCX_WG_PHASE void update(double a_step)
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int nxs = v.nxs, nr = v.nr;
        double *z = v.at(P.o_z), *dx = v.at(P.o_dx), *p = v.at(P.o_p), *sv = v.at(P.o_sv), *st = v.at(P.o_st);
        Red<WAVES> R(v.at(P.o_red));
        double s1 = 0, z1 = 0, smax = 0;
        const bool bounded = M.nbnd > 0;
        for (int q = tid; q < nr; q += NT) sv[q] = a_step * p[q];
        for (int k = tid; k < nxs + nr; k += NT) {
            const double dk = a_step * (k < nxs ? dx[k] : p[k - nxs]);
            double zn = z[k] + dk;
            // (bounds on the decision vector are kept exactly, as nlopt's SLSQP keeps them -- its iterates are clamped to [lb, ub]; the
            // sub-problem's bound rows hold to round-off of the step, 1e-8 after a few partial steps)
            if (bounded) zn = fmin(fmax(zn, M.zlb[k]), M.zub[k]);
            z[k] = zn;
            s1 += fabs(dk); z1 += fabs(zn); smax = fmax(smax, fabs(dk));
        }
        const WgRed4 r4 = R.mix4(s1, z1, smax, 0.0, 0x4);
        if (tid == 0) { st[ST_R0] = r4.a; st[ST_R1] = r4.b; st[ST_R2] = r4.c; }
        T::sync();
    }

    // largest violation at z, for nlopt's stopping rules (applied to a step that ended at a feasible point)
    static MPCX_WG_PHASE double violation()
    {
        const V v; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int nxs = v.nxs, mi = v.mi, m = v.m;
        const double *c = v.at(P.o_c), *gin = v.at(P.o_gin);
        Red<WAVES> R(v.at(P.o_red));
        double vmax = 0;
        for (int k = tid; k < nxs; k += NT) vmax = fmax(vmax, fabs(c[k]));
        for (int k = tid; k < m; k += NT) vmax = fmax(vmax, k < mi ? gin[k] : fabs(gin[k]));
        vmax = R.max(vmax);
        T::sync();
        return vmax;
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // The curvature estimate's START: the condensed Gauss-Newton Hessian of the cost at the first iterate, inverted.
    //     B0 = Phi' Qx Phi + Ru,   Phi_{i+1} = Abar_i Phi_i + Bbar_i E_b(i)      (the sensitivities of the states to the blocked inputs),
    // Qx_r / Ru_i the second derivatives of stage r's share of the cost in its own row of X / U (central second differences of Mdl::stage --
    // exact for the quadratic costs of the reference's examples), the slack's entry from Mdl::slack_cost.  NLopt's SLSQP starts its BFGS matrix
    // from the identity, and so did this kernel: a problem in 60 (120) reduced variables then spends 60 (120) iterations LEARNING a matrix that
    // can be written down -- config 3 took 78 SQP iterations per solve, config 5 48.  What is left for the BFGS updates is the curvature of the
    // dynamics and of the constraints.  The optimum is the same (the iteration's fixed points do not depend on B), the route to it is shorter.
    // Phi_i' (Qx Phi_i) is the one GEMM-shaped product of the non-linear path ("the condensed Hessian" of BASELINE's config 5): sixteen-row tiles
    // of B0 accumulate over the horizon in the f64 MFMA accumulators (v_mfma_f64_16x16x4_f64), each wavefront its own tiles.
    // Runs once per solve, before the first iteration, with Xs / Us / the folded blocks of the first evaluation in place; scratch: the overlay
    // behind Xs and Us, and the sub-problem's four vectors.  Not positive definite (a cost that is not convex): the identity, as before.
#ifdef MPCX_NL_STATS
#define MPCX_CLAP(k) do { const long long now_ = __builtin_readcyclecounter(); ct_[k] += now_ - cl_; cl_ = now_; } while (0)
#else
#define MPCX_CLAP(k) do { } while (0)
#endif
    struct Pert2 {                                            // two perturbed entries of one row
        const double *M; int n, row, c1, c2; double d1, d2;
        __device__ __forceinline__ double operator()(int i, int j) const
        {
            double v = M[i * n + j];
            if (i == row) { if (j == c1) v += d1; if (j == c2) v += d2; }
            return v;
        }
    };
    template <bool ISX> static __device__ __forceinline__ double stage_d2(const V &v, int r, int a, int b2)
    {
        const auto &P = v.A->P;
        const double *Xs = v.at(P.o_Xs), *Us = v.at(P.o_Us), *prm = v.at(P.o_prm);
        const double *Mx = ISX ? Xs : Us;
        constexpr int N = ISX ? NX : NU;
        const double ha = 1e-4 * fmax(1.0, fabs(Mx[r * N + a])), hb = 1e-4 * fmax(1.0, fabs(Mx[r * N + b2]));
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double da = (s & 1) ? -ha : ha, db = (s & 2) ? -hb : hb;
            const Pert2 Pp{Mx, N, r, a, b2, da, db};
            const Pert O{ISX ? Us : Xs, ISX ? NU : NX, -1, -1, -1, 0.0};
            double f;
            if constexpr (ISX) f = Mdl::stage(r, Pp, O, v.ph, prm); else f = Mdl::stage(r, O, Pp, v.ph, prm);
            acc += ((s == 0 || s == 3) ? f : -f);
        }
        return acc / (4.0 * ha * hb);
    }
    // the horizon of init_curvature.  PHI_LDS: the two sensitivity buffers lie in the overlay behind Xs / Us; otherwise in the instance's workspace (a
    // plan that cut the working set's capacity has cut the overlay with it) -- the pointer typed either way, so that no access is a flat one
    template <bool PHI_LDS>
    static __device__ __forceinline__ void curv_horizon(const V &v)
    {
        const auto &P = v.A->P;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int ph = v.ph, ch = v.ch, nzu = v.nzu;
        double *hinv = v.at(P.o_hinv), *Qb = v.at(P.o_xq);
        typedef typename BlockPtr<PHI_LDS>::type PhiPtr;
        PhiPtr phiA = BlockPtr<PHI_LDS>::make(PHI_LDS ? v.at(P.o_Us) + (((ph + 1) * NU + 1) & ~1) : v.w + P.w_phi), phiB = phiA + NX * nzu;
        typename FP::type F = FP::get(v);
#ifdef MPCX_NL_STATS
        long long ct_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cl_ = __builtin_readcyclecounter();
#endif
        for (int e = tid; e < 2 * NX * nzu; e += NT) phiA[e] = 0.0;
        T::sync();
        // the horizon: Phi one step on, T = Qx Phi, B0 += Phi' T -- the last as 16 x 16 tiles in the MFMA accumulators (tile (tp, tq), tq <= tp, is
        // tile number tp (tp + 1) / 2 + tq; wavefront w owns tiles w, w + WAVES, ..: up to kCurvTiles of them)
        const int nt = (nzu + 15) >> 4, ntiles = nt * (nt + 1) / 2;
        constexpr int kCurvTiles = 5;                           // 8 x 8 tiles' lower triangle over eight wavefronts; fewer wavefronts: several passes over the horizon
        const int j = lane & 15, kq = lane >> 4;
        for (int t0 = 0; t0 < ntiles; t0 += WAVES * kCurvTiles) {
            wg_v4d acc[kCurvTiles];
            int tp[kCurvTiles], tqq[kCurvTiles];
#pragma unroll
            for (int u = 0; u < kCurvTiles; ++u) {
                acc[u] = wg_v4d{0.0, 0.0, 0.0, 0.0};
                const int tn = t0 + wave + u * WAVES;
                int r, c;
                tri_index(tn < ntiles ? tn : 0, r, c);
                tp[u] = tn < ntiles ? r : -1; tqq[u] = c;
            }
            if (t0 > 0) { for (int e = tid; e < 2 * NX * nzu; e += NT) phiA[e] = 0.0; T::sync(); }
            PhiPtr cur = phiA, nxt = phiB;
            for (int i = 0; i < ph; ++i) {
                const int bi = min(i, ch - 1), ncol = (bi + 1) * NU;        // the columns that are not zero yet
                // Phi_{i+1} = Abar_i Phi_i + Bbar_i E_bi, sixteen columns to a wavefront at a time on the matrix pipe (A operand: row j of Abar_i, B operand:
                // Phi_i's rows); the second differences of stage i + 1 in its row of X by the lanes next to it
                const int nct = (ncol + 15) >> 4;
                for (int t = wave; t < nct; t += WAVES) {
                    const int q = 16 * t + j;
                    wg_v4d pa = {0.0, 0.0, 0.0, 0.0};
                    for (int k0 = 0; k0 < NX; k0 += 4) {
                        const int k = k0 + kq;
                        const double av = (j < NX && k < NX) ? F[(size_t)(i * NX + j) * FW + k] : 0.0, bv = (k < NX && q < ncol) ? cur[k * nzu + q] : 0.0;
                        pa = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, pa, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = 4 * r + kq;
                        if (a < NX && q < ncol) nxt[a * nzu + q] = pa[r] + (q >= bi * NU ? F[(size_t)(i * NX + a) * FW + NX + (q - bi * NU)] : 0.0);
                    }
                }
                MPCX_CLAP(1);
                for (int e = tid; e < NX * NX; e += NT) { const int a = e / NX, b2 = e - a * NX; Qb[e] = stage_d2<true>(v, i + 1, max(a, b2), min(a, b2)); }
                T::sync();
                MPCX_CLAP(2);
                // B0 tile (tp, tq) += Phi[:, 16 tp ..]' (Qx Phi[:, 16 tq ..]): T = Qx Phi's tile first -- its accumulator registers ARE the B operands of the
                // second product (register r holds rows 4 r .. 4 r + 3 in the layout a k-step wants), so T never leaves the registers
#pragma unroll
                for (int u = 0; u < kCurvTiles; ++u) {
                    if (tp[u] < 0 || 16 * tqq[u] >= ncol || 16 * tp[u] >= ncol) continue;
                    const int pc = 16 * tp[u] + j, qc = 16 * tqq[u] + j;
                    wg_v4d ta = {0.0, 0.0, 0.0, 0.0};
                    for (int k0 = 0; k0 < NX; k0 += 4) {
                        const int k = k0 + kq;
                        const double av = (j < NX && k < NX) ? Qb[j * NX + k] : 0.0, bv = (k < NX && qc < ncol) ? nxt[k * nzu + qc] : 0.0;
                        ta = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, ta, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (4 * r >= NX) continue;
                        const int k = 4 * r + kq;
                        const double av = (k < NX && pc < ncol) ? nxt[k * nzu + pc] : 0.0;
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, ta[r], acc[u], 0, 0, 0);
                    }
                }
                T::sync();
                MPCX_CLAP(4);
                PhiPtr sw = cur; cur = nxt; nxt = sw;          // (Phi_{i+1} is the next step's Phi_i)
            }
            // the tiles into the packed matrix (element (4 r + kq, j) of a tile in register r): every entry has one owner
#pragma unroll
            for (int u = 0; u < kCurvTiles; ++u) {
                if (tp[u] < 0) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = 16 * tp[u] + 4 * r + kq, q = 16 * tqq[u] + j;
                    if (p < nzu && q <= p) hinv[p * (p + 1) / 2 + q] += acc[u][r];
                }
            }
            T::sync();
        }
#ifdef MPCX_NL_STATS
        if (tid == 0) for (int k = 0; k < 5; ++k) (v.w + P.w_scal)[32 + k] += (double)ct_[k];
#endif
    }
    static MPCX_WG_PHASE void init_curvature()
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ph = v.ph, ch = v.ch, nzu = v.nzu, nq = v.nq;
        const Scale sc = v.scale();
        double *hinv = v.at(P.o_hinv), *st = v.at(P.o_st);
        const int nh = nq * (nq + 1) / 2;
#ifdef MPCX_NL_STATS
        long long ct_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cl_ = __builtin_readcyclecounter();      // cycles of this phase's parts: scal[32 ..] (tools/nlmpc_phases.py)
#endif
        for (int e = tid; e < nh; e += NT) hinv[e] = 0.0;
        T::sync();
        // the inputs' own curvature: row i of U (row ph is the copy of row ph - 1) belongs to block min(i, ph - 1, ch - 1); one thread per entry of a
        // block's NU x NU matrix, the rows of one block added up in their order
        for (int e = tid; e < ch * NU * NU; e += NT) {
            const int bq = e / (NU * NU), jj = e - bq * NU * NU, j1 = jj / NU, j2 = jj - j1 * NU;
            if (j2 > j1) continue;
            double acc = 0.0;
            for (int i = 0; i <= ph; ++i) if (min(min(i, ph - 1), ch - 1) == bq) acc += stage_d2<false>(v, i, j1, j2);
            const int p = bq * NU + j1, q = bq * NU + j2;
            hinv[p * (p + 1) / 2 + q] = sc.by_su(sc.by_su(acc, j1), j2);
        }
        if (nq > nzu && tid == 0) {                             // the slack (soft constraints): its own second difference
            const double e0 = v.at(P.o_z)[v.nz - 1], he = 1e-4 * fmax(1.0, fabs(e0));
            const double *prm = v.at(P.o_prm);
            hinv[nzu * (nzu + 1) / 2 + nzu] = (Mdl::slack_cost(e0 + he, prm) - 2.0 * Mdl::slack_cost(e0, prm) + Mdl::slack_cost(e0 - he, prm)) / (he * he);
        }
        MPCX_CLAP(0);
#ifdef MPCX_NL_STATS
        if (tid == 0) { for (int k = 0; k < 5; ++k) (v.w + P.w_scal)[32 + k] = 0.0; (v.w + P.w_scal)[32] = (double)ct_[0]; }
        T::sync();
#endif
        if (P.curv_lds) curv_horizon<true>(v); else curv_horizon<false>(v);
#ifdef MPCX_NL_STATS
        cl_ = __builtin_readcyclecounter();
#endif
        MPCX_CLAP(5);
        // (blocked, on the matrix pipe, where the overlay behind Xs / Us holds the panel buffers: the plan says with which block size)
        const int so = P.inv_lds ? P.o_Us + (((ph + 1) * NU + 1) & ~1) : P.w_phi;
        const int ok = P.inv_nb == 16 ? (P.inv_lds ? invert_packed_blocked<16, true>(P.o_hinv, so, nq) : invert_packed_blocked<16, false>(P.o_hinv, so, nq))
                     : P.inv_nb == 8 ? (P.inv_lds ? invert_packed_blocked<8, true>(P.o_hinv, so, nq) : invert_packed_blocked<8, false>(P.o_hinv, so, nq))
                                     : invert_packed(P.o_hinv, P.o_np, nq);
        MPCX_CLAP(6);
#ifdef MPCX_NL_STATS
        if (tid == 0) for (int k = 5; k < 8; ++k) (v.w + P.w_scal)[32 + k] = (double)ct_[k];
#endif
        if (tid == 0) st[ST_CARRY] = 0.0;                       // (a saved inverse of the Schur complement belongs to the estimate that is gone)
        if (!ok) { for (int e = tid; e < nh; e += NT) { int r, c; tri_index(e, r, c); hinv[e] = r == c ? 1.0 : 0.0; } }
        (void)M; (void)st;
        T::sync();
    }

    // from here on the sub-problems of this attempt keep the working set's factor (see the kernel's loop)
    static MPCX_WG_PHASE void leave_inverse()
    {
        const V v;
        double *st = v.at(v.A->P.o_st);
        T::sync();                                               // (everybody has read the flags the failed sub-problem left)
        if (threadIdx.x == 0) { st[ST_MINV] = 0.0; st[ST_SWITCHED] = 1.0; st[ST_CARRY] = 0.0; st[ST_OVER] = 0.0; }
        T::sync();
    }
    static MPCX_WG_PHASE void reset_hessian()
    {
        const V v; const auto &P = v.A->P;
        double *hinv = v.at(P.o_hinv);
        const int nr = v.nr;
        for (int e = threadIdx.x; e < nr * (nr + 1) / 2; e += NT) { int r, c; tri_index(e, r, c); hinv[e] = r == c ? 1.0 : 0.0; }
        if (threadIdx.x == 0) v.at(P.o_st)[ST_CARRY] = 0.0;       // (a saved inverse of the Schur complement belongs to the estimate that is gone)
        T::sync();
    }
    static MPCX_WG_PHASE void take_last_step()
    {
        const V v; const auto &M = v.A->M; const auto &P = v.A->P;
        const int nxs = v.nxs, nr = v.nr;
        double *z = v.at(P.o_z), *dx = v.at(P.o_dx), *p = v.at(P.o_p);
        const bool bounded = M.nbnd > 0;
        for (int k = threadIdx.x; k < nxs + nr; k += NT) {
            double zn = z[k] + (k < nxs ? dx[k] : p[k - nxs]);
            if (bounded) zn = fmin(fmax(zn, M.zlb[k]), M.zub[k]);      // (as in update)
            z[k] = zn;
        }
        T::sync();
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // results (NLOptimizer.hpp:536-624): cmd = U.row(0), cost, status map, feasibility of the user constraints; Xs / Us hold the
    // trajectory of the last evaluation
    static MPCX_WG_PHASE void finish(int code, int it)
    {
        const V v; const auto &S = v.A->S; const auto &P = v.A->P;
        const int tid = threadIdx.x;
        const int ph = v.ph, nz = v.nz, nr = v.nr, mi = v.mi, m = v.m, mt = v.mt, b = v.b;
        const double *z = v.at(P.o_z), *Xs = v.at(P.o_Xs), *Us = v.at(P.o_Us), *gin = v.at(P.o_gin), *mu = v.at(P.o_mu), *hinv = v.at(P.o_hinv),
                     *st = v.at(P.o_st);
        const double *u0 = v.u0(), *prm = v.at(P.o_prm);
        Red<WAVES> R(v.at(P.o_red));
        double gmax = -1e300, hmax = 0.0;
        for (int k = tid; k < m; k += NT) { if (k < mi) gmax = fmax(gmax, gin[k]); else hmax = fmax(hmax, fabs(gin[k])); }
        gmax = R.max(gmax); hmax = R.max(hmax);
        const bool failed = code < 0;
        double *o_cmd = S.cmd, *o_z = S.z_out, *o_mu = S.mu_out, *o_sx = S.seq_state, *o_su = S.seq_input, *o_sy = S.seq_output;
        if (o_cmd) for (int j = tid; j < NU; j += NT) o_cmd[(size_t)b * NU + j] = failed ? u0[j] : Us[j];
        // (an instance that outgrew a cut capacity is taken again by the second launch, from the same z_warm and curvature estimate: neither is
        // overwritten here -- z_out may be the caller's z_warm, the estimate's place is the same in both plans)
        const bool again = P.cut && st[ST_OVER] != 0.0;
        if (o_z && !again) for (int k = tid; k < nz; k += NT) o_z[(size_t)b * nz + k] = z[k];
        // the multipliers of the last sub-problem (0 = not in its working set; zeros after a failed solve): from the working set's lists
        if (o_mu) {
            const int nw = failed ? 0 : (int)st[ST_QNW];
            const int *wq = v.iat(P.o_wq);
            const double *sgq = v.at(P.o_sgq), *uq = v.at(P.o_uq);
            for (int k = tid; k < mt; k += NT) o_mu[(size_t)b * mt + k] = 0.0;
            T::sync();
            for (int t = tid; t < nw; t += NT) o_mu[(size_t)b * mt + wq[t]] = sgq[t] * uq[t];
        }
        if (o_sx) for (int k = tid; k < (ph + 1) * NX; k += NT) o_sx[(size_t)b * (ph + 1) * NX + k] = failed ? 0.0 : Xs[k];
        if (o_su) for (int k = tid; k < (ph + 1) * NU; k += NT) o_su[(size_t)b * (ph + 1) * NU + k] = failed ? 0.0 : Us[k];
        if (o_sy)                                           // Model::getOutput (Model.hpp:72-96): row i = out(x_i, u_i), zeros without one
            for (int i = tid; i <= ph; i += NT) {
                double y[Mdl::NY > 0 ? Mdl::NY : 1];
                for (int a = 0; a < Mdl::NY; ++a) y[a] = 0.0;
                if (Mdl::HAS_OUTPUT && !failed) Mdl::out(y, Xs + i * NX, Us + i * NU, prm);
                for (int a = 0; a < Mdl::NY; ++a) o_sy[((size_t)b * (ph + 1) + i) * Mdl::NY + a] = y[a];
            }
        // the curvature estimate stays in the workspace for a receding-horizon successor (keep_curvature)
        gwp hs = (gwp)(v.w + P.w_hinv);
        if (!again) for (int e = tid; e < nr * (nr + 1) / 2; e += NT) hs[e] = hinv[e];
        double *scal = v.w + P.w_scal;
        if (tid == 0) {
            if (S.cost) S.cost[b] = failed ? __builtin_huge_val() : st[ST_COST];
            if (S.solver_status) S.solver_status[b] = code;
            if (S.status) S.status[b] = (code == 3 || code == 4) ? 0 : (code == 5 ? 1 : 3);     // SUCCESS / MAX_ITERATION / ERROR (NLOptimizer.hpp:729-750)
            if (S.is_feasible) S.is_feasible[b] = ((mi == 0 || gmax <= S.ieq_tol) && hmax <= S.eq_tol) ? 1 : 0;    // Constraints.hpp:157-202
            if (S.iterations) S.iterations[b] = it;
            scal[0] = st[ST_COST];
        }
        T::sync();
    }
};

template <class Mdl, int WAVES, bool FL = true> constexpr int kWgWavesPerSimd = kWgWavesPerSimdOf<Mdl, WAVES, FL>::value;
// Eight wavefronts per instance: for a system whose LDS block fills a CU alone (config 5: 153 KB).  Four wavefronts would leave every SIMD with
// one -- nothing to issue while a dependent operation is in flight; eight put two on each, at the same 256 registers, and the phases that are
// loops over steps, perturbations or matrix rows (evaluation, line search, products with B^-1) take half the rounds.  Instantiated for wide states only.
template <class Mdl> constexpr bool kWgEightWaves = Mdl::NX >= 8;
// one workgroup = one instance
template <class Mdl, int WAVES, bool FL>
__global__ __launch_bounds__(64 * WAVES, (kWgWavesPerSimd<Mdl, WAVES, FL>)) void nlmpc_sqp_wg(const WgArgs A)
{
    using K = WgSqp<Mdl, WAVES, FL>;
    using T = Team<WAVES>;
    constexpr int NT = 64 * WAVES;
    const auto &M = wg_args()->M;
    const auto &S = wg_args()->S;
    const auto &P = wg_args()->P;
    const int tid = threadIdx.x, b = blockIdx.x;
    double *sm = wg_lds();
    const double *st = sm + P.o_st;
    // shader-clock cycles per phase (tools/nlmpc_phases.py): evaluate (cost, dynamics, constraints), condense, BFGS, sub-problem, step, merit, line search, update
    if (P.only_overflowed && !((int)S.ws[(size_t)b * M.ws.total + P.w_scal + 14] & 1)) return;      // (uniform over the workgroup, before any barrier)
    long long cyc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tstamp = __builtin_readcyclecounter();
#ifdef MPCX_EMU_TRACE
    auto lap = [&](int k) { const long long now = hipemu::st().n_block_syncs; cyc[k] += now - tstamp; tstamp = now; };    // (the interpreter: barriers per phase)
    tstamp = hipemu::st().n_block_syncs;
#elif defined(MPCX_NL_STATS)
    auto lap = [&](int k) { const long long now = __builtin_readcyclecounter(); cyc[k] += now - tstamp; tstamp = now; };
#else
    // (the product build keeps no phase clock: ten counters live across every out-of-line phase are ten stack slots -- scratch, i.e. memory -- read and
    // written a dozen times per iteration by every lane, on the dependent chain of a kernel that is bound by exactly that)
    auto lap = [](int) {};
    (void)cyc; (void)tstamp;
#endif
    const bool tol_on = S.ftol_abs > 0 || S.ftol_rel > 0 || S.xtol_abs > 0 || S.xtol_rel > 0;
    int it = 0, code = 5, attempt = 0;                   // nlopt codes: 3 FTOL_REACHED, 4 XTOL_REACHED, 5 MAXEVAL_REACHED, -1 FAILURE, -3 OUT_OF_MEMORY, -4 ROUNDOFF_LIMITED
    // An instance is attempted in the plan's form.  With the Schur complement kept as its inverse (errors of the inverse grow with the condition
    // number of S, the factor's with its root: working sets that fill the sub-problem's variables under tight bounds) every sub-problem's solution
    // is checked, and the first that fails is solved again with the factor, as are all that follow.  An instance that began with the inverse and
    // still FAILS is solved once more from the start with the factor alone -- the result is then, bit for bit, what a plan without the inverse
    // form delivers for it.
    for (;; ++attempt) {
    K::start(attempt);
    lap(9);
    double nu_pen = 0.0, a_prev = 0.0, f_prev = 0, step_l1 = 0, z_l1 = 0, step_max = 0;
    bool have_old = false, stepped = false, final_eval = false;
    int resets = 0, nw_keep = 0;
    it = 0; code = 5;
    for (;;) {
        K::eval_cost(final_eval ? 1 : 0);
        lap(0);
        K::eval_dyn(final_eval ? 1 : 0);
        lap(1);
        K::eval_con(final_eval ? 1 : 0);
        lap(2);
        if (final_eval) break;
        if (st[ST_ERR] != 0.0) { code = -3; break; }
        if (stepped && tol_on) {
            // nlopt's stopping rules (NLOptimizer.hpp:135-138), applied as SLSQP applies them: to a step that ended at a feasible point
            const double vmax = K::violation();
            if (vmax <= S.tol_con) {
                const double fn = st[ST_COST], df = fabs(fn - f_prev);
                const bool ft = (S.ftol_abs > 0 && df < S.ftol_abs) ||
                                (S.ftol_rel > 0 && (df < S.ftol_rel * 0.5 * (fabs(fn) + fabs(f_prev)) || fn == f_prev));
                const bool xt = (S.xtol_rel > 0 && step_l1 <= S.xtol_rel * z_l1) || (S.xtol_abs > 0 && step_max < S.xtol_abs);
                if (ft) { code = 3; break; }
                if (xt) { code = 4; break; }
            }
        }
        stepped = false;
        if (it >= S.max_iter) break;
        if (P.curv0 && it == P.curv0_it && !S.keep_curvature && resets == 0) {
            // the curvature estimate from the cost's own second derivatives at this iterate (trajectory and folded blocks are in place); the
            // phase works in the overlay the cost's gradient lies in: that part of the evaluation again
            lap(3);
            K::init_curvature();
            K::eval_cost(0);
            have_old = false;
            lap(9);                                           // (counted with "update + start": once per solve)
        }
        if (P.needs_phi) K::condense_phi(); else K::condense_chain();
        lap(3);
        if (have_old) K::bfgs(a_prev, nw_keep);
        lap(4);
        int nw = K::qp(nw_keep);
        if (nw < 0 && st[ST_MINV] != 0.0 && st[ST_ERR] == 0.0) {
            // a sub-problem that fails with the inverse (its check of the working rows, or the dual method itself) is solved again with the factor --
            // from the set it ended with where that set is what the check rejected (a good guess of the active set: the warm start sheds what does
            // not belong), cold otherwise -- and the instance stays with the factor from here on
            const int keep = nw == -5 ? (int)st[ST_QNW] : 0;
            K::leave_inverse();
            nw = K::qp(keep);
        }
        lap(5);
        if (nw < 0) { code = nw; break; }
        nw_keep = nw;
        K::step();
        lap(6);
        const double dmax = st[ST_R0], cmax = st[ST_R1], gd = st[ST_R2], zmax = st[ST_R3];
        if (dmax <= S.tol_step * fmax(1.0, zmax) && cmax <= S.tol_con) {
            // converged: take this last (tiny) step too -- it carries the final correction of the active constraints
            K::take_last_step();
            code = 4; ++it;
            final_eval = true;
            continue;
        }
        T::sync();
        K::merit(nw);
        lap(7);
        const double lam_max = st[ST_R0], viol = st[ST_R1];
        if (1.1 * lam_max > nu_pen) nu_pen = 1.5 * lam_max;
        const double phi0 = st[ST_COST] + nu_pen * viol;
        const double dphi = fmin(gd - nu_pen * viol, 0.0);
        T::sync();
        const double a_step = K::linesearch(nu_pen, phi0, dphi);
        lap(8);
        if (a_step < 0.0) {                                  // no decrease left within 2^-40: the iteration has stalled
            // (every sub-problem solution of the inverse form has passed its check, so the stall is the iteration's own; the instance goes on with
            // the factor all the same -- what is left of it is the hard part)
            if (st[ST_MINV] != 0.0) K::leave_inverse();
            if (cmax <= fmax(S.tol_con, 1e-8) && dmax <= 1e-3 * fmax(1.0, zmax)) { code = 4; break; }
            if (resets >= 5) { code = -4; break; }
            // far from a solution: the curvature estimate has gone bad -- forget it and try a steepest-descent-like step
            ++resets;
            K::reset_hessian();
            have_old = false;
            ++it;
            continue;
        }
        f_prev = st[ST_COST];
        T::sync();
        K::update(a_step);
        lap(9);
        step_l1 = st[ST_R0]; z_l1 = st[ST_R1]; step_max = st[ST_R2];
        T::sync();
        a_prev = a_step; have_old = true; stepped = true;
        ++it;
    }
    if (!(code < 0 && attempt == 0 && P.minv && st[ST_ERR] == 0.0)) break;
    T::sync();                                               // (everybody has read the attempt's scalars before start() clears them)
    }
    K::finish(code, it);
    MPCX_TRACE("barriers per iteration (%d iterations): cost %.1f dyn %.1f con %.1f condense %.1f bfgs %.1f qp %.1f step %.1f merit %.1f ls %.1f update+start %.1f\n", it,
               (double)cyc[0] / NT / it, (double)cyc[1] / NT / it, (double)cyc[2] / NT / it, (double)cyc[3] / NT / it, (double)cyc[4] / NT / it, (double)cyc[5] / NT / it,
               (double)cyc[6] / NT / it, (double)cyc[7] / NT / it, (double)cyc[8] / NT / it, (double)cyc[9] / NT / it);
    if (tid == 0) {
        double *scal = S.ws + (size_t)b * M.ws.total + P.w_scal;
        scal[1] = st[ST_R5];
        scal[12] = st[ST_R5 + 1];                                // the largest working set of the solve
        scal[13] = st[ST_NSHED];
        // bit 0: a pass with the full capacity has to take this instance again; bit 1: solved again from the start with the factor; bit 2: left the inverse form on the way
        scal[14] = st[ST_OVER] + 2.0 * (double)attempt + 4.0 * st[ST_SWITCHED];
        scal[15] = st[ST_NCARRY];
#ifdef MPCX_NL_STATS
        for (int k = 0; k < 16; ++k) scal[16 + k] = st[ST_QSTAT + k];  // (beyond the statistics block: a part of the workspace this form does not use)
#endif
#if defined(MPCX_NL_STATS) || defined(MPCX_EMU_TRACE)
        for (int k = 0; k < 10; ++k) scal[2 + k] = (double)cyc[k];
#endif
    }
    (void)NT;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
#if !defined(__HIPCC_RTC__)
// the LDS / workspace plan of the workgroup form for controller m (dimensions, bounds) and the hard / soft flag; 0, or -2 if the
// shape does not fit (the caller falls back to nlmpc_sqp)
// (minv_wanted / carry_wanted / curv_wanted: -1 the plan's own choice, 0 | 1 forced -- the launcher's MPCX_NLMPC_MINV / MPCX_NLMPC_CARRY / MPCX_NLMPC_CURV0, read when the handle is created)
template <class Mdl>
inline int wg_plan(const NlmpcDev &m, int hard, int waves_wanted, int state_bounds, WgPlan &P, int blocks_wanted = -1, bool cut_ok = true, int lds_per_cu = 160 * 1024,
                   int minv_wanted = -1, int carry_wanted = -1, int curv_wanted = -1, int curv_it = -1, int inv_wanted = -1)
{
    constexpr int NX = Mdl::NX, NU = Mdl::NU, FW = NX + NU + 1;
    const int ph = m.ph, nxs = ph * NX, nr = m.nr, nz = m.nz, mi = m.nineq, mu_ = mi + m.nue, mt = mu_ + m.nbnd;
    if (ph > 64 || ph >= 255 || mu_ >= (1 << 19) || 3 * NX + NU + 1 > 64 || m.nr >= 0xffff) return -2;
    P = WgPlan{};
    P.hard = hard ? 1 : 0;
    P.nq = hard ? m.nzu : nr;
    // structure: dense columns and Jacobian block slots
    int nsx = 0, ndu = 0;
    bool reads_x = false;
    for (int k = 0; k < mu_; ++k) {
        int cnt = 0;
        for (int i = 1; i <= ph; ++i) cnt += (k < mi ? Mdl::ineq_reads_x(k, i) : Mdl::eq_reads_x(k - mi, i)) ? 1 : 0;
        nsx += cnt;
        if (cnt > 0) reads_x = true;
        if (cnt > 0 || !Mdl::XFREE_ROWS_SPARSE) ++ndu;
    }
    const int nsb = state_bounds;                          // finite bounds on states: the first rows of the bound table
    if (nsx >= 4096) return -2;
    P.nsx = nsx; P.nd_user = ndu; P.nsb = nsb; P.nd = ndu + nsb; P.ndld = (P.nd + 1) | 1;
    {
        int tot = 0;
        for (int k = 0; k < mu_; ++k) {
            bool dense = !Mdl::XFREE_ROWS_SPARSE;
            for (int i = 1; i <= ph && !dense; ++i) dense = k < mi ? Mdl::ineq_reads_x(k, i) : Mdl::eq_reads_x(k - mi, i);
            if (dense) tot += wg_row_len<Mdl>(k, ph, m.ch, P.nq, m.nzu, mi) & 0xffff;
        }
        tot += nsb * (NX > 8 ? P.nq : (wg_bound_row_len(m.nzu) & 0xffff));
        P.art_total = tot;
    }
    P.needs_phi = (reads_x || nsb > 0) ? 1 : 0;
    // a working set holds linearly independent rows: never more than there are rows or sub-problem variables
    auto imin = [](int a, int b) { return a < b ? a : b; };
    auto imax = [](int a, int b) { return a > b ? a : b; };
    const int kw_full = imin(kNlMaxWorking, imax(2, imin(mt, P.nq) + 1));
    // working sets that can hold more than 32 rows: the Schur complement's inverse instead of its factor (see ws_invert_m; since the inverse is
    // carried between sub-problems it pays from there on -- six oscillators, 61 rows: 79 k -> 90 k solves/s; below, the factor's one-wavefront
    // substitutions are the shorter way).  An instance the inverse form fails is solved again with the factor (the kernel's attempt loop).
    const int minv_env = minv_wanted;
    // (only where every row is one of the short lists -- bounds, constraints on single inputs: S is then all but a principal block of B^-1 and as
    // well conditioned; with dense rows through the sensitivities -- config 3's obstacle rows, two of them nearly parallel at a time -- the
    // inverse lost 3 instances of 4096 that the factor solves)
    P.minv = minv_env >= 0 ? (minv_env ? 1 : 0) : ((P.nd == 0 && (kw_full > 64 || (kw_full > 32 && (mu_ == 0 || Mdl::XFREE_ROWS_AFFINE)))) ? 1 : 0);      // (33 .. 64: where the inverse can be carried)
    int waves = waves_wanted;
    if (waves != 0 && waves != 1 && waves != 2 && waves != 4 && !(waves == 8 && kWgEightWaves<Mdl>)) return -2;
    auto layout = [&](int kw, int f_lds) {
        int o = kWgCtxDoubles;
        auto take = [&](int n) { const int at = o; o += (n + 1) & ~1; return at; };
        P.kw = kw; P.f_lds = f_lds;
        P.o_red = take(kWgRedDoubles); P.o_st = take(ST_TOTAL);
        P.o_z = take(nz); P.o_c = take(nxs); P.o_gin = take(mu_); P.o_gu = take(nr); P.o_gr = take(nr);
        P.o_glold = take(nr); P.o_sv = take(nr); P.o_hinv = take(nr * (nr + 1) / 2);
        P.o_mu = take(P.needs_phi ? mt : 0); P.o_flag = take((mt + 1) / 2); P.o_br = take(mt); P.o_s1v = take(mt); P.o_s1m = take((mt + 1) / 2);
        P.o_dcol = take((mt + 1) / 2); P.o_xmask = take(P.needs_phi ? mu_ : 0); P.o_jxoff = take(P.needs_phi ? (mu_ + 2) / 2 : 0); P.o_slot = take((nsx + 1) / 2);
        P.o_sbf = take((ph + 2) / 2); P.o_jx = take(nsx * NX); P.o_art = f_lds ? take(P.art_total) : 0;
        P.o_wq = take((kw + 1) / 2); P.o_sgq = take(kw); P.o_uq = take(kw); P.o_tq = take(kw); P.o_invd = take(kw);
        P.o_xq = take(nr); P.o_p = P.o_xq;        // (the sub-problem's iterate is the step when it ends)
        P.o_np = take(nr); P.o_vv = take(nr); P.o_wv = take(nr);
        P.o_prm = take(Mdl::NPARAMS); P.o_cd = take(P.ndld); P.o_yd = take(P.ndld);
        P.o_bidx = take((m.nbnd + 1) / 2); P.o_bsign = take(m.nbnd); P.o_bval = take(m.nbnd);
        P.o_xrf = take((ph + 3) / 2); P.o_xre = take((nsx + 1) / 2); P.o_drow = take((P.nd_user + 1) / 2);
        P.o_mbuf = P.minv ? take(kw) : 0;
        P.o_aoff = take((P.nd + 2) / 2); P.o_alen = take((P.nd + 1) / 2);
        P.o_F = f_lds ? take(ph * NX * FW) : 0;
        const int ov = o;
        P.o_Xs = take((ph + 1) * NX); P.o_Us = take((ph + 1) * NU); P.o_dXs = take((ph + 1) * NX); P.o_dUs = take((ph + 1) * NU);
        P.o_Jm = take(ph * NU); P.o_lam = take(nxs); P.o_dx = take(nxs);
        const int endA = o;
        o = ov;
        P.o_L = take(kw * (kw + 1) / 2);
        P.lds_total = imax(endA, o);
        return (size_t)P.lds_total * sizeof(double);
    };
    // How many workgroups a CU holds is decided by the LDS block (160 KB per CU): with one wavefront per SIMD nothing hides the latency of
    // a dependent LDS access, a second or third workgroup does.  The plan therefore takes the smallest budget (most workgroups per CU, up to
    // the 32 wavefronts a CU runs) that holds the whole problem; where only the factor of the largest possible working set stands in the
    // way of one more workgroup per CU, its capacity is cut -- not below 46 rows.  An instance whose working set outgrows the cut capacity is marked
    // (scal[14]) and taken again by a second launch planned with cut_ok = false (WgPlan::only_overflowed: every other workgroup of that launch
    // returns at once); only one beyond kNlMaxWorking ends with nlopt's OUT_OF_MEMORY code (status ERROR), as it always did.
    const int kw_floor = imin(kw_full, 46);
    bool placed = false;
    auto place = [&]() {
        placed = false;
        for (int per_cu = imin(16, 32 / P.waves); per_cu >= 1 && !placed; --per_cu) {
            const size_t budget = (size_t)(lds_per_cu / per_cu) & ~(size_t)15;
            // at this many workgroups per CU: the full capacity first (blocks in LDS, then in the workspace), a cut one only if neither fits
            for (int cut = 0; cut <= 1 && !placed; ++cut) {
                if (cut && (per_cu == 1 || !cut_ok)) continue;  // (alone on the CU the factor keeps its full capacity)
                for (int f_lds = 1; f_lds >= 0 && !placed; --f_lds) {
                    if (blocks_wanted >= 0 && f_lds != (blocks_wanted ? 1 : 0)) continue;
                    // the registers of the variant (kWgWavesPerSimd) bound the workgroups per CU as well
                    const int by_regs = 4 * (P.waves == 8 ? 2 : P.waves == 4 ? (f_lds ? kWgWavesPerSimd<Mdl, 4, true> : kWgWavesPerSimd<Mdl, 4, false>) : P.waves == 2 ? kWgWavesPerSimd<Mdl, 2> : kWgWavesPerSimd<Mdl, 1>) / P.waves;
                    if (per_cu > by_regs) continue;
                    int kw = kw_full;
                    if (cut) while (kw > kw_floor && layout(kw, f_lds) > budget) --kw;
                    if (layout(kw, f_lds) <= budget) { placed = true; P.per_cu = per_cu; P.cut = kw < kw_full ? 1 : 0; }
                }
            }
        }
    };
    // Wavefronts per instance.  What a CU delivers is instances in flight: a small problem (eight or more blocks per CU at one wavefront each)
    // gets one wavefront per instance and fills the CU with instances; a larger one, whose LDS block leaves room for one or two
    // instances, gets four wavefronts (two for a medium one) -- that form is for latency, the launcher prefers nlmpc_sqp for throughput.
    if (waves == 0) {
        P.waves = 1;
        place();
        if (!placed || P.per_cu < 8) { P.waves = nz >= 96 ? 4 : 2; place(); }
        if (kWgEightWaves<Mdl> && placed && P.waves == 4 && P.per_cu == 1) { P.waves = 8; place(); }     // alone on its CU: two wavefronts per SIMD
    } else {
        P.waves = waves;
        place();
    }
    if (!placed) return -2;
    // the curvature estimate set to the condensed Gauss-Newton Hessian (init_curvature): where the cost is a sum over the horizon's rows and the
    // sub-problem's four vectors hold an NX x NX block
    P.curv0_it = curv_it >= 0 ? curv_it : Mdl::CURV0_AFTER;
    P.curv0 = (curv_wanted != 0 && Mdl::COST_STAGEWISE && NX * NX <= 4 * ((nr + 1) & ~1)) ? 1 : 0;
    P.curv_lds = P.o_Us + (((ph + 1) * NU + 1) & ~1) + 2 * NX * m.nzu <= P.lds_total ? 1 : 0;
    const int n16 = (P.nq + 15) & ~15;
    P.inv_nb = inv_wanted >= 0 ? (inv_wanted >= 16 ? 16 : inv_wanted >= 8 ? 8 : 0) : (P.nq > 96 ? 16 : 8);      // (by the problem's size alone: see invert_packed_blocked)
    P.inv_lds = P.lds_total - (P.o_Us + (((ph + 1) * NU + 1) & ~1)) >= P.inv_nb * P.inv_nb + 2 * P.inv_nb * n16 ? 1 : 0;
    {
        int o = 0;
        auto take = [&](int n) { const int at = o; o += (n + 1) & ~1; return at; };
        P.w_F = take(P.f_lds ? 0 : ph * NX * FW);
        P.w_art = take(P.f_lds ? 0 : P.art_total);
        P.w_einv = take(Mdl::CONTINUOUS ? ph * NX * NX : 0);
        P.w_gx = take(nxs);
        P.w_hinv = take(nr * (nr + 1) / 2);
        P.w_sp = take(mt * kNlSparse + (mt * kNlSparse + 1) / 2);
        {
            const int phi = (P.curv0 && !P.curv_lds) ? 2 * NX * m.nzu : 0, inv = (P.curv0 && P.inv_nb && !P.inv_lds) ? P.inv_nb * P.inv_nb + 2 * P.inv_nb * n16 : 0;
            P.w_phi = take(phi > inv ? phi : inv);              // (one after the other: the sensitivities' buffers, then the inversion's)
        }
        // the carried inverse (see ws_warm): where no row's entries change between sub-problems -- bounds on inputs, user rows affine in the inputs (Mdl::XFREE_ROWS_AFFINE) --
        // and the controller's workspace has the room
        const int carry_env = carry_wanted < 0 ? 1 : carry_wanted;
        P.carry_m = 0; P.w_msave = 0;
        if (carry_env && P.minv && P.nd == 0 && (mu_ == 0 || Mdl::XFREE_ROWS_AFFINE) && o + P.kw * (P.kw + 1) / 2 + 2 <= m.ws.scal) { P.carry_m = 1; P.w_msave = take(P.kw * (P.kw + 1) / 2); }
        P.ws_total = o;
        // the instances keep the stride of the controller's workspace (NlmpcWsLayout), the statistics block its place in it
        P.w_scal = m.ws.scal;
        if (P.ws_total > m.ws.scal || m.ws.scal + 16 > m.ws.total) return -2;
#ifdef MPCX_NL_STATS
        if (m.ws.scal + 32 > m.ws.total) return -2;            // (the statistics build files the sub-problem's cycle counts behind the block)
#endif
    }
    return 0;
}

template <class Mdl>
int launch_solve_wg(const NlmpcDev *m, const NlmpcSolveDev *b, const WgPlan *P, void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)P->lds_total * sizeof(double);
    auto go = [&](auto kern) {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        static const bool show = getenv("MPCX_DEBUG_OCCUPANCY") != nullptr;      // (read once)
        if (show) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, P->waves * 64, lds);
            fprintf(stderr, "nlmpc_sqp_wg: %d workgroups of %d wavefronts, %zu bytes of LDS each; resident per CU: %d; blocks in LDS %d\n",
                    b->batch, P->waves, lds, nb, P->f_lds);
        }
        const WgArgs A{*m, *b, *P};
        hipLaunchKernelGGL(kern, dim3(b->batch), dim3(P->waves * 64), lds, s, A);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    };
    if constexpr (kWgEightWaves<Mdl>) {
        if (P->waves == 8) return P->f_lds ? go(nlmpc_sqp_wg<Mdl, 8, true>) : go(nlmpc_sqp_wg<Mdl, 8, false>);
    }
    if (P->f_lds) {
        if (P->waves == 4) return go(nlmpc_sqp_wg<Mdl, 4, true>);
        if (P->waves == 2) return go(nlmpc_sqp_wg<Mdl, 2, true>);
        return go(nlmpc_sqp_wg<Mdl, 1, true>);
    }
    if (P->waves == 4) return go(nlmpc_sqp_wg<Mdl, 4, false>);
    if (P->waves == 2) return go(nlmpc_sqp_wg<Mdl, 2, false>);
    return go(nlmpc_sqp_wg<Mdl, 1, false>);
}
#endif   // !__HIPCC_RTC__

}  // namespace engine
}  // namespace mpcx
