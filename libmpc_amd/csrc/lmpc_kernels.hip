// Batched linear-MPC kernels for gfx950 (MI355X).
//
// Replace, for a batch of instances sharing one controller, what the reference does per
// call in LOptimizer::run (include/mpc/LMPC/LOptimizer.hpp:189-368):
//
//   lmpc_assemble_*   ProblemBuilder::get (ProblemBuilder.hpp:528-633): per-instance linear
//                     term, constraint offsets and cost constant of the condensed QP from
//                     (x0, lastU, references, exogenous inputs), plus the unconstrained
//                     optimum.
//                       _generic  one instance per wavefront, horizon roll-out + adjoint pass
//                                 (any reference layout, disturbances, per-step references).
//   lmpc_solve        osqp_setup/osqp_solve (LOptimizer.hpp:261,284) + unpacking (:305-347):
//                     one instance per wavefront; active-set polish (OSQP's polish, iterated
//                     until the KKT conditions verify) on a Schur complement staged in LDS,
//                     with ADMM iterations on the shared pre-inverted ADMM matrix as the
//                     globalisation step whenever the polish does not verify.
//
// A wavefront owns its instance from load to store; nothing is exchanged between
// wavefronts, so there is no workgroup barrier anywhere.  Per-instance vectors live in
// registers as element pairs (lmpc_device.hpp); the vector a mat-vec broadcasts is staged
// through the wave's LDS slice; the shared matrices stream from L2 with 16-byte loads that
// are coalesced across the wavefront and never predicated (out-of-range lanes re-read
// element 0), so the compiler keeps many of them in flight.
#include "lmpc_kernel_common.hpp"

namespace mpcx {

namespace {

// =====================================================================================
// assemble, generic form: one instance per wavefront
// =====================================================================================
// global -> registers -> LDS, K chunks of 64 elements per lane.  The loads are never predicated (an out-of-range lane re-reads element 0)
// and every fetch of a group is issued before the first LDS store waits for one: the wave pays one memory latency per GROUP of arrays.
// That is what this kernel's time is made of -- a dependent global load costs ~2.7k cycles on the loaded chip and the first form of this
// function had ~80 of them in series (one per roll-out / adjoint step, one per reference element); now the dependent chains read LDS only.
template <int K, class F>
__device__ __forceinline__ void fetch(double (&v)[K], int n, int lane, F &&at)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = lane + 64 * k;
        v[k] = at(e < n ? e : 0);
    }
}
template <int K, class F>
__device__ __forceinline__ void put(double *dst, const double (&v)[K], int n, int lane, F &&at)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = lane + 64 * k;
        if (e < n) dst[e] = v[k];
    }
    for (int e = lane + 64 * K; e < n; e += 64) dst[e] = at(e);      // shapes past 64 K elements: the slow way
}

template <int CPZ, int CPG>
__device__ void assemble_one(const LmpcDev &M, const LmpcBatchDev &Bt, const int b, const int lane,
                             double *stage, double *arena, gdw ws)
{
    constexpr int NZS = 2 * CPZ, NGS = 2 * CPG, CPY = CPZ + CPG, KF = 4;
    const int nx = M.nx, nu = M.nu, ny = M.ny, ndu = M.ndu, ph = M.ph;
    const int nz = M.nz, ldz = M.ldz, ldg = M.ldg, ldy = M.ldy;
    const bool has_dist = M.has_dist != 0;
    const double ea = M.eps_abs, er = M.eps_rel;
    // a null reference pointer (heterogeneous batches, "shared" mode): this model's own reference arrays
    const gdp gdm = gl(Bt.dmeas ? Bt.dmeas : M.dmeas_s), gyr = gl(Bt.yref ? Bt.yref : M.yref_s), gur = gl(Bt.uref ? Bt.uref : M.uref_s),
              gdr = gl(Bt.duref ? Bt.duref : M.duref_s);

    // the wave's LDS plan (lmpc_lds_per_wave sizes it)
    const int ney = (ph + 1) * ny, nuu = ph * nu;
    double *xb = arena;                       // free response, (ph+1) x nx
    double *ey = xb + (ph + 1) * nx;          // output reference, then the weighted output error, (ph+1) x ny
    double *wys = ey + ney;                   // output weights, (ph+1) x ny
    double *pall = wys + ney;                 // adjoint states of every step, (ph+2) x nx (slot ph+1 = 0)
    double *wur = pall + (ph + 2) * nx;       // Wu_i uref_{i-1}, then the input gradient of step i, ph x nu
    double *dur = wur + nuu;                  // Wdu_i duref_{i-1}, ph x nu
    double *u0s = dur + nuu;                  // lastU
    double *sxu = u0s + nu;                   // the scalar row's coefficients, nx + nu
    double *dms = sxu + nx + nu;              // measured disturbance of every step, ph x ndu
    int *blks = reinterpret_cast<int *>(dms + ph * ndu);         // move-blocking map, ph+1 ints
    // the model matrices: the roll-out and the adjoint pass are chains of ph dependent steps that read them in every step
    double *gA = dms + ph * ndu + (ph + 2) / 2, *gB = gA + nx * nx, *gC = gB + nx * nu, *gBd = gC + ny * nx, *gDd = gBd + nx * ndu;

    // ---- gather: everything this instance reads from global memory, in two groups of loads ----
    const gdp mA = GP(A), mB = GP(B), mC = GP(C), mBd = GP(Bd), mDd = GP(Dd);
    const gip gblk = GP(blk);
    auto atA = [&](int e) { return mA[e]; };
    auto atB = [&](int e) { return mB[e]; };
    auto atC = [&](int e) { return mC[e]; };
    auto atW = [&](int e) { return GP(Wy)[e]; };
    auto atR = [&](int e) { const int i = e / ny, a = e - i * ny; return ref_at(gyr, Bt.yref_bs, Bt.yref_ks, b, i > 0 ? i - 1 : 0, a); };
    auto atU = [&](int e) { const int k = e / nu, j = e - k * nu; return GP(Wu)[e + nu] * ref_at(gur, Bt.uref_bs, Bt.uref_ks, b, k, j); };
    auto atD = [&](int e) { const int i = e / nu, j = e - i * nu; return GP(Wdu)[e] * ref_at(gdr, Bt.duref_bs, Bt.duref_ks, b, i > 0 ? i - 1 : 0, j); };
    double rA[KF], rB[KF], rC[KF];
    fetch<KF>(rA, nx * nx, lane, atA);
    fetch<KF>(rB, nx * nu, lane, atB);
    fetch<KF>(rC, ny * nx, lane, atC);
    const int lx = lane < nx ? lane : 0, lu = lane < nu ? lane : 0, ly = lane < ny ? lane : 0;
    const double x0v = gl(Bt.x0)[(size_t)b * nx + lx], u0v = gl(Bt.u0)[(size_t)b * nu + lu];
    const double lo0x = GP(lo0x)[lx], hi0x = GP(hi0x)[lx], lo0u = GP(lo0u)[lu], hi0u = GP(hi0u)[lu], lo0y = GP(lo0y)[ly], hi0y = GP(hi0y)[ly];
    const double sxv = GP(sX)[lx], suv = GP(sU)[lu];
    const double wu0 = GP(Wu)[lu], wdu0 = GP(Wdu)[lu];
    const double ur0 = ref_at(gur, Bt.uref_bs, Bt.uref_ks, b, 0, lu), dr0 = ref_at(gdr, Bt.duref_bs, Bt.duref_ks, b, 0, lu);
    const int blkv = gblk[lane <= ph ? lane : 0];
    double rW[KF], rR[KF], rU[KF], rD[KF];
    fetch<KF>(rW, ney, lane, atW);
    fetch<KF>(rR, ney, lane, atR);
    fetch<KF>(rU, nuu, lane, atU);
    fetch<KF>(rD, nuu, lane, atD);
    int rkind[NGS], rstep[NGS], rcomp[NGS];
    double rlg[NGS], rug[NGS];
#pragma unroll
    for (int s = 0; s < NGS; ++s) {
        const int r = 128 * (s >> 1) + 2 * lane + (s & 1), rr = r < ldg ? r : 0;
        rkind[s] = GP(g_kind)[rr]; rstep[s] = GP(g_step)[rr]; rcomp[s] = GP(g_comp)[rr];
        rlg[s] = GP(lg0)[rr]; rug[s] = GP(ug0)[rr];
    }

    put<KF>(gA, rA, nx * nx, lane, atA);
    put<KF>(gB, rB, nx * nu, lane, atB);
    put<KF>(gC, rC, ny * nx, lane, atC);
    if (lane < nx) { xb[lane] = x0v; sxu[lane] = sxv; pall[(ph + 1) * nx + lane] = 0.0; }
    if (lane < nu) { u0s[lane] = u0v; sxu[nx + lane] = suv; }
    if (lane <= ph) blks[lane] = blkv;
    for (int e = lane + 64; e <= ph; e += 64) blks[e] = gblk[e];
    put<KF>(wys, rW, ney, lane, atW);
    put<KF>(ey, rR, ney, lane, atR);
    put<KF>(wur, rU, nuu, lane, atU);
    put<KF>(dur, rD, nuu, lane, atD);
    if (has_dist) {
        for (int e = lane; e < nx * ndu; e += 64) gBd[e] = mBd[e];
        for (int e = lane; e < ny * ndu; e += 64) gDd[e] = mDd[e];
        for (int e = lane; e < ph * ndu; e += 64) { const int k = e / ndu; dms[e] = ref_at(gdm, Bt.dmeas_bs, Bt.dmeas_ks, b, k, e - k * ndu); }
    }
    for (int e = lane; e < ldz; e += 64) stage[e] = 0.0;
    wave_sync();

    auto dm = [&](int k, int dd) -> double { return dms[k * ndu + dd]; };

    // step-0 rows involve no decision variable: pure feasibility conditions on (x0, lastU)
    bool bad = false;
    if (lane < nx) bad |= violates(x0v, lo0x, hi0x, ea, er);
    if (lane < nu) bad |= violates(u0v, lo0u, hi0u, ea, er);
    if (lane < ny) {
        double y0 = 0;
        for (int c = 0; c < nx; ++c) y0 = fma(gC[lane + c * ny], xb[c], y0);
        if (has_dist) for (int dd = 0; dd < ndu; ++dd) y0 = fma(gDd[lane + dd * ny], dm(0, dd), y0);
        bad |= violates(y0, lo0y, hi0y, ea, er);
    }
    if (lane == 0) {
        double s0 = 0;
        for (int c = 0; c < nx; ++c) s0 = fma(sxu[c], xb[c], s0);
        for (int c = 0; c < nu; ++c) s0 = fma(sxu[nx + c], u0s[c], s0);
        bad |= violates(s0, M.s0lo, M.s0hi, ea, er);
    }

    // horizon roll-out of the free response
    for (int i = 1; i <= ph; ++i) {
        if (lane < nx) {
            const double *xp = xb + (i - 1) * nx;
            double s = 0;
#pragma unroll 4
            for (int c = 0; c < nx; ++c) s = fma(gA[lane + c * nx], xp[c], s);
            if (has_dist) for (int dd = 0; dd < ndu; ++dd) s = fma(gBd[lane + dd * nx], dm(i - 1, dd), s);
            xb[i * nx + lane] = s;
        }
        wave_sync();
    }
    double c0 = 0;
    for (int idx = lane; idx < ney; idx += 64) {
        const int i = idx / ny, a = idx - i * ny, k = i > 0 ? i - 1 : 0;
        double cx = 0;
#pragma unroll 4
        for (int c = 0; c < nx; ++c) cx = fma(gC[a + c * ny], xb[i * nx + c], cx);
        double r = ey[idx];
        if (has_dist) for (int dd = 0; dd < ndu; ++dd) r -= gDd[a + dd * ny] * dm(k, dd);
        const double w = wys[idx];
        ey[idx] = w * (cx - r);
        c0 += w * (0.5 * cx * cx - r * cx);
    }
    if (lane < nu) {
        const double u = u0v;
        c0 += wu0 * (0.5 * u * u - ur0 * u);
        c0 += wdu0 * (0.5 * u * u + dr0 * u);
    }
    c0 = wave_sum(c0);

    // constraint rows: bounds shifted by the free response
    const double INF = __builtin_huge_val();
    double lg[NGS], ug[NGS];
#pragma unroll
    for (int s = 0; s < NGS; ++s) {
        const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
        lg[s] = -INF; ug[s] = INF;
        if (r < ldg) {
            const int kind = rkind[s], st = rstep[s], cp = rcomp[s];
            double off;
            if (kind == 0) off = xb[st * nx + cp];
            else if (kind == 1) {
                off = 0;
                for (int c = 0; c < nx; ++c) off = fma(gC[cp + c * ny], xb[st * nx + c], off);
                if (has_dist) for (int dd = 0; dd < ndu; ++dd) off = fma(gDd[cp + dd * ny], dm(st - 1, dd), off);
            } else {
                off = 0;
                for (int c = 0; c < nx; ++c) off = fma(sxu[c], xb[st * nx + c], off);
            }
            lg[s] = rlg[s] - off; ug[s] = rug[s] - off;
        }
    }
    for (int idx = lane; idx < M.n_fixed; idx += 64) {
        const int kind = GP(f_kind)[idx], st = GP(f_step)[idx], cp = GP(f_comp)[idx];
        double v = 0;
        if (kind == 0) v = xb[st * nx + cp];
        else if (kind == 1) {
            for (int c = 0; c < nx; ++c) v = fma(gC[cp + c * ny], xb[st * nx + c], v);
            if (has_dist) for (int dd = 0; dd < ndu; ++dd) v = fma(gDd[cp + dd * ny], dm(st - 1, dd), v);
        } else {
            for (int c = 0; c < nx; ++c) v = fma(sxu[c], xb[st * nx + c], v);
        }
        bad |= violates(v, GP(f_lo)[idx], GP(f_hi)[idx], ea, er);
    }
    const bool infeasible0 = wave_any(bad);
    wave_sync();                              // ey is complete

    // adjoint pass -> linear term.  p_i = C' ey_i + A' p_{i+1}: the output terms of every step at once, then the chain of ph steps
    // (one LDS mat-vec each), then the input gradients B' p_i of every step at once.
    for (int e = lane; e < ph * nx; e += 64) {
        const int i = e / nx + 1, r = e - (i - 1) * nx;
        double s = 0;
#pragma unroll 4
        for (int a = 0; a < ny; ++a) s = fma(gC[a + r * ny], ey[i * ny + a], s);
        pall[i * nx + r] = s;
    }
    wave_sync();
    for (int i = ph; i >= 1; --i) {
        if (lane < nx) {
            const double *pin = pall + (i + 1) * nx;
            double s = pall[i * nx + lane];
#pragma unroll 4
            for (int a = 0; a < nx; ++a) s = fma(gA[a + lane * nx], pin[a], s);
            pall[i * nx + lane] = s;
        }
        wave_sync();
    }
    for (int e = lane; e < nuu; e += 64) {
        const int i = e / nu + 1, j = e - (i - 1) * nu;
        double g = 0;
#pragma unroll 4
        for (int a = 0; a < nx; ++a) g = fma(gB[a + j * nx], pall[i * nx + a], g);
        wur[e] = g - wur[e];
    }
    wave_sync();
    if (lane < nu) {
        const int j = lane;
        for (int i = ph; i >= 1; --i) stage[blks[i] * nu + j] += wur[(i - 1) * nu + j];
        stage[blks[1] * nu + j] -= wdu0 * (u0v + dr0);
        for (int i = 1; i < ph; ++i) {
            const int bn = blks[i + 1], bp = blks[i];
            if (bn != bp) {
                const double t = -dur[i * nu + j];
                stage[bn * nu + j] += t;
                stage[bp * nu + j] -= t;
            }
        }
    }
    wave_sync();
    double f[NZS], nf_[NZS];
#pragma unroll
    for (int s = 0; s < NZS; ++s) {
        const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
        f[s] = e < ldz ? stage[e] : 0.0;
        nf_[s] = -f[s];
    }
    wave_sync();
    stage_store<CPZ>(stage, nf_, ldz, lane);
    wave_sync();

    // unconstrained optimum t0 = -Hinv f and its image G t0 under the constraint rows
    if (Bt.n_models > 0) {
        // Every instance its own controller: the factors come from HBM, and G Hinv (ldg x nz) is 40 % of them.  G t0 needs no matrix:
        // the rows of G are states, outputs and scalar rows of the response to the inputs, so G t0 is read off a roll-out
        // x_i = A x_{i-1} + B t0[blk(i)] from x_0 = 0 with the model matrices that are in LDS already.  Only Hinv is fetched.
        double t0[NZS];
#pragma unroll
        for (int s = 0; s < NZS; ++s) t0[s] = 0;
        matvec_acc<CPZ, (CPZ == 1 ? 16 : (CPZ == 2 ? 8 : 4))>(GP(Y), ldy, ldz, nz, stage, t0, lane);
        wave_sync();                          // every lane is done with -f
        stage_store<CPZ>(stage, t0, ldz, lane);
        if (lane < nx) xb[lane] = 0.0;
        wave_sync();
        for (int i = 1; i <= ph; ++i) {
            if (lane < nx) {
                const double *xp = xb + (i - 1) * nx, *ui = stage + blks[i] * nu;
                double s = 0;
#pragma unroll 4
                for (int c = 0; c < nx; ++c) s = fma(gA[lane + c * nx], xp[c], s);
                for (int j = 0; j < nu; ++j) s = fma(gB[lane + j * nx], ui[j], s);
                xb[i * nx + lane] = s;
            }
            wave_sync();
        }
        double gt0[NGS];
#pragma unroll
        for (int s = 0; s < NGS; ++s) {
            const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
            gt0[s] = 0.0;
            if (r < ldg) {
                const int kind = rkind[s], st = rstep[s], cp = rcomp[s];
                double v = 0;
                if (kind == 0) v = xb[st * nx + cp];
                else if (kind == 1) {
                    for (int c = 0; c < nx; ++c) v = fma(gC[cp + c * ny], xb[st * nx + c], v);
                } else {
                    for (int c = 0; c < nx; ++c) v = fma(sxu[c], xb[st * nx + c], v);
                    for (int j = 0; j < nu; ++j) v = fma(sxu[nx + j], stage[blks[st] * nu + j], v);
                }
                gt0[s] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < CPZ; ++c) {
            const int e = 128 * c + 2 * lane;
            if (e < ldz) { st2(ws + e, f[2 * c], f[2 * c + 1]); st2(ws + ldz + e, t0[2 * c], t0[2 * c + 1]); }
        }
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const int r = 128 * c + 2 * lane;
            if (r < ldg) st2(ws + ldz + ldz + r, gt0[2 * c], gt0[2 * c + 1]);
        }
    } else {
        // one controller for the batch: the first nz columns of Y = [Hinv; G Hinv] stream from L2 in one pass (the record keeps t0 | gt0
        // back to back, as Y's rows are)
        double tg[2 * CPY];
#pragma unroll
        for (int s = 0; s < 2 * CPY; ++s) tg[s] = 0;
        matvec_acc<CPY, (CPY <= 2 ? 8 : (CPY <= 4 ? 4 : 2))>(GP(Y), ldy, ldy, nz, stage, tg, lane);
#pragma unroll
        for (int c = 0; c < CPZ; ++c) {
            const int e = 128 * c + 2 * lane;
            if (e < ldz) st2(ws + e, f[2 * c], f[2 * c + 1]);
        }
#pragma unroll
        for (int c = 0; c < CPY; ++c) {
            const int e = 128 * c + 2 * lane;
            if (e < ldy) st2(ws + ldz + e, tg[2 * c], tg[2 * c + 1]);
        }
    }
    // workspace record: f | t0 | gt0 | lg | ug | c0, flag
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const int r = 128 * c + 2 * lane;
        if (r < ldg) {
            st2(ws + ldz + ldy + r, lg[2 * c], lg[2 * c + 1]);
            st2(ws + ldz + ldy + ldg + r, ug[2 * c], ug[2 * c + 1]);
        }
    }
    if (lane == 0) st2(ws + ldz + ldy + 2 * ldg, c0, infeasible0 ? 1.0 : 0.0);
    wave_sync();
}

template <int CPZ, int CPG>
__global__ __launch_bounds__(kWavesPerBlock * 64) void lmpc_assemble_generic(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt, double *wsbase)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LmpcDev &M0 = *Mp;                       // dimensions and the LDS plan are the same for every model of a heterogeneous batch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *stage = smem + (size_t)wave * M0.lds_per_wave;
    double *arena = stage + M0.stage_len + M0.ldy;
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;             // one instance per wavefront; the grid covers the batch
    if (b < Bt.batch)
        assemble_one<CPZ, CPG>(Mp[lmpc_model_of(Bt, b)], Bt, b, lane, stage, arena, glw(wsbase) + (size_t)b * M0.wsld);
}


// =====================================================================================
// assemble, MFMA form: 16 instances per workgroup of 4 wavefronts
// =====================================================================================
// Everything the generic kernel computes is affine in vin = [x0 | lastU | yref | 1] (the cost
// constant: a quadratic form), with maps tabulated at set-up.  With 16 instances as the N
// dimension each step is a small GEMM on the f64 MFMA pipe (v_mfma_f64_16x16x4_f64):
//     [f ; goff ; feasibility rows ; Qc vin] = MA * vin            (rowsA x kin) (kin x 16)
//     [t0 ; gt0]                             = (-Y[:, :nz]) * f    (ldy16 x nz16)(nz16 x 16)
// The D layout of a row tile (lane l, register r: row 4r + (l>>4), column l&15) is exactly the
// B-operand layout of k-step 4*tile + r, so results chain into the next product without any
// cross-lane movement: each lane parks its own registers in LDS and reads them back by k index.
// The four wavefronts of a workgroup split the row tiles and share the operands through LDS.

__global__ __launch_bounds__(256) void lmpc_assemble_mfma(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt,
                                                           double *wsbase, const int variant)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LmpcDev &M = *Mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int nx = M.nx, nu = M.nu, ny = M.ny;
    const int kin4 = M.kin >> 2, nz4 = M.nz16 >> 2;
    const int ldz = M.ldz, ldg = M.ldg, ldy = M.ldy;
    double *Bv = smem;                               // [kin4][64]   vin as MFMA B operands
    double *Bf = Bv + (size_t)kin4 * 64;             // [nz4][64]    f as MFMA B operands
    double *c0s = Bf + (size_t)nz4 * 64;             // [4][16]: one slot per wavefront and instance, added up in wave order
    unsigned *bad = reinterpret_cast<unsigned *>(c0s + 64);   // [16]
    const gdp MAp = gl(variant ? M.MA1p : M.MA0p), Ymp = GP(Ymp);
    const int ntile1 = M.rowsA >> 4, tg = M.nz16 >> 4, ts = tg + (M.mg16 >> 4), tq = ts + (M.ns16 >> 4);
    const int G1 = (kin4 + 3) >> 2, G2 = (nz4 + 3) >> 2;          // k-step groups per row tile of the packed maps
    constexpr int kAsmGB = 4;                                      // groups in flight per wavefront: sixteen k-steps
    auto ld4 = [](gdp p) -> v4d { return *reinterpret_cast<const v4d MPCX_GAS *>(p); };

    for (int b0 = blockIdx.x * 16; b0 < Bt.batch; b0 += gridDim.x * 16) {
        const int bj = b0 + j;
        const bool live = bj < Bt.batch;
        const int bc = live ? bj : Bt.batch - 1;
        // vin operands: k-step kb holds rows 4kb + kq of instance j
        for (int kb = wave; kb < kin4; kb += 4) {
            const int k = 4 * kb + kq;
            double v = 0.0;
            if (k < M.nxp) { if (k < nx) v = gl(Bt.x0)[(size_t)bc * nx + k]; }
            else if (k < M.nxp + M.nup) { const int c = k - M.nxp; if (c < nu) v = gl(Bt.u0)[(size_t)bc * nu + c]; }
            else if (k < M.ione) { const int c = k - M.nxp - M.nup; if (variant && c < ny) v = gl(Bt.yref)[(size_t)bc * Bt.yref_bs + c]; }
            else if (k == M.ione) v = 1.0;
            Bv[kb * 64 + lane] = v;
        }
        if (threadIdx.x < 16) bad[threadIdx.x] = 0u;
        __syncthreads();

        gdw wsj = glw(wsbase) + (size_t)bc * M.wsld;
        double c0p = 0.0;
        bool badl = false;
        for (int t = wave; t < ntile1; t += 4) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            // (operands from the packed copy of the map: 32 contiguous bytes per lane and load, four k-steps each -- lmpc_pack_mfma_tiles)
            for (int g0 = 0; g0 < G1; g0 += kAsmGB) {
                v4d a[kAsmGB];
#pragma unroll
                for (int g = 0; g < kAsmGB; ++g) if (g0 + g < G1) a[g] = ld4(MAp + (((size_t)t * G1 + g0 + g) * 64 + lane) * 4);
#pragma unroll
                for (int g = 0; g < kAsmGB; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kb = 4 * (g0 + g) + e;
                        if (kb < kin4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][e], Bv[kb * 64 + lane], acc, 0, 0, 0);
                    }
            }
            if (t < tg) {
                // linear term: keep as operand for the second product, and file it
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Bf[(4 * t + r) * 64 + lane] = acc[r];
                    const int row = 16 * t + 4 * r + kq;
                    if (live && row < ldz) wsj[row] = acc[r];
                }
            } else if (t < ts) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * (t - tg) + 4 * r + kq;
                    if (live && row < ldg) {
                        wsj[ldz + ldy + row] = GP(lg0)[row] - acc[r];
                        wsj[ldz + ldy + ldg + row] = GP(ug0)[row] - acc[r];
                    }
                }
            } else if (t < tq) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * (t - ts) + 4 * r + kq;
                    if (row < M.ns) badl |= violates(acc[r], GP(slo)[row], GP(shi)[row], M.eps_abs, M.eps_rel);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kb2 = 4 * (t - tq) + r;
                    if (kb2 < kin4) c0p = fma(0.5 * Bv[kb2 * 64 + lane], acc[r], c0p);
                }
            }
        }
        // per-instance reductions over the four k-quarters of the wave, then over the waves
        if (badl) atomicOr(&bad[j], 1u);
        __syncthreads();

        const int ntile2 = M.ldy16 >> 4;
        for (int t = wave; t < ntile2; t += 4) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int g0 = 0; g0 < G2; g0 += kAsmGB) {
                v4d a[kAsmGB];
#pragma unroll
                for (int g = 0; g < kAsmGB; ++g) if (g0 + g < G2) a[g] = ld4(Ymp + (((size_t)t * G2 + g0 + g) * 64 + lane) * 4);
#pragma unroll
                for (int g = 0; g < kAsmGB; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kb = 4 * (g0 + g) + e;
                        if (kb < nz4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][e], Bf[kb * 64 + lane], acc, 0, 0, 0);
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + kq;
                if (live && row < ldy) wsj[ldz + row] = acc[r];
            }
        }
        // this wavefront's share of the cost constant of instance j (no atomics: the sum must not depend on arrival order)
        c0p += __shfl_xor(c0p, 16, 64);
        c0p += __shfl_xor(c0p, 32, 64);
        if (kq == 0) c0s[wave * 16 + j] = c0p;
        __syncthreads();
        if (threadIdx.x < 16 && b0 + (int)threadIdx.x < Bt.batch) {
            gdw wst = glw(wsbase) + (size_t)(b0 + threadIdx.x) * M.wsld + ldz + ldy + 2 * ldg;
            wst[0] = ((c0s[threadIdx.x] + c0s[16 + threadIdx.x]) + c0s[32 + threadIdx.x]) + c0s[48 + threadIdx.x];
            wst[1] = bad[threadIdx.x] ? 1.0 : 0.0;
        }
        __syncthreads();
    }
}

// =====================================================================================
// solve: one instance per wavefront
// =====================================================================================
template <int CPZ, int CPG, bool ADMM, bool FUSED = false>
__device__ void solve_one(const LmpcDev &M, const LmpcBatchDev &Bt, const int b, const int lane,
                          double *stage, double *nt0, double *arena, gdw ws, const double *mf_lds = nullptr)
{
    constexpr int NZS = 2 * CPZ, NGS = 2 * CPG;
    const int nx = M.nx, nu = M.nu, ny = M.ny, ndu = M.ndu, ph = M.ph;
    const int nz = M.nz, mg = M.mg, ldz = M.ldz, ldg = M.ldg, ldy = M.ldy;
    const double INF = __builtin_huge_val();
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const gdp gY = GP(Y);

    long long tstamp[8];
    // profiling aid (tools/phase_cycles.py): cycles per phase of a polish round.  Compiled in with -DMPCX_PROFILE_ROUNDS only:
    // even switched off at run time the counters cost registers the kernel does not have (16 more spilled VGPRs).
#ifdef MPCX_PROFILE_ROUNDS
    long long pacc[4] = {0, 0, 0, 0}, plast = 0;
    auto plap = [&](int k) { if (Bt.dbg_cycles) { const long long now = (long long)__builtin_readcyclecounter(); pacc[k] += now - plast; plast = now; } };
#else
    auto plap = [](int) {};
#endif
    int tsi = 0;
    auto stamp = [&]() { if (Bt.dbg_cycles && tsi < 8) tstamp[tsi++] = (long long)__builtin_readcyclecounter(); };
    stamp();

    // ---- load the assembled problem: from the workspace record the assemble kernel left, or from the one this wavefront just
    // computed into its LDS slice (same layout)
    if constexpr (FUSED) {
        RecPtrs rp{arena, arena + ldz, arena + 2 * ldz, arena + ldz + ldy, arena + ldz + ldy + ldg, arena + ldz + ldy + 2 * ldg};
        fused_record(M, Bt, b, lane, stage, rp, mf_lds);
    }
    auto rec2 = [&](int at) -> d2 {
        if constexpr (FUSED) { const double2 v = *reinterpret_cast<const double2 *>(arena + at); d2 r; r.x = v.x; r.y = v.y; return r; }
        else return ld2(ws + at);
    };
    double f[NZS], lw[NZS], uw[NZS], rb[NZS], t0[NZS];
    bool eqb[NZS];
#pragma unroll
    for (int c = 0; c < CPZ; ++c) {
        const int e = 128 * c + 2 * lane;
        const int eo = e < ldz ? e : 0;
        const d2 vt = rec2(ldz + eo), vl = ld2(GP(lw) + eo), vu = ld2(GP(uw) + eo), vr = ld2(GP(rho_b) + eo);
        const bool ok = e < ldz;
        const d2 vf = rec2(eo);
        f[2 * c] = ok ? vf.x : 0.0; f[2 * c + 1] = ok ? vf.y : 0.0;
        t0[2 * c] = ok ? vt.x : 0.0; t0[2 * c + 1] = ok ? vt.y : 0.0;
        lw[2 * c] = ok ? vl.x : -INF; lw[2 * c + 1] = ok ? vl.y : -INF;
        uw[2 * c] = ok ? vu.x : INF; uw[2 * c + 1] = ok ? vu.y : INF;
        rb[2 * c] = ok ? vr.x : 0.0; rb[2 * c + 1] = ok ? vr.y : 0.0;
    }
    double lg[NGS], ug[NGS], rg[NGS], gt0[NGS];
    bool eqg[NGS];
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const int r = 128 * c + 2 * lane;
        const int ro = r < ldg ? r : 0;
        const d2 vt = rec2(2 * ldz + ro), vl = rec2(ldz + ldy + ro), vu = rec2(ldz + ldy + ldg + ro), vr = ld2(GP(rho_g) + ro);
        const d2 l0 = ld2(GP(lg0) + ro), u0 = ld2(GP(ug0) + ro);
        const bool ok = r < ldg;
        gt0[2 * c] = ok ? vt.x : 0.0; gt0[2 * c + 1] = ok ? vt.y : 0.0;
        lg[2 * c] = ok ? vl.x : -INF; lg[2 * c + 1] = ok ? vl.y : -INF;
        ug[2 * c] = ok ? vu.x : INF; ug[2 * c + 1] = ok ? vu.y : INF;
        rg[2 * c] = ok ? vr.x : 1.0; rg[2 * c + 1] = ok ? vr.y : 1.0;
        eqg[2 * c] = ok && (l0.x == u0.x); eqg[2 * c + 1] = ok && (l0.y == u0.y);
    }
#pragma unroll
    for (int s = 0; s < NZS; ++s) eqb[s] = (lw[s] == uw[s]);
    const d2 tail = rec2(ldz + ldy + 2 * ldg);
    const double c0 = tail.x;
    // A violated step-0 / input-independent row makes the QP infeasible.  The reference never
    // reports that (OSQP's certificate test yields NaN on libmpc++'s true infinities): it runs
    // out of iterations and returns an iterate that satisfies everything else, flagged
    // MAX_ITER_REACHED.  Default: the same outcome (solve without those rows, same flag);
    // strict mode: INFEASIBLE.
    const bool fixed_violation = tail.y == 1.0;
    const bool infeasible0 = fixed_violation && M.strict_infeasible;
    if (ADMM && tail.y == 2.0) return;          // already solved by the polish-only kernel
    if constexpr (FUSED) wave_sync();           // the record has been read: its LDS is the polish arena from here on
    stage_store<CPZ>(nt0, t0, ldz, lane);
    stage_store<CPG>(nt0 + ldz, gt0, ldg, lane);
    wave_sync();
    stamp();   // 1: loaded

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

    if (Bt.warm_lower) {
        // warm start: the first working set is the previous solve's active set (bits over the reference's rows,
        // ProblemBuilder.hpp:814-822: equalities | box on [x; x_u] | outputs | delta-u | scalar).  With warm_shift the
        // row looked up is the same constraint one step later: the previous tick's step i+1 is this tick's step i.
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
            if (e >= nz || eqb[s]) continue;
            int side = 0;
            for (int p = GP(boxrow_ptr)[e]; p < GP(boxrow_ptr)[e + 1]; ++p) {
                const int rr = look(GP(boxrow_ref)[p]);
                if (((wl[rr >> 5] >> (rr & 31)) & 1u) && GP(boxrow_lo)[p] == lw[s]) side = -1;
                if (((wu[rr >> 5] >> (rr & 31)) & 1u) && GP(boxrow_hi)[p] == uw[s]) side = 1;
            }
            actb[s] = side;
        }
#pragma unroll
        for (int s = 0; s < NGS; ++s) {
            const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
            if (r >= mg || eqg[s]) continue;
            const int rr = look(GP(g_refrow)[r]);
            actg[s] = ((wl[rr >> 5] >> (rr & 31)) & 1u) ? -1 : (((wu[rr >> 5] >> (rr & 31)) & 1u) ? 1 : 0);
        }
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
    int na_last = 0, rounds_total = 0;

    // -------- active-set polish with repair: true when the KKT conditions verify
    auto polish = [&](int rounds) -> bool {
        // working sets already visited in this call (hashed): a repeat means the repair rule is
        // cycling, which happens on a few instances in a thousand -- hand over to ADMM at once
        unsigned long long seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool safe = false;       // after a cycle: one exchange per round (most negative multiplier out,
                                 // else most violated row in), which does not cycle in practice
        for (int rd = 0; rd < rounds; ++rd) {
            ++rounds_total;
#ifdef MPCX_PROFILE_ROUNDS
            if (Bt.dbg_cycles) plast = (long long)__builtin_readcyclecounter();
#endif
            int na = 0;
            unsigned long long hsh = 0x9E3779B97F4A7C15ull;
#pragma unroll
            for (int s = 0; s < NZS; ++s) {
                const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
                const bool act = actb[s] != 0;
                const unsigned long long mk = __ballot(act);
                hsh = (hsh ^ mk) * 0xBF58476D1CE4E5B9ull;
                hsh = (hsh ^ __ballot(actb[s] > 0)) * 0x94D049BB133111EBull;
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
                hsh = (hsh ^ mk) * 0xBF58476D1CE4E5B9ull;
                hsh = (hsh ^ __ballot(actg[s] > 0)) * 0x94D049BB133111EBull;
                const int pos = na + __popcll(mk & lt_mask);
                posg[s] = pos;
                if (act && pos < kMaxActive) { wsidx[pos] = ldz + r; wsb[pos] = actg[s] < 0 ? lg[s] : ug[s]; }
                na += __popcll(mk);
            }
            if (na > kMaxActive) return false;
            hsh |= 1ull;
            bool cyc = false;
#pragma unroll
            for (int q = 0; q < 8; ++q) cyc |= (seen[q] == hsh);
            if (cyc) {
                if (safe) return false;
                safe = true;
#pragma unroll
                for (int q = 0; q < 8; ++q) seen[q] = 0;
            }
#pragma unroll
            for (int q = 7; q > 0; --q) seen[q] = seen[q - 1];
            seen[0] = hsh;
            na_last = na;
            wave_sync();
            plap(0);
            int dep_at = -1;
            if (na > 0) {
                // ---- Schur complement in registers: lane i owns row i; LDL' with the pivot column
                // broadcast by v_readlane (no LDS round trip on the dependent chain).  The code is
                // fully unrolled, so it is instantiated for three capacities and the smallest that
                // holds the working set runs (mean |A| is ~6: the 8-row version does most rounds).
                auto reg_polish = [&](auto capc) {
                    constexpr int CAPB = decltype(capc)::value;
                    // ---- Schur complement in registers: lane i owns row i; LDL' with the pivot
                    // column broadcast by v_readlane (no LDS round trip on the dependent chain)
                    const int qi = wsidx[lane < na ? lane : 0];
                    double Sr[CAPB];
#pragma unroll
                    for (int c = 0; c < CAPB; ++c) {
                        const int qc = wsidx[c < na ? c : 0];
                        Sr[c] = gY[(size_t)qi * ldy + qc];
                    }
                    double y = nt0[qi] - wsb[lane < na ? lane : 0];
                    double dgi = 1.0, mydinv = 1.0;
#pragma unroll
                    for (int c = 0; c < CAPB; ++c) if (lane == c) dgi = Sr[c];
#pragma unroll
                    for (int k = 0; k < CAPB; ++k) {
                        if (k < na && dep_at < 0) {
                            const double dk = readlane_d(Sr[k], k);
                            const double d0 = readlane_d(dgi, k);
                            if (!(dk > 1e-11 * d0)) {
                                dep_at = k;
                            } else {
                                const double rinv = pivot_rcp(dk);
                                if (lane == k) mydinv = rinv;
                                const double lik = Sr[k] * rinv;
                                // (columns past na only touch lanes past na: no guard, no branch)
#pragma unroll
                                for (int j = k + 1; j < CAPB; ++j) {
                                    const double tjk = readlane_d(Sr[k], j);
                                    if (lane >= j) Sr[j] = fma(-lik, tjk, Sr[j]);
                                }
                                if (lane > k) Sr[k] = lik;
                            }
                        }
                    }
                    if (dep_at < 0) {
#pragma unroll
                        for (int k = 0; k < CAPB; ++k) {
                            const double yk = readlane_d(y, k);
                            if (lane > k && lane < na) y = fma(-Sr[k], yk, y);
                        }
                        y *= mydinv;
                        // transpose L through LDS so that the back-substitution also walks registers
                        double *T = S;
#pragma unroll
                        for (int c = 0; c < CAPB; ++c)
                            if (c < lane && lane < na) T[lane * CAPB + c] = Sr[c];
                        wave_sync();
#pragma unroll
                        for (int k = 0; k < CAPB; ++k) {
                            const bool ok = k > lane && k < na;
                            const double v = T[(ok ? k : 0) * CAPB + (ok ? lane : 0)];
                            Sr[k] = ok ? v : 0.0;
                        }
#pragma unroll
                        for (int k = CAPB - 1; k >= 0; --k) {
                            if (k < na) {
                                const double xk = readlane_d(y, k);
                                if (lane < k) y = fma(-Sr[k], xk, y);
                            }
                        }
                        if (lane < na) lam[lane] = y;
                        wave_sync();
                    }
                };
                if (na <= 4) reg_polish(std::integral_constant<int, 4>{});
                else if (na <= 6) reg_polish(std::integral_constant<int, 6>{});
                else if (na <= 8) reg_polish(std::integral_constant<int, 8>{});
                else if (na <= 10) reg_polish(std::integral_constant<int, 10>{});
                else if (na <= 12) reg_polish(std::integral_constant<int, 12>{});      // the slow instances live here: |A| of 9..12
                else if (na <= kRegCap) reg_polish(std::integral_constant<int, kRegCap>{});
                else {
                for (int p = lane; p < na * na; p += 64) {
                    const int a = p / na, c = p - a * na;
                    S[a * kSld + c] = gY[(size_t)wsidx[a] * ldy + wsidx[c]];
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
            plap(1);
            // w = t0 - Y[:, A] lambda, all row fetches unpredicated
            int offz[CPZ], offg[CPG];
#pragma unroll
            for (int c = 0; c < CPZ; ++c) { const int e = 128 * c + 2 * lane; offz[c] = e < ldz ? e : 0; }
#pragma unroll
            for (int c = 0; c < CPG; ++c) { const int r = 128 * c + 2 * lane; offg[c] = ldz + (r < ldg ? r : 0); }
#pragma unroll
            for (int s = 0; s < NZS; ++s) wv[s] = t0[s];
#pragma unroll
            for (int s = 0; s < NGS; ++s) gw[s] = gt0[s];
            double lmax = 0;
            for (int a0 = 0; a0 < na; a0 += 4) {
                // four working-set rows of Y in flight at a time (rows past na re-read row a0, weight 0)
                double la[4];
                d2 mz[4][CPZ], mgv[4][CPG];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int a = a0 + u < na ? a0 + u : a0;
                    la[u] = a0 + u < na ? lam[a] : 0.0;
                    const gdp row = gY + (size_t)wsidx[a] * ldy;
#pragma unroll
                    for (int c = 0; c < CPZ; ++c) mz[u][c] = ld2(row + offz[c]);
#pragma unroll
                    for (int c = 0; c < CPG; ++c) mgv[u][c] = ld2(row + offg[c]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    lmax = fmax(lmax, fabs(la[u]));
#pragma unroll
                    for (int c = 0; c < CPZ; ++c) {
                        wv[2 * c] = fma(-la[u], mz[u][c].x, wv[2 * c]); wv[2 * c + 1] = fma(-la[u], mz[u][c].y, wv[2 * c + 1]);
                    }
#pragma unroll
                    for (int c = 0; c < CPG; ++c) {
                        gw[2 * c] = fma(-la[u], mgv[u][c].x, gw[2 * c]); gw[2 * c + 1] = fma(-la[u], mgv[u][c].y, gw[2 * c + 1]);
                    }
                }
            }
            plap(2);
            const double dtol = 1e-9 * lmax + 1e-300;
            dtol_last = dtol;
            bool nanv = false, changed = false;
#pragma unroll
            for (int s = 0; s < NZS; ++s) {
                const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
                if (e >= nz) wv[s] = 0.0;
                nanv |= !(wv[s] == wv[s]);
            }
#pragma unroll
            for (int s = 0; s < NGS; ++s) {
                const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
                if (r >= mg) gw[s] = 0.0;
                nanv |= !(gw[s] == gw[s]);
            }
            if (!safe) {
                // Repair rule: first shed every working-set row whose multiplier has the wrong sign;
                // only a working set with all signs right is grown by the violated rows.
                bool drop = false;
#pragma unroll
                for (int s = 0; s < NZS; ++s)
                    if (actb[s] != 0 && !eqb[s]) {
                        const double l = lam[posb[s]];
                        if ((actb[s] < 0 && l > dtol) || (actb[s] > 0 && l < -dtol)) { actb[s] = 0; drop = true; }
                    }
#pragma unroll
                for (int s = 0; s < NGS; ++s)
                    if (actg[s] != 0 && !eqg[s]) {
                        const double l = lam[posg[s]];
                        if ((actg[s] < 0 && l > dtol) || (actg[s] > 0 && l < -dtol)) { actg[s] = 0; drop = true; }
                    }
                changed = wave_any(drop);
                if (!changed) {
                    bool add = false;
                    double vb[NZS], vg[NGS], vm = 0.0;
#pragma unroll
                    for (int s = 0; s < NZS; ++s) {
                        vb[s] = 0.0;
                        if (actb[s] == 0) vb[s] = fmax(fmax(lw[s] - ptol * fmax(1.0, fabs(lw[s])) - wv[s], wv[s] - uw[s] - ptol * fmax(1.0, fabs(uw[s]))), 0.0);
                        vm = fmax(vm, vb[s]);
                    }
#pragma unroll
                    for (int s = 0; s < NGS; ++s) {
                        vg[s] = 0.0;
                        if (actg[s] == 0) vg[s] = fmax(fmax(lg[s] - ptol * fmax(1.0, fabs(lg[s])) - gw[s], gw[s] - ug[s] - ptol * fmax(1.0, fabs(ug[s]))), 0.0);
                        vm = fmax(vm, vg[s]);
                    }
                    const double thr = MPCX_ADD_THETA * wave_max(vm);
#pragma unroll
                    for (int s = 0; s < NZS; ++s)
                        if (vb[s] > 0.0 && vb[s] >= thr) { actb[s] = wv[s] < lw[s] ? -1 : 1; add = true; }
#pragma unroll
                    for (int s = 0; s < NGS; ++s)
                        if (vg[s] > 0.0 && vg[s] >= thr) { actg[s] = gw[s] < lg[s] ? -1 : 1; add = true; }
                    changed = wave_any(add);
                }
            } else {
                // single exchange
                double bestv = 0; int bests = -1;
#pragma unroll
                for (int s = 0; s < NZS; ++s)
                    if (actb[s] != 0 && !eqb[s]) {
                        const double l = lam[posb[s]];
                        const double bad = actb[s] < 0 ? l : -l;
                        if (bad > dtol && bad > bestv) { bestv = bad; bests = s; }
                    }
#pragma unroll
                for (int s = 0; s < NGS; ++s)
                    if (actg[s] != 0 && !eqg[s]) {
                        const double l = lam[posg[s]];
                        const double bad = actg[s] < 0 ? l : -l;
                        if (bad > dtol && bad > bestv) { bestv = bad; bests = NZS + s; }
                    }
                double mx = wave_max(bestv);
                if (mx > 0) {
                    const unsigned long long mk = __ballot(bestv == mx && bests >= 0);
                    if (lane == __ffsll((long long)mk) - 1) {
#pragma unroll
                        for (int s = 0; s < NZS; ++s) if (bests == s) actb[s] = 0;
#pragma unroll
                        for (int s = 0; s < NGS; ++s) if (bests == NZS + s) actg[s] = 0;
                    }
                    changed = true;
                } else {
                    int side = 0;
                    bestv = 0; bests = -1;
#pragma unroll
                    for (int s = 0; s < NZS; ++s) {
                        const int e = 128 * (s >> 1) + 2 * lane + (s & 1);
                        if (actb[s] == 0 && e < nz) {
                            const double vl = lw[s] - ptol * fmax(1.0, fabs(lw[s])) - wv[s];
                            const double vu = wv[s] - uw[s] - ptol * fmax(1.0, fabs(uw[s]));
                            const double v = fmax(vl, vu);
                            if (v > 0) {
                                const double nv = v * rsqrt(gY[(size_t)e * ldy + e]);
                                if (nv > bestv) { bestv = nv; bests = s; side = vl > vu ? -1 : 1; }
                            }
                        }
                    }
#pragma unroll
                    for (int s = 0; s < NGS; ++s) {
                        const int r = 128 * (s >> 1) + 2 * lane + (s & 1);
                        if (actg[s] == 0 && r < mg) {
                            const double vl = lg[s] - ptol * fmax(1.0, fabs(lg[s])) - gw[s];
                            const double vu = gw[s] - ug[s] - ptol * fmax(1.0, fabs(ug[s]));
                            const double v = fmax(vl, vu);
                            if (v > 0) {
                                const double nv = v * rsqrt(gY[(size_t)(ldz + r) * ldy + ldz + r]);
                                if (nv > bestv) { bestv = nv; bests = NZS + s; side = vl > vu ? -1 : 1; }
                            }
                        }
                    }
                    mx = wave_max(bestv);
                    if (mx > 0) {
                        const unsigned long long mk = __ballot(bestv == mx && bests >= 0);
                        if (lane == __ffsll((long long)mk) - 1) {
#pragma unroll
                            for (int s = 0; s < NZS; ++s) if (bests == s) actb[s] = side;
#pragma unroll
                            for (int s = 0; s < NGS; ++s) if (bests == NZS + s) actg[s] = side;
                        }
                        changed = true;
                    }
                }
            }
            plap(3);
            if (wave_any(nanv)) return false;
            if (!changed) return true;
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
        matvec_acc<CPZ>(GP(Gr), ldz, ldz, mg, stage, rhs, lane);
        wave_sync();
        stage_store<CPZ>(stage, rhs, ldz, lane);
        wave_sync();
        double xt[NZS];
#pragma unroll
        for (int s = 0; s < NZS; ++s) xt[s] = 0;
        matvec_acc<CPZ>(GP(Kinv), ldz, ldz, nz, stage, xt, lane);
        wave_sync();
        stage_store<CPZ>(stage, xt, ldz, lane);
        wave_sync();
        double ztg[NGS];
#pragma unroll
        for (int s = 0; s < NGS; ++s) ztg[s] = 0;
        matvec_acc<CPG>(GP(Gc), ldg, ldg, nz, stage, ztg, lane);
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
        matvec_acc<CPZ>(GP(Gr), ldz, ldz, mg, stage, pb, lane);
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
        matvec_acc<CPG>(GP(Gc), ldg, ldg, nz, stage, gx, lane);
        matvec_acc<CPZ>(GP(H), ldz, ldz, nz, stage, hx, lane);
        wave_sync();
        stage_store<CPG>(stage, yg, ldg, lane);
        wave_sync();
        matvec_acc<CPZ>(GP(Gr), ldz, ldz, mg, stage, aty, lane);
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

    // ---- solve
    int iters = 0;
    bool solved = false, polished = false, infeasible = infeasible0;
    int solver_status = -10;
    if (!infeasible) {
        if (M.polish && !ADMM) { solved = polish(M.polish_rounds0); polished = solved; }
        if (!ADMM && !solved) {                       // left for the fallback kernel (flag stays 0 / 1)
            if constexpr (FUSED) {
                // the fallback reads the workspace record: file the one this wavefront computed (a handful of instances in a thousand)
#pragma unroll
                for (int c = 0; c < CPZ; ++c) {
                    const int e = 128 * c + 2 * lane;
                    if (e < ldz) { st2(ws + e, f[2 * c], f[2 * c + 1]); st2(ws + ldz + e, t0[2 * c], t0[2 * c + 1]); }
                }
#pragma unroll
                for (int c = 0; c < CPG; ++c) {
                    const int r = 128 * c + 2 * lane;
                    if (r < ldg) {
                        st2(ws + ldz + ldz + r, gt0[2 * c], gt0[2 * c + 1]);
                        st2(ws + ldz + ldy + r, lg[2 * c], lg[2 * c + 1]);
                        st2(ws + ldz + ldy + ldg + r, ug[2 * c], ug[2 * c + 1]);
                    }
                }
                if (lane == 0) st2(ws + ldz + ldy + 2 * ldg, c0, tail.y);
            }
            return;
        }
        while (ADMM && !solved && iters < M.max_iter) {
            const int nblk = min(M.check_every, M.max_iter - iters);
            for (int k = 0; k < nblk; ++k) admm_iter();
            iters += nblk;
            if (certificate()) { infeasible = true; break; }       // see below for the non-strict outcome
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
        if (infeasible && !M.strict_infeasible) {
            // the reference would return whatever ADMM iterate max_iter leaves it with; return ours,
            // moved inside the input box so that cmd is at least admissible
            infeasible = false; solver_status = -2;
#pragma unroll
            for (int s = 0; s < NZS; ++s) x[s] = clampd(x[s], lw[s], uw[s]);
        }
        else if (infeasible) solver_status = -3;
        else if (solved) solver_status = fixed_violation ? -2 : 1;
        else {
            const int rs = residual_status();
            solver_status = fixed_violation ? -2 : (rs == 1 ? 1 : (rs == 2 ? 2 : -2));
        }
    } else {
        solver_status = -3;
    }
    stamp();   // 2: solved

    // ---- unpack (LOptimizer.hpp:305-347)
    double w[NZS];
#pragma unroll
    for (int s = 0; s < NZS; ++s) w[s] = polished ? wv[s] : x[s];
    const double qnan = __builtin_nan("");
    double cost;
    bool cost_pending = false;
    if (infeasible) {
#pragma unroll
        for (int s = 0; s < NZS; ++s) w[s] = qnan;
    }
    wave_sync();
    stage_store<CPZ>(stage, w, ldz, lane);      // stays staged for the sequence roll-out below
    wave_sync();
    if (infeasible) {
        cost = 1e30;
    } else if (!ADMM && polished && !M.cost_direct) {
        // At the verified point H w + f + N_A' lambda = 0 and N_A w = b_A, hence w'Hw/2 + f'w = (f'w - lambda'b_A)/2: no pass
        // over H (the single largest read of an instance: nz x nz doubles), and the products involve the bounded solution w,
        // not the possibly huge unconstrained optimum
        double j = 0;
#pragma unroll
        for (int s = 0; s < NZS; ++s) j = fma(f[s], w[s], j);
        if (lane < na_last) j = fma(-lam[lane], wsb[lane], j);
        cost = 0.5 * wave_sum(j) + c0;
    } else if (!ADMM && polished && !FUSED) {
        // cost from its definition (M.cost_direct: the identity above loses digits on an ill-conditioned Hessian), but not here:
        // H is nz x nz doubles per instance from L2 for a mat-vec, 320 KB at N = 50.  The solution goes to the workspace in
        // t0's place and lmpc_cost_mfma does H W for sixteen instances per pass over H on the matrix pipe.
#pragma unroll
        for (int c = 0; c < CPZ; ++c) {
            const int e = 128 * c + 2 * lane;
            if (e < ldz) st2(ws + ldz + e, w[2 * c], w[2 * c + 1]);
        }
        cost = 0.0;
        cost_pending = true;
    } else {
        // any other point (ADMM iterate, regularised Hessian): the definition
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
        if (e < nu) glw(Bt.cmd)[(size_t)b * nu + e] = w[s];
    }
    if (lane == 0) {
        if (Bt.cost && !cost_pending) glw(Bt.cost)[b] = cost;
        if (Bt.solver_status) glw(Bt.solver_status)[b] = solver_status;
        if (Bt.status) {
            // LOptimizer.hpp:386-415
            int st = 4;
            if (solver_status == 1 || solver_status == 2) st = 0;
            else if (solver_status == -2) st = 1;
            else if (solver_status == -3) st = 2;
            glw(Bt.status)[b] = st;
        }
        if (Bt.is_feasible) glw(Bt.is_feasible)[b] = (solver_status == 1 || solver_status == 2 || solver_status == -2) ? 1 : 0;
        if (Bt.iterations) glw(Bt.iterations)[b] = iters;
        if (Bt.polish_rounds) glw(Bt.polish_rounds)[b] = rounds_total;
        if (Bt.active_count) glw(Bt.active_count)[b] = polished ? na_last : 0;
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
                for (int p = GP(boxrow_ptr)[e]; p < GP(boxrow_ptr)[e + 1]; ++p) {
                    const int rr = GP(boxrow_ref)[p];
                    if (side < 0 && GP(boxrow_lo)[p] == lw[s]) atomicOr(&bl[rr >> 5], 1u << (rr & 31));
                    if (side > 0 && GP(boxrow_hi)[p] == uw[s]) atomicOr(&bu[rr >> 5], 1u << (rr & 31));
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
                const int rr = GP(g_refrow)[r];
                if (side < 0) atomicOr(&bl[rr >> 5], 1u << (rr & 31));
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
        double *xs0 = arena, *xs1 = arena + nx;      // ping-pong state
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
    if (!ADMM && lane == 0) ws[ldz + ldy + 2 * ldg + 1] = cost_pending ? 3.0 : 2.0;     // 2: done, the fallback kernel skips it; 3: lmpc_cost_mfma first
    stamp();   // 3: unpacked
    if (Bt.dbg_cycles && lane == 0)
#ifdef MPCX_PROFILE_ROUNDS
        for (int k = 0; k < 8; ++k) Bt.dbg_cycles[(size_t)b * 8 + k] = k < 4 ? (k < tsi ? tstamp[k] : 0) : pacc[k - 4];
#else
        for (int k = 0; k < 8; ++k) Bt.dbg_cycles[(size_t)b * 8 + k] = k < tsi ? tstamp[k] : 0;
#endif
}


// Fallback for the instances the polish-only kernel left unsolved (a handful in a thousand, or
// everything when polish is switched off): ADMM iterations, then polish again.
template <int CPZ, int CPG>
__global__ __launch_bounds__(kWavesPerBlock * 64) void lmpc_solve_admm(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt, double *wsbase)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    // the usual launch finds nothing to do: with the flags in an array of their own (lmpc_solve_group's), a wavefront whose first chunk is the only one it
    // has and is all done leaves before it has looked at the model struct at all -- one trip to memory instead of three in a row
    if (Bt.chunked && Bt.done) {
        const int c0 = (blockIdx.x * wpb + wave) * kFallbackChunk;
        if (c0 + (int)(gridDim.x * wpb * kFallbackChunk) >= Bt.batch) {
            const int bi = c0 + lane;
            if (__ballot(lane < kFallbackChunk && bi < Bt.batch && gl(Bt.done)[bi] == 0) == 0ull) return;
        }
    }
    const LmpcDev &M = *Mp;
    double *stage = smem + (size_t)wave * M.lds_per_wave;
    double *nt0 = stage + M.stage_len;
    double *arena = nt0 + M.ldy;
    if (Bt.chunked) {
        // after the polish-only kernel almost nothing is left: a wavefront looks at the flags of kFallbackChunk instances at
        // once (one load each, side by side) and only enters the solver for those still open -- an eighth of the wavefronts
        // to launch and retire
        for (int c0 = (blockIdx.x * wpb + wave) * kFallbackChunk; c0 < Bt.batch; c0 += gridDim.x * wpb * kFallbackChunk) {
            const int bi = c0 + lane;
            const bool open = lane < kFallbackChunk && bi < Bt.batch &&
                              (Bt.done ? gl(Bt.done)[bi] == 0 : glw(wsbase)[(size_t)bi * M.wsld + M.ldz + M.ldy + 2 * M.ldg + 1] != 2.0);
            unsigned long long todo = __ballot(open);
            while (todo) {
                const int b = c0 + (int)__builtin_ctzll(todo);
                todo &= todo - 1;
                solve_one<CPZ, CPG, true>(Mp[lmpc_model_of(Bt, b)], Bt, b, lane, stage, nt0, arena, glw(wsbase) + (size_t)b * M.wsld);
            }
        }
    } else {
        for (int b = blockIdx.x * wpb + wave; b < Bt.batch; b += gridDim.x * wpb)
            solve_one<CPZ, CPG, true>(Mp[lmpc_model_of(Bt, b)], Bt, b, lane, stage, nt0, arena, glw(wsbase) + (size_t)b * M.wsld);
    }
}


// cost = 0.5 w'Hw + f'w + c0 for the instances lmpc_solve marked "cost pending" (flag 3): H W on the f64 MFMA pipe, sixteen
// instances per workgroup (same operand chaining as lmpc_assemble_mfma), one pass over H per sixteen instances
__global__ __launch_bounds__(256) void lmpc_cost_mfma(const LmpcDev *__restrict__ Mp, const LmpcBatchDev Bt, double *wsbase)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LmpcDev &M = *Mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    const int ldz = M.ldz, ldy = M.ldy, ldg = M.ldg, nz4 = M.nz16 >> 2, ntile = M.nz16 >> 4;
    double *Bw = smem;                              // [nz4][64]
    double *cs = Bw + (size_t)nz4 * 64;             // [4 wavefronts][16 instances]
    const gdp H = GP(H), Hp = GP(Hp);
    const int G = (nz4 + 3) >> 2;
    constexpr int kCostGB = 4;
    for (int b0 = blockIdx.x * 16; b0 < Bt.batch; b0 += gridDim.x * 16) {
        const int bj = b0 + j;
        const bool live = bj < Bt.batch;
        const int bc = live ? bj : Bt.batch - 1;
        const gdp wsr = gl((const double *)wsbase) + (size_t)bc * M.wsld;
        const bool pending = live && wsr[ldz + ldy + 2 * ldg + 1] == 3.0;
        for (int kb = wave; kb < nz4; kb += 4) {
            const int k = 4 * kb + kq;
            Bw[kb * 64 + lane] = (pending && k < M.nz) ? wsr[ldz + k] : 0.0;
        }
        __syncthreads();
        double part = 0.0;
        for (int tl = wave; tl < ntile; tl += 4) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            if (Hp) {
                // operands from the packed, zero-padded copy of H: 32 contiguous bytes per lane and load, four k-steps each (the same products in the same order)
                for (int g0 = 0; g0 < G; g0 += kCostGB) {
                    v4d a[kCostGB];
#pragma unroll
                    for (int g = 0; g < kCostGB; ++g) if (g0 + g < G) a[g] = *reinterpret_cast<const v4d MPCX_GAS *>(Hp + (((size_t)tl * G + g0 + g) * 64 + lane) * 4);
#pragma unroll
                    for (int g = 0; g < kCostGB; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kb = 4 * (g0 + g) + e;
                            if (kb < nz4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][e], Bw[kb * 64 + lane], acc, 0, 0, 0);
                        }
                }
            } else {
                const int rowa = 16 * tl + j;
                const gdp Ht = H + (rowa < ldz ? rowa : 0);
                int kb = 0;
                for (; kb + 4 <= nz4; kb += 4) {
                    double a[4], bq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int col = 4 * (kb + u) + kq;
                        a[u] = (col < M.nz && rowa < ldz) ? Ht[(size_t)col * ldz] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) bq[u] = Bw[(kb + u) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bq[u], acc, 0, 0, 0);
                }
                for (; kb < nz4; ++kb) {
                    const int col = 4 * kb + kq;
                    const double a = (col < M.nz && rowa < ldz) ? Ht[(size_t)col * ldz] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bw[kb * 64 + lane], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * tl + 4 * r + kq;
                const double w = Bw[(4 * tl + r) * 64 + lane];
                const double fr = (pending && row < M.nz) ? wsr[row] : 0.0;
                part = fma(w, 0.5 * acc[r] + fr, part);
            }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (kq == 0) cs[wave * 16 + j] = part;      // one slot per wavefront, added up in a fixed order: the same bits every run
        __syncthreads();
        if (threadIdx.x < 16) {
            const int bb = b0 + threadIdx.x;
            if (bb < Bt.batch) {
                gdw wst = glw(wsbase) + (size_t)bb * M.wsld + ldz + ldy + 2 * ldg;
                if (wst[1] == 3.0) {
                    if (Bt.cost) glw(Bt.cost)[bb] = ((cs[threadIdx.x] + cs[16 + threadIdx.x]) + (cs[32 + threadIdx.x] + cs[48 + threadIdx.x])) + wst[0];
                    wst[1] = 2.0;
                }
            }
        }
        __syncthreads();
    }
}

template <int CPZ, int CPG>
int launch_variant(const LmpcDev &m, const LmpcDev *m_dev, const LmpcBatchDev &b, double *ws, hipStream_t stream, int which, int fast)
{
    const size_t lds = (size_t)kWavesPerBlock * m.lds_per_wave * sizeof(double);
    if (lds > lmpc_lds_limit()) return -2;
    auto k1 = lmpc_assemble_generic<CPZ, CPG>;
    auto k3 = lmpc_solve_admm<CPZ, CPG>;
    // the attribute is per device: remember what each device was given (an atomic per device, so that two host threads or
    // two handles on different GPUs cannot skip or tear the update)
    static std::atomic<size_t> configured[64];
    int devid = 0;
    (void)hipGetDevice(&devid);
    devid &= 63;
    if (lds > configured[devid].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        size_t prev = configured[devid].load(std::memory_order_relaxed);
        while (prev < lds && !configured[devid].compare_exchange_weak(prev, lds, std::memory_order_release)) {}
    }
    int blocks = (b.batch + kWavesPerBlock - 1) / kWavesPerBlock;
    const int cap = 256 * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    // (one-kernel forms: the fused / persistent mat-vec forms serve the one-chunk variant, the in-workgroup form -- b.fused >= 3 -- the two-chunk one too)
    const bool fused = b.fused != 0 && ((CPZ == 1 && CPG == 1) || (b.fused >= 3 && CPZ == 2 && CPG == 2));
    if ((which & 1) && !fused) {
        if (fast >= 0) {
            const size_t lds1 = ((size_t)(m.kin / 4 + m.nz16 / 4) * 64 + 64 + 32) * sizeof(double);
            int blocks1 = (b.batch + 15) / 16;
            if (blocks1 > 4096) blocks1 = 4096;
            hipLaunchKernelGGL(lmpc_assemble_mfma, dim3(blocks1), dim3(256), lds1, stream, m_dev, b, ws, fast);
        } else {
            hipLaunchKernelGGL(k1, dim3((b.batch + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kWavesPerBlock * 64), lds, stream, m_dev, b, ws);
        }
    }
    if (which & 2) {
        {
            LmpcBatchDev bf = b;
            if (!fused) bf.fused = 0;
            const int rf = lmpc_launch_fast(m, m_dev, bf, ws, stream);      // lean kernels (lmpc_fast.hip)
            if (rf != 0) return rf;
        }
        if (m.cost_direct && m.polish && !fused && b.n_models <= 0) {       // the costs lmpc_solve left pending
            const size_t ldsc = ((size_t)(m.nz16 / 4) * 64 + 64) * sizeof(double);
            int blocksq = (b.batch + 15) / 16;
            if (blocksq > 4096) blocksq = 4096;
            hipLaunchKernelGGL(lmpc_cost_mfma, dim3(blocksq), dim3(256), ldsc, stream, m_dev, b, ws);
        }
    }
    if (which & 4) {
        LmpcBatchDev b3 = b;
        int blocks3 = blocks;
        if (m.polish) {                          // a polish pass always comes first then: few instances are left
            b3.chunked = 1;
            blocks3 = (b.batch + kFallbackChunk * kWavesPerBlock - 1) / (kFallbackChunk * kWavesPerBlock);
            if (blocks3 > cap) blocks3 = cap;
            if (blocks3 < 1) blocks3 = 1;
        }
        hipLaunchKernelGGL(k3, dim3(blocks3), dim3(kWavesPerBlock * 64), lds, stream, m_dev, b3, ws);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

size_t lmpc_lds_limit()
{
    static std::atomic<size_t> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 160 * 1024;
    dev &= 63;
    size_t v = cached[dev].load(std::memory_order_acquire);
    if (v == 0) {
        int a = 0;
        v = (hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && a > 0) ? (size_t)a : (size_t)160 * 1024;
        cached[dev].store(v, std::memory_order_release);
    }
    return v;
}

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
    int a1 = (m.ph + 1) * (m.nx + 2 * m.ny) + (m.ph + 2) * m.nx + 2 * m.ph * m.nu + m.nu + (m.nx + m.nu) + m.ph * m.ndu + (m.ph + 2) / 2 + 8 +
             m.nx * m.nx + m.nx * m.nu + m.ny * m.nx + m.nx * m.ndu + m.ny * m.ndu;      // assemble_one's plan, model matrices last
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

int lmpc_launch(const LmpcDev &m, const LmpcDev *m_dev, const LmpcBatchDev &b, double *ws, void *stream, int which, int fast)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (lmpc_kernel_variant(m.ldz, m.ldg)) {
    case 1: return launch_variant<1, 1>(m, m_dev, b, ws, s, which, fast);
    case 2: return launch_variant<2, 2>(m, m_dev, b, ws, s, which, fast);
    case 4: return launch_variant<4, 4>(m, m_dev, b, ws, s, which, fast);
    default: return -2;
    }
}

}  // namespace mpcx
