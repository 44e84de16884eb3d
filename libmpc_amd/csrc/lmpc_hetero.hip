// Device-side condensing for heterogeneous LMPC batches (mpcx_lmpc_hetero_*): what ProblemBuilder::buildTimeInvariantTems
// (reference include/mpc/LMPC/ProblemBuilder.hpp:642-825) and the solver set-up do per controller object, for K controllers at
// once -- one workgroup of four wavefronts per controller, the matrices in LDS, every GEMM-shaped product on
// v_mfma_f64_16x16x4_f64:
//     S_i = A S_{i-1} + B E_i            prediction matrices ("A^h, B-stack"), rolled over the horizon, never stored whole
//     H   = sum_i (C S_i)' W_i (C S_i) + weights on the inputs and their increments                       (MFMA, nz x ny x nz per step)
//     G   = the rows of S_i / C S_i / sX' S_i the finite bounds select                                     (written as they appear)
//     H = L L',  Hinv = L^-T L^-1        Cholesky in LDS, triangular inverse one column per thread, product on MFMA
//     Y   = [Hinv, (G Hinv)'; G Hinv, G Hinv G']                                                          (MFMA)
//     K   = H + sigma I + diag(rho_b) + G' diag(rho_g) G,  Kinv                                           (MFMA, Cholesky, MFMA)
// The O(n) parts of a controller (bounds, references, row maps) are laid out by the host; this kernel fills the O(n^3) arrays of
// every model's device struct in place.  Same arithmetic as LmpcController::condense (lmpc_model.cpp), which remains the set-up of
// a single controller and the check of this kernel (tests/test_lmpc_hetero.py compares the two banks).
#include "lmpc_kernel_common.hpp"

namespace mpcx {

namespace {

constexpr int kCondWaves = 4;

__device__ __forceinline__ v4d mfma4(double a, double b, v4d acc) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); }

// D tiles of a product on the matrix pipe: tile (tm, tn) of the mt x nt tile grid goes to wavefront (tm * nt + tn) % nwaves.
// fa(row, k), fb(k, col) fetch operands (zero outside the matrices), fd(row, col, value) files a result element.
template <class FA, class FB, class FD>
__device__ __forceinline__ void mfma_product(const int mt, const int nt, const int k4, FA fa, FB fb, FD fd, const int wave, const int lane)
{
    const int j = lane & 15, kq = lane >> 4;
    for (int t = wave; t < mt * nt; t += kCondWaves) {
        const int tm = t / nt, tn = t - tm * nt;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        int kb = 0;
        for (; kb + 4 <= k4; kb += 4) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = fa(16 * tm + j, 4 * (kb + u) + kq); b[u] = fb(4 * (kb + u) + kq, 16 * tn + j); }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = mfma4(a[u], b[u], acc);
        }
        for (; kb < k4; ++kb) acc = mfma4(fa(16 * tm + j, 4 * kb + kq), fb(4 * kb + kq, 16 * tn + j), acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) fd(16 * tm + 4 * r + kq, 16 * tn + j, acc[r]);
    }
}

// in-place lower Cholesky of the n x n matrix P (row-major, leading dimension ld) by the whole workgroup; false if a pivot is not
// larger than tol (the matrix is then unusable)
__device__ bool chol_lds(double *P, const int n, const int ld, const double tol, int *flag)
{
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    if (t == 0) *flag = 0;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const double d = P[k * ld + k];
        if (!(d > tol)) { if (t == 0) *flag = 1; }
        __syncthreads();
        if (*flag) return false;
        const double sd = sqrt(d), isd = 1.0 / sd;
        for (int i = k + 1 + t; i < n; i += blockDim.x) P[i * ld + k] *= isd;
        __syncthreads();
        if (t == 0) P[k * ld + k] = sd;
        for (int i = k + 1 + ty; i < n; i += 16) {
            const double lik = P[i * ld + k];
            for (int jj = k + 1 + tx; jj <= i; jj += 16) P[i * ld + jj] = fma(-lik, P[jj * ld + k], P[i * ld + jj]);
        }
        __syncthreads();
    }
    return true;
}

// Q = L^-1 (lower triangular, the rest zero), one column per thread
__device__ void tri_inverse_lds(const double *L, double *Q, const int n, const int ld)
{
    for (int e = threadIdx.x; e < n * ld; e += blockDim.x) Q[e] = 0.0;
    __syncthreads();
    for (int jj = threadIdx.x; jj < n; jj += blockDim.x) {
        Q[jj * ld + jj] = 1.0 / L[jj * ld + jj];
        for (int i = jj + 1; i < n; ++i) {
            double s = 0.0;
            for (int p = jj; p < i; ++p) s = fma(L[i * ld + p], Q[p * ld + jj], s);
            Q[i * ld + jj] = -s / L[i * ld + i];
        }
    }
    __syncthreads();
}

// LDS plan (doubles): P [NP x ld] | Q [NQ x ld] | Sx [2][nx x ld] | CS [nyr x ld] | small vectors
// BIG (more than 96 condensed variables -- config 4's shape, N = 50: P and Q alone are 2 x 340 KB): P and Q live in a scratch block of the
// workgroup in global memory (L2-resident; the same code walks them through flat addresses), the Hessian's tiles are added to P step by
// step instead of being carried in registers; everything else as below.  A set-up step: it is bound by the latency of the factorisations'
// dependent passes, not by bytes -- and still two orders of magnitude ahead of condensing K controllers on the host's cores.
template <bool BIG>
__global__ __launch_bounds__(kCondWaves * 64) void lmpc_condense_models(LmpcDev *models, const int count, const int NP, const int NQ, double *scratch, const size_t scratch_stride)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int flag, cstat;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = threadIdx.x;
    for (int mdl = blockIdx.x; mdl < count; mdl += gridDim.x) {
        LmpcDev &M = models[mdl];
        if (threadIdx.x == 0) cstat = 0;
        __syncthreads();
        const int nx = M.nx, nu = M.nu, ny = M.ny, ph = M.ph, nz = M.nz, mg = M.mg, ldz = M.ldz, ldg = M.ldg, ldy = M.ldy;
        const int ld = NP + 1;
        double *P = BIG ? scratch + (size_t)blockIdx.x * scratch_stride : smem, *Q = P + (size_t)NP * ld;
        double *Sx0 = BIG ? smem : Q + (size_t)NQ * ld, *Sx1 = Sx0 + (size_t)nx * ld, *CS = Sx1 + (size_t)nx * ld;
        double *wv = CS + (size_t)(ny > 1 ? ny : 1) * ld;                      // ny: the step's output weights
        double *rb = wv + ((ny + 3) & ~3) + 4, *rg = rb + NP;                  // rho_b [NP], rho_g [NQ]
        const gdp gA = gl(M.A), gB = gl(M.B), gC = gl(M.C), gWy = gl(M.Wy), gWu = gl(M.Wu), gWdu = gl(M.Wdu), gsX = gl(M.sX), gsU = gl(M.sU);
        const gip blk = gl(M.blk), gk = gl(M.g_kind), gs = gl(M.g_step), gc = gl(M.g_comp);
        gdw oH = glw(const_cast<double *>(M.H)), oK = glw(const_cast<double *>(M.Kinv)), oGr = glw(const_cast<double *>(M.Gr)),
            oGc = glw(const_cast<double *>(M.Gc)), oY = glw(const_cast<double *>(M.Y)), orb = glw(const_cast<double *>(M.rho_b)),
            org = glw(const_cast<double *>(M.rho_g));
        const int npt = NP >> 4;

        // ---- H: roll the prediction matrices over the horizon; this wavefront's upper-triangle tiles accumulate in registers
        constexpr int MAXS = 6;                                                 // tiles per wavefront: NP <= 96 -> 21 tiles / 4
        v4d hacc[MAXS];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) hacc[s] = v4d{0.0, 0.0, 0.0, 0.0};
        for (int e = t; e < nx * ld; e += blockDim.x) Sx0[e] = 0.0;
        if constexpr (BIG) for (int e = t; e < NP * ld; e += blockDim.x) P[e] = 0.0;
        __syncthreads();
        double *Sp = Sx0, *Sn = Sx1;
        for (int i = 1; i <= ph; ++i) {
            // S_i = A S_{i-1} + B E_i  (E_i selects block blk[i] of the decision vector)
            for (int e = t; e < nx * NP; e += blockDim.x) {
                const int a = e / NP, q = e - a * NP;
                double s = 0.0;
                if (q < nz) {
                    for (int c = 0; c < nx; ++c) s = fma(gA[a + c * nx], Sp[c * ld + q], s);
                    const int bq = q / nu, jq = q - bq * nu;
                    if (bq == blk[i]) s += gB[a + jq * nx];
                }
                Sn[a * ld + q] = s;
            }
            __syncthreads();
            for (int e = t; e < ny * NP; e += blockDim.x) {
                const int a = e / NP, q = e - a * NP;
                double s = 0.0;
                if (q < nz) for (int c = 0; c < nx; ++c) s = fma(gC[a + c * ny], Sn[c * ld + q], s);
                CS[a * ld + q] = s;
            }
            if (t < ny) wv[t] = gWy[i * ny + t];
            __syncthreads();
            // rows of G this step contributes (state / output / scalar rows, ProblemBuilder.hpp:735-809)
            for (int e = t; e < mg * nz; e += blockDim.x) {
                const int r = e / nz, q = e - r * nz;
                if (gs[r] != i) continue;
                const int kind = gk[r], cp = gc[r];
                double v;
                if (kind == 0) v = Sn[cp * ld + q];
                else if (kind == 1) v = CS[cp * ld + q];
                else {
                    v = 0.0;
                    for (int c = 0; c < nx; ++c) v = fma(gsX[c], Sn[c * ld + q], v);
                    const int bq = q / nu, jq = q - bq * nu;
                    if (bq == blk[i]) v += gsU[jq];
                }
                oGr[(size_t)r * ldz + q] = v;
                oGc[(size_t)q * ldg + r] = v;
            }
            // H += (C S_i)' W_i (C S_i) on the upper-triangle tiles
            {
                const int j = lane & 15, kq = lane >> 4, k4 = (ny + 3) >> 2;
                int s = 0, tcount = 0;
                for (int tm = 0; tm < npt; ++tm)
                    for (int tn = tm; tn < npt; ++tn, ++tcount) {
                        if ((tcount & (kCondWaves - 1)) != wave) continue;
                        v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
                        for (int kb = 0; kb < k4; ++kb) {
                            const int a = 4 * kb + kq;
                            const double av = a < ny ? CS[a * ld + 16 * tm + j] * wv[a] : 0.0;
                            const double bv = a < ny ? CS[a * ld + 16 * tn + j] : 0.0;
                            acc = mfma4(av, bv, acc);
                        }
                        if constexpr (BIG) {
                            // (a tile belongs to one wavefront, an element of it to one lane: no two threads add to the same place)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * tm + 4 * r + kq, col = 16 * tn + j;
                                if (row < nz && col < nz && row <= col) P[row * ld + col] += acc[r];
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < MAXS; ++u) if (u == s) hacc[u] += acc;
                        }
                        ++s;
                    }
            }
            __syncthreads();
            double *tmp = Sp; Sp = Sn; Sn = tmp;
        }
        if constexpr (BIG) {
            // the lower triangle from the upper one
            for (int e = t; e < nz * nz; e += blockDim.x) { const int row = e / nz, col = e - row * nz; if (row < col) P[col * ld + row] = P[row * ld + col]; }
        } else {
        // the accumulated tiles to LDS (both triangles)
        for (int e = t; e < NP * ld; e += blockDim.x) P[e] = 0.0;
        __syncthreads();
        {
            const int j = lane & 15, kq = lane >> 4;
            int s = 0, tcount = 0;
            for (int tm = 0; tm < npt; ++tm)
                for (int tn = tm; tn < npt; ++tn, ++tcount) {
                    if ((tcount & (kCondWaves - 1)) != wave) continue;
                    v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int u = 0; u < MAXS; ++u) if (u == s) acc = hacc[u];
                    ++s;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * tm + 4 * r + kq, col = 16 * tn + j;
                        if (row < nz && col < nz && row <= col) { P[row * ld + col] = acc[r]; P[col * ld + row] = acc[r]; }
                    }
                }
        }
        }
        __syncthreads();
        // weights on the inputs and on their increments (LmpcController::condense)
        if (t == 0) {
            for (int i = 1; i <= ph; ++i)
                for (int jj = 0; jj < nu; ++jj) P[(blk[i] * nu + jj) * ld + blk[i] * nu + jj] += gWu[i * nu + jj];
            for (int i = 0; i < ph; ++i) {
                const int bn = blk[i + 1];
                for (int jj = 0; jj < nu; ++jj) {
                    const double w = gWdu[i * nu + jj];
                    if (i == 0) { P[(bn * nu + jj) * ld + bn * nu + jj] += w; continue; }
                    const int bp = blk[i];
                    if (bp == bn) continue;
                    P[(bn * nu + jj) * ld + bn * nu + jj] += w;
                    P[(bp * nu + jj) * ld + bp * nu + jj] += w;
                    P[(bp * nu + jj) * ld + bn * nu + jj] -= w;
                    P[(bn * nu + jj) * ld + bp * nu + jj] -= w;
                }
            }
        }
        __syncthreads();
        double maxd = 0.0;
        for (int q = 0; q < nz; ++q) maxd = fmax(maxd, fabs(P[q * ld + q]));
        for (int e = t; e < nz * nz; e += blockDim.x) { const int q = e / nz, p = e - q * nz; oH[(size_t)q * ldz + p] = P[p * ld + q]; }
        __syncthreads();

        // ---- Hinv
        bool regularised = false;
        if (!chol_lds(P, nz, ld, 1e-13 * fmax(1.0, maxd), &flag)) {
            // a singular Hessian (zero weights): regularise, as the host set-up does
            regularised = true;
            const double delta = 1e-8 * fmax(1.0, maxd);
            __syncthreads();
            for (int e = t; e < nz * nz; e += blockDim.x) { const int q = e / nz, p = e - q * nz; P[p * ld + q] = oH[(size_t)q * ldz + p] + (p == q ? delta : 0.0); }
            __syncthreads();
            if (!chol_lds(P, nz, ld, 0.0, &flag)) {
                // LmpcController::condense on the host: "condensed Hessian is not positive semidefinite".  Recorded for the caller; the
                // factor is replaced by the identity so that what follows stays finite (the model is never solved: create fails)
                if (t == 0) cstat |= 1;
                __syncthreads();
                for (int e = t; e < nz * nz; e += blockDim.x) { const int q = e / nz, p = e - q * nz; P[p * ld + q] = p == q ? 1.0 : 0.0; }
                __syncthreads();
            }
        }
        tri_inverse_lds(P, Q, nz, ld);
        // Hinv = Q' Q into P and into Y's leading block
        mfma_product(npt, npt, (nz + 3) >> 2,
                     [&](int row, int k) { return (row < nz && k < nz) ? Q[k * ld + row] : 0.0; },
                     [&](int k, int col) { return (col < nz && k < nz) ? Q[k * ld + col] : 0.0; },
                     [&](int row, int col, double v) { if (row < nz && col < nz) P[row * ld + col] = v; }, wave, lane);
        __syncthreads();
        for (int e = t; e < nz * nz; e += blockDim.x) { const int p = e / nz, q = e - p * nz; oY[(size_t)p * ldy + q] = P[p * ld + q]; }
        // || H Hinv - I ||_max decides whether the optimal cost may come from the multipliers (LmpcDev::cost_direct)
        {
            double worst = 0.0;
            for (int e = t; e < nz * nz; e += blockDim.x) {
                const int i = e / nz, jj = e - i * nz;
                double acc = i == jj ? -1.0 : 0.0;
                for (int k = 0; k < nz; ++k) acc = fma(oH[(size_t)k * ldz + i], P[k * ld + jj], acc);
                worst = fmax(worst, fabs(acc));
            }
            // (the freshly written H is read back through L2: same workgroup, after the barrier above)
            worst = wave_max(worst);
            if (lane == 0) rg[wave] = worst;
            __syncthreads();
            if (t == 0) M.cost_direct = (regularised || fmax(fmax(rg[0], rg[1]), fmax(rg[2], rg[3])) > 1e-9) ? 1 : 0;
            __syncthreads();
        }
        // ---- G Hinv (into Q) and G Hinv G'
        const int mpt = (mg + 15) >> 4;
        mfma_product(mpt, npt, (nz + 3) >> 2,
                     [&](int row, int k) { return (row < mg && k < nz) ? oGr[(size_t)row * ldz + k] : 0.0; },
                     [&](int k, int col) { return (col < nz && k < nz) ? P[k * ld + col] : 0.0; },
                     [&](int row, int col, double v) {
                         if (row < mg && col < nz) { Q[row * ld + col] = v; oY[(size_t)(ldz + row) * ldy + col] = v; oY[(size_t)col * ldy + ldz + row] = v; }
                     }, wave, lane);
        __syncthreads();
        mfma_product(mpt, mpt, (nz + 3) >> 2,
                     [&](int row, int k) { return (row < mg && k < nz) ? Q[row * ld + k] : 0.0; },
                     [&](int k, int col) { return (col < mg && k < nz) ? oGr[(size_t)col * ldz + k] : 0.0; },
                     [&](int row, int col, double v) {
                         if (row < mg && col < mg) { oY[(size_t)(ldz + row) * ldy + ldz + col] = v; if (row == col) rg[row] = v; }
                     }, wave, lane);
        __syncthreads();
        // ---- ADMM step sizes (Jacobi preconditioning of the dual problem; equality rows 1e3 x, as OSQP does)
        {
            const double kappa = 2.0, rho_min = 1e-6, rho_max = 1e6;
            for (int q = t; q < nz; q += blockDim.x) {
                const double lo = gl(M.lw)[q], hi = gl(M.uw)[q];
                double r = 0.0;
                if (lo > -__builtin_huge_val() || hi < __builtin_huge_val()) {
                    r = M.adaptive_rho ? kappa / fmax(P[q * ld + q], 1e-300) : M.rho_user;
                    if (lo == hi) r *= 1e3;
                    r = fmin(fmax(r, rho_min), rho_max);
                }
                rb[q] = r; orb[q] = r;
            }
            for (int r0 = t; r0 < mg; r0 += blockDim.x) {
                double r = M.adaptive_rho ? kappa / fmax(rg[r0], 1e-300) : M.rho_user;
                if (gl(M.lg0)[r0] == gl(M.ug0)[r0]) r *= 1e3;
                r = fmin(fmax(r, rho_min), rho_max);
                rg[r0] = r; org[r0] = r;
            }
        }
        __syncthreads();
        // ---- K = H + sigma I + diag(rho_b) + G' diag(rho_g) G  (into P), Kinv
        mfma_product(npt, npt, (mg + 3) >> 2,
                     [&](int row, int k) { return (row < nz && k < mg) ? oGr[(size_t)k * ldz + row] * rg[k] : 0.0; },
                     [&](int k, int col) { return (col < nz && k < mg) ? oGr[(size_t)k * ldz + col] : 0.0; },
                     [&](int row, int col, double v) {
                         if (row < nz && col < nz) P[row * ld + col] = v + oH[(size_t)col * ldz + row] + (row == col ? M.sigma + rb[row] : 0.0);
                     }, wave, lane);
        __syncthreads();
        if (!chol_lds(P, nz, ld, 0.0, &flag)) {                  // host: "ADMM matrix is not positive definite"
            if (t == 0) cstat |= 2;
            __syncthreads();
            for (int e = t; e < nz * nz; e += blockDim.x) { const int q = e / nz, p = e - q * nz; P[p * ld + q] = p == q ? 1.0 : 0.0; }
            __syncthreads();
        }
        tri_inverse_lds(P, Q, nz, ld);
        mfma_product(npt, npt, (nz + 3) >> 2,
                     [&](int row, int k) { return (row < nz && k < nz) ? Q[k * ld + row] : 0.0; },
                     [&](int k, int col) { return (col < nz && k < nz) ? Q[k * ld + col] : 0.0; },
                     [&](int row, int col, double v) { if (row < nz && col < nz) oK[(size_t)col * ldz + row] = v; }, wave, lane);
        __syncthreads();
        // a kept row of G that is identically zero: on the host that row would have been a feasibility-only (fixed) row -- the split into
        // fixed and general rows was taken from controller 0, so this controller's constraint structure differs from it
        for (int r0 = t; r0 < mg; r0 += blockDim.x) {
            double amax = 0.0;
            for (int q = 0; q < nz; ++q) amax = fmax(amax, fabs(oGr[(size_t)r0 * ldz + q]));
            if (!(amax > 0.0)) atomicOr(&cstat, 4);
        }
        __syncthreads();
        if (t == 0) M.cond_status = cstat;
        __syncthreads();
    }
}

}  // namespace

// LDS the condensing kernel needs for these dimensions (bytes); > 160 KB: the bank condenses on the host instead.  More than 96 condensed
// variables (the register-held Hessian tiles: at most 21 upper-triangle tiles over four wavefronts) or P and Q beyond the LDS: the BIG form,
// P and Q in a scratch block per workgroup (big_out: its doubles, 0 for the LDS form).
size_t lmpc_condense_lds(const LmpcDev &m, int *NP_out, int *NQ_out, size_t *big_out)
{
    const int NP = (m.nz + 15) / 16 * 16, MP = (m.mg + 15) / 16 * 16, NQ = NP > MP ? NP : MP, ld = NP + 1;
    if (NP_out) *NP_out = NP;
    if (NQ_out) *NQ_out = NQ;
    const size_t rest = 2 * (size_t)m.nx * ld + (size_t)(m.ny > 1 ? m.ny : 1) * ld + ((m.ny + 3) & ~3) + 4 + NP + NQ + 8;
    const size_t pq = (size_t)NP * ld + (size_t)NQ * ld;
    const bool big = NP > 96 || (pq + rest) * sizeof(double) > lmpc_lds_limit() - 64;
    if (big_out) *big_out = big ? pq : 0;
    return ((big ? 0 : pq) + rest) * sizeof(double);
}

int lmpc_condense_launch(LmpcDev *models_d, const LmpcDev &m0, int count, void *stream)
{
    int NP = 0, NQ = 0;
    size_t big = 0;
    const size_t lds = lmpc_condense_lds(m0, &NP, &NQ, &big);
    if (lds > lmpc_lds_limit() - 64) return -2;
    static std::atomic<size_t> conf[2][64];
    int devid = 0;
    (void)hipGetDevice(&devid);
    devid &= 63;
    const void *fn = big ? reinterpret_cast<const void *>(lmpc_condense_models<true>) : reinterpret_cast<const void *>(lmpc_condense_models<false>);
    if (lds > conf[big ? 1 : 0][devid].load(std::memory_order_acquire)) {
        // (the kernel also has four bytes of static LDS: ask for what is needed, not for the whole CU)
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
        conf[big ? 1 : 0][devid].store(lds, std::memory_order_release);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!big) {
        const int blocks = count < 1024 ? count : 1024;
        hipLaunchKernelGGL(lmpc_condense_models<false>, dim3(blocks), dim3(kCondWaves * 64), lds, s, models_d, count, NP, NQ, nullptr, 0);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    // the BIG form: a scratch block per workgroup, two workgroups per CU's worth of them; released when the kernel has finished
    // (a set-up path: the bank is being created, the caller synchronises right behind this call.  On a device short of memory the scratch shrinks --
    // fewer workgroups walk the bank -- before the launch gives up)
    int blocks = count < 512 ? count : 512;
    double *scratch = nullptr;
    while (hipMalloc(reinterpret_cast<void **>(&scratch), (size_t)blocks * big * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        scratch = nullptr;
        if (blocks == 1) return -4;
        blocks = (blocks + 1) / 2;
    }
    hipLaunchKernelGGL(lmpc_condense_models<true>, dim3(blocks), dim3(kCondWaves * 64), lds, s, models_d, count, NP, NQ, scratch, big);
    const bool ok = hipGetLastError() == hipSuccess;
    const hipError_t es = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    return ok && es == hipSuccess ? 0 : -3;
}

}  // namespace mpcx
