// POD views of the device-resident controller handed to the HIP kernels.
//
// HBM layout (all doubles, zero padded; every leading dimension is even so a lane
// can fetch an element pair with one 16-byte load):
//   H, Kinv : nz columns x ldz rows, column-major (both symmetric)
//   Gr      : row r of G contiguous   [ldg x ldz]  -> G' v  (lanes own elements of nz)
//   Gc      : column j of G contiguous [ldz x ldg] -> G x   (lanes own rows of G)
//   Y       : [ldy x ldy], ldy = ldz + ldg, the dual Hessian N Hinv N' with N = [I; G];
//             unified index q: box variable e -> q = e, general row r -> q = ldz + r
// A vector of length n is held by a wavefront as element pairs: lane l owns elements
// 128*c + 2*l and 128*c + 2*l + 1 for c < CP (CP = ceil(n / 128)).
#pragma once

#include <cstdint>

namespace mpcx {

constexpr int kMaxActive = 28;          // working-set capacity of the in-kernel polish
constexpr int kSld = kMaxActive + 1;    // LDS row stride of the Schur complement

struct LmpcDev {
    int nx, nu, ndu, ny, ph, ch, nf, nz, mg;
    int ldz, ldg, ldy;
    int m_ref, neq_ref, active_words;
    int has_dist, n_fixed;
    // solver parameters
    int max_iter, polish, check_every, polish_rounds0, polish_rounds;
    int cost_direct;                             // 1: cost from its definition (regularised Hessian), 0: from the multipliers
    int cond_status;                             // device-side condensing (lmpc_condense_models): 0 fine; bit 0: the condensed Hessian is not positive
                                                 // semidefinite; bit 1: the ADMM matrix is not positive definite; bit 2: a kept row of G is identically
                                                 // zero (the controller's constraint structure differs from controller 0's)
    int strict_infeasible;                       // 1: report INFEASIBLE / NaN; 0: behave as the reference does
    double alpha, sigma, eps_abs, eps_rel, eps_prim_inf;
    int adaptive_rho; double rho_user;            // LParameters::adaptive_rho / rho (the ADMM step sizes of the set-up)
    // per-wave LDS carve (in doubles)
    int stage_len, arena_len, lds_per_wave;
    int fast_slice;                              // per-wave LDS slice of the lean solve kernels (doubles)
    int wsld;                                    // per-instance workspace record (doubles): f | t0 | gt0 | lg | ug | c0, flag
    // model, column-major
    const double *A, *B, *C, *Bd, *Dd;
    const double *Wy, *Wu, *Wdu;                 // [(ph+1) x ny], [(ph+1) x nu], [ph x nu]; column = internal step
    const double *yref_s, *uref_s, *duref_s, *dmeas_s;   // shared references [ph x n]
    // step-0 feasibility rows
    const double *lo0x, *hi0x, *lo0u, *hi0u, *lo0y, *hi0y, *sX, *sU;
    double s0lo, s0hi;
    // condensed QP
    const double *H, *Kinv, *Gr, *Gc, *Y;
    const double *lw, *uw, *rho_b;               // [ldz]
    const double *lg0, *ug0, *rho_g;             // [ldg]
    const int *g_kind, *g_step, *g_comp, *g_refrow;      // [ldg]
    const int *f_kind, *f_step, *f_comp; const double *f_lo, *f_hi;   // fixed rows [n_fixed]
    const int *boxrow_ptr, *boxrow_ref; const double *boxrow_lo, *boxrow_hi;
    const int *blk;                              // [ph+1]
    // stacked maps of the MFMA assemble kernel (see Condensed in lmpc_model.hpp)
    int kin, nxp, nup, nyp, ione, nz16, mg16, ns, ns16, kq16, rowsA, ldy16;
    const double *MA0, *MA1, *Ym, *slo, *shi;
    // the same maps as lmpc_solve_group takes them (lmpc_pack_mfma_tiles): a lane's A operands of four consecutive k-steps side by side, a wavefront's of
    // one row tile and k-step group 2 KB in a row -- the phase is bound by the vector memory pipe's instruction rate, not by bytes
    const double *MA0p, *MA1p, *Ymp;
    const double *Hp;                            // H the same way, nz16 x nz16 zero padded (lmpc_cost_mfma's operand; null unless cost_direct)
    // composed maps of the fused solve kernel: rows [t0; gt0 (ldy) | goff (ldg) | f (ldz) | feasibility rows (nsp) | Qc vin (kin)]
    int rowsF, nsp, fused_ok, group_ok;
    const double *MF0, *MF1;
};

struct LmpcBatchDev {
    int batch;
    const double *x0, *u0;
    // reference accessors: value(b, k, a) = p[b*bs + k*ks + a]
    const double *yref; long yref_bs, yref_ks;
    const double *uref; long uref_bs, uref_ks;
    const double *duref; long duref_bs, duref_ks;
    const double *dmeas; long dmeas_bs, dmeas_ks;
    double *cmd, *cost;
    int32_t *status, *solver_status, *is_feasible, *iterations;
    uint32_t *active_lower, *active_upper;
    double *seq_state, *seq_output, *seq_input;
    int32_t *polish_rounds, *active_count;
    const uint32_t *warm_lower, *warm_upper;      // optional previous active sets (reference row numbering)
    int warm_shift;
    int chunked;                                  // fallback kernel: one wavefront screens a chunk of instances
    int fused;                                    // 0: record from the workspace; 1 / 2: lmpc_solve_fused with MF0 / MF1; 3 / 4: lmpc_solve_group with MA0 / MA1
    int32_t *done;                                // lmpc_solve_group: [B] 2 = solved there, 0 = left to the fallback kernel (null: the flag in the workspace record)
    int *pcounter;                                // work counter of the persistent fused kernel (null: one instance per launched wavefront)
    // heterogeneous batch (mpcx_lmpc_hetero_*): the kernels' model pointer is an array of n_models structs of identical dimensions and
    // constraint structure, instance b uses entry model_index[b] (null: entry b); 0 models = the one shared controller
    int n_models;
    const int32_t *model_index;
    long long *dbg_cycles;       // optional [B x 8] per-phase cycle counts (profiling aid)
};
// which entry of the model array instance b uses
__host__ __device__ inline int lmpc_model_of(const LmpcBatchDev &Bt, int b) { return Bt.n_models <= 0 ? 0 : (Bt.model_index ? Bt.model_index[b] : b); }

// LDS a workgroup of the current device may take (one CU's: gfx950 160 KB), asked of the runtime once per device -- implemented in lmpc_kernels.hip
size_t lmpc_lds_limit();
// implemented in lmpc_kernels.hip
int lmpc_kernel_variant(int ldz, int ldg);     // -1 if the dimensions are not covered
// which: bit 0 = assemble, bit 1 = polish-only solve, bit 2 = ADMM fallback (7 = the normal path;
// single bits are for per-kernel timing).  fast_variant: -1 = generic assemble kernel, 0/1 = MFMA
// assemble kernel with shared / per-instance-constant output reference.
int lmpc_launch(const LmpcDev &m, const LmpcDev *m_dev, const LmpcBatchDev &b, double *ws, void *stream,
                int which = 7, int fast_variant = -1);
int lmpc_lds_per_wave(const LmpcDev &m, int *stage_len, int *arena_len);
// src: rows x K column-major (rows a multiple of 16, K of 4) -> out[((t G + g) 64 + lane) 4 + e] = src[(4 (4 g + e) + kq) rows + 16 t + j] with lane = 16 kq + j,
// G = ceil(K / 16) k-step groups per row tile t (zero beyond K): what one wavefront's MFMA A operands of four k-steps look like in registers
void lmpc_pack_mfma_tiles(const double *src, int rows, int K, double *out);
inline size_t lmpc_packed_len(int rows, int K) { return (size_t)(rows / 16) * ((K + 15) / 16) * 256; }
int lmpc_fast_slice(const LmpcDev &m);          // needs wsld, kin, nx
size_t lmpc_group_lds_bytes(const LmpcDev &m);  // LDS block of lmpc_solve_group (0: no group form for the variant); needs fast_slice, kin, nz16, nu
// implemented in lmpc_fast.hip: the lean polish kernel (b.fused: the fused / persistent forms) on `stream`
int lmpc_launch_fast(const LmpcDev &m, const LmpcDev *m_dev, const LmpcBatchDev &b, double *ws, void *stream);
// implemented in lmpc_hetero.hip: the O(n^3) arrays of `count` model structs (device array) computed in place, one workgroup each;
// -2: the dimensions do not fit the kernel's LDS plan (the bank then condenses on the host)
size_t lmpc_condense_lds(const LmpcDev &m, int *NP_out, int *NQ_out, size_t *big_out = nullptr);
int lmpc_condense_launch(LmpcDev *models_d, const LmpcDev &m0, int count, void *stream);

}  // namespace mpcx
