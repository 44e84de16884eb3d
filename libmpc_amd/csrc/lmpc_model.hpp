// Host-side controller state and condensing for the batched LMPC path.
//
// LmpcController keeps exactly the state libmpc++'s linear front-end keeps in its
// ProblemBuilder (reference include/mpc/LMPC/ProblemBuilder.hpp:829-853) and
// LOptimizer (include/mpc/LMPC/LOptimizer.hpp:523-525): augmented model, per-step
// weights and bounds with the reference's "+1 column shift", scalar constraint,
// references, exogenous inputs, LParameters.  condense() turns it into the dense
// condensed QP in absolute inputs that the HIP kernel iterates on (SURVEY.md
// Appendix A "condensed equivalent").
#pragma once

#include <cmath>
#include <cstddef>
#include <limits>
#include <string>
#include <vector>

#include "../../include/mpcx.h"

namespace mpcx {

constexpr double kInf = std::numeric_limits<double>::infinity();

// column-major dense matrix, just enough for set-up
struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() = default;
    Mat(int r_, int c_, double v = 0.0) : r(r_), c(c_), a((size_t)r_ * c_, v) {}
    double &operator()(int i, int j) { return a[(size_t)i + (size_t)j * r]; }
    double operator()(int i, int j) const { return a[(size_t)i + (size_t)j * r]; }
    double *col(int j) { return a.data() + (size_t)j * r; }
    const double *col(int j) const { return a.data() + (size_t)j * r; }
};

Mat matmul(const Mat &A, const Mat &B);
Mat transpose(const Mat &A);
// in-place lower Cholesky of SPD A; returns smallest pivot (<=0 on failure)
double cholesky_lower(Mat &A);
// inverse of SPD matrix from its Cholesky factor
Mat spd_inverse_from_chol(const Mat &L);

enum GKind : int { G_STATE = 0, G_OUTPUT = 1, G_SCALAR = 2 };

struct GeneralRow {
    int kind, step, comp;   // which functional of the predicted trajectory
    int refrow;             // row in the reference QP's inequality block numbering + neq offset applied later
    double lo, hi;          // bounds before the per-instance free-response offset
};

struct BoxRef {            // a reference row that bounds condensed variable `var`
    int var, refrow;
    double lo, hi;
};

struct Condensed {
    int nf = 0, nz = 0, mg = 0, ldz = 0, ldg = 0, ldy = 0;
    int n_ref = 0, m_ref = 0, neq_ref = 0, active_words = 0;
    bool has_dist = false, h_regularised = false;
    double inverse_residual = 0;                 // || H Hinv - I ||_max of the computed inverse
    std::vector<double> H, Kinv, Gr, Gc, Y;      // padded, see lmpc_device.hpp
    size_t big_n[5] = {0, 0, 0, 0, 0};           // structure-only condensing: the lengths H, Kinv, Gr, Gc, Y would have (they stay empty)
    std::vector<double> lw, uw, rho_b;           // [ldz]
    std::vector<double> lg0, ug0, rho_g;         // [ldg]
    std::vector<int> g_kind, g_step, g_comp, g_refrow;      // [ldg]
    std::vector<GeneralRow> fixed_rows;          // rows that do not depend on the decision variables
    std::vector<int> boxrow_ptr, boxrow_ref;     // CSR var -> reference rows
    std::vector<double> boxrow_lo, boxrow_hi;
    std::vector<int> blk;                        // [ph+1] condensed block of v_i (blk[0] unused)
    double flops_setup = 0;

    // ---- stacked affine maps for the MFMA assemble kernel (lmpc_kernels.hip) -------------
    // input vector vin = [x0 (nxp) | u0 (nup) | yref (nyp) | 1,0,0,0], every block padded to a
    // multiple of 4 (one f64 MFMA k-step); MA rows = [f (nz16) | goff (mg16) | feasibility rows
    // (ns16) | Qc vin (kq16)], every block padded to a multiple of 16 (one MFMA row tile).
    int kin = 0, nxp = 0, nup = 0, nyp = 0, ione = 0;
    int nz16 = 0, mg16 = 0, ns = 0, ns16 = 0, kq16 = 0, rowsA = 0, ldy16 = 0;
    std::vector<double> MA[2];                   // [0]: shared yref, [1]: per-instance constant yref; rowsA x kin, column-major
    std::vector<double> slo, shi;                // [ns16] bounds of the feasibility rows
    std::vector<double> Ym;                      // -Y[:, 0:nz], ldy16 x nz16 column-major (tile padded)
    // ---- the same maps composed for the fused solve kernel: everything lmpc_solve needs of an instance is MF * vin, rows
    // [t0 ; gt0 (ldy) | goff (ldg) | f (ldz) | feasibility rows (nsp) | Qc vin (kin)], rowsF x kin column-major
    int rowsF = 0, nsp = 0;
    std::vector<double> MF[2];
};

struct AsmOut {
    std::vector<double> f, goff, sval;
    double c0 = 0;
};

struct LmpcController {
    mpcx_dims d{};
    int na = 0;
    // model
    Mat A, B, C, Bd, Dd;
    bool have_model = false;
    // ProblemBuilder state (internal columns 0..ph, see ProblemBuilder.hpp:254-260)
    Mat wOutput, wU, wDeltaU;
    Mat minX, maxX, minY, maxY, minU, maxU;
    std::vector<double> sMin, sMax, sX, sU;
    // LOptimizer state
    Mat yRef, uRef, duRef, dMeas;
    mpcx_lparams prm{};

    explicit LmpcController(const mpcx_dims &dims);

    bool pred_slice_valid(int start, int end) const;   // IMPC.hpp:251-260
    bool ctrl_slice_valid(int start, int end) const;   // IMPC.hpp:269-278

    // matrix-form setters (ProblemBuilder.hpp:247-263, 378-432)
    void set_objective(const double *OW, const double *UW, const double *DUW);
    void set_objective_idx(int idx, const double *ow, const double *uw, const double *duw);
    void set_state_bounds(const double *lo, const double *hi);
    void set_state_bounds_idx(int idx, const double *lo, const double *hi);
    void set_input_bounds(const double *lo, const double *hi);
    void set_input_bounds_idx(int idx, const double *lo, const double *hi);
    void set_output_bounds(const double *lo, const double *hi);
    void set_output_bounds_idx(int idx, const double *lo, const double *hi);
    void set_scalar_vec(const double *smin, const double *smax, const double *X, const double *U);
    void set_scalar_idx(int idx, double smin, double smax, const double *X, const double *U);

    // returns empty string on success, message otherwise
    // like != nullptr: the structure and the O(n) arrays only, rows classified as in `like` (see lmpc_model.cpp)
    std::string condense(Condensed &out, const Condensed *like = nullptr) const;
    // host evaluation of what the generic assemble kernel computes for one instance
    void assemble_host(const Condensed &o, const double *x0, const double *u0, const Mat &yR, const Mat &uR,
                       const Mat &dR, const Mat &dM, AsmOut &out) const;
    void build_fast_maps(Condensed &o) const;
    // the same after a change of references / exogenous inputs only: the constant column (and the linear part of the cost
    // form) of both maps; the quadratic part does not depend on them
    void refresh_fast_maps(Condensed &o) const;
    void compose_fused_maps(Condensed &o) const;
};

}  // namespace mpcx
