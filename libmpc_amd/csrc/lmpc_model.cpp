// Host-side controller state and condensing.  See lmpc_model.hpp.
#include "lmpc_model.hpp"
#include "lmpc_device.hpp"

#include <algorithm>
#include <cstring>

namespace mpcx {

Mat matmul(const Mat &A, const Mat &B)
{
    Mat Cm(A.r, B.c);
    for (int j = 0; j < B.c; j++)
        for (int k = 0; k < A.c; k++) {
            double b = B(k, j);
            if (b == 0.0) continue;
            const double *ak = A.col(k);
            double *cj = Cm.col(j);
            for (int i = 0; i < A.r; i++) cj[i] += ak[i] * b;
        }
    return Cm;
}

Mat transpose(const Mat &A)
{
    Mat T(A.c, A.r);
    for (int j = 0; j < A.c; j++)
        for (int i = 0; i < A.r; i++) T(j, i) = A(i, j);
    return T;
}

double cholesky_lower(Mat &A)
{
    int n = A.r;
    double minpiv = kInf;
    for (int k = 0; k < n; k++) {
        double d = A(k, k);
        for (int p = 0; p < k; p++) d -= A(k, p) * A(k, p);
        if (!(d > 0.0)) return -1.0;
        minpiv = std::min(minpiv, d);
        double sd = std::sqrt(d);
        A(k, k) = sd;
        for (int i = k + 1; i < n; i++) {
            double s = A(i, k);
            for (int p = 0; p < k; p++) s -= A(i, p) * A(k, p);
            A(i, k) = s / sd;
        }
    }
    for (int j = 1; j < n; j++)
        for (int i = 0; i < j; i++) A(i, j) = 0.0;
    return minpiv;
}

Mat spd_inverse_from_chol(const Mat &L)
{
    int n = L.r;
    // Linv (lower), then inv = Linv' * Linv
    Mat Li(n, n);
    for (int j = 0; j < n; j++) {
        Li(j, j) = 1.0 / L(j, j);
        for (int i = j + 1; i < n; i++) {
            double s = 0;
            for (int p = j; p < i; p++) s += L(i, p) * Li(p, j);
            Li(i, j) = -s / L(i, i);
        }
    }
    Mat inv(n, n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = 0;
            for (int p = i; p < n; p++) s += Li(p, i) * Li(p, j);
            inv(i, j) = s;
            inv(j, i) = s;
        }
    return inv;
}

LmpcController::LmpcController(const mpcx_dims &dims) : d(dims)
{
    na = d.nx + d.nu;
    A = Mat(d.nx, d.nx); B = Mat(d.nx, d.nu); C = Mat(d.ny, d.nx);
    Bd = Mat(d.nx, d.ndu); Dd = Mat(d.ny, d.ndu);
    wOutput = Mat(d.ny, d.ph + 1); wU = Mat(d.nu, d.ph + 1); wDeltaU = Mat(d.nu, d.ph);
    minX = Mat(d.nx, d.ph + 1, -kInf); maxX = Mat(d.nx, d.ph + 1, kInf);
    minY = Mat(d.ny, d.ph + 1, -kInf); maxY = Mat(d.ny, d.ph + 1, kInf);
    minU = Mat(d.nu, d.ph, -kInf); maxU = Mat(d.nu, d.ph, kInf);
    sMin.assign(d.ph + 1, -kInf); sMax.assign(d.ph + 1, kInf);
    sX.assign(d.nx, 0.0); sU.assign(d.nu, 0.0);
    yRef = Mat(d.ny, d.ph); uRef = Mat(d.nu, d.ph); duRef = Mat(d.nu, d.ph); dMeas = Mat(d.ndu, d.ph);
    mpcx_lparams_default(&prm);
}

bool LmpcController::pred_slice_valid(int start, int end) const
{
    return !(start >= end || start > d.ph || end > d.ph || start < 0);
}
bool LmpcController::ctrl_slice_valid(int start, int end) const
{
    return !(start >= end || start > d.ch || end > d.ch || start < 0);
}

static void shift_in(Mat &dst, const double *src, int rows, int ph)
{
    // user column k -> internal column k+1; internal column 0 := user column 0
    for (int k = 0; k < ph; k++) std::memcpy(dst.col(k + 1), src + (size_t)k * rows, sizeof(double) * rows);
    std::memcpy(dst.col(0), src, sizeof(double) * rows);
}
static void put_idx(Mat &dst, int idx, const double *v, int rows)
{
    std::memcpy(dst.col(idx + 1), v, sizeof(double) * rows);
    if (idx == 0) std::memcpy(dst.col(0), v, sizeof(double) * rows);
}

void LmpcController::set_objective(const double *OW, const double *UW, const double *DUW)
{
    shift_in(wOutput, OW, d.ny, d.ph);
    shift_in(wU, UW, d.nu, d.ph);
    std::memcpy(wDeltaU.a.data(), DUW, sizeof(double) * d.nu * d.ph);
}
void LmpcController::set_objective_idx(int idx, const double *ow, const double *uw, const double *duw)
{
    put_idx(wOutput, idx, ow, d.ny);
    put_idx(wU, idx, uw, d.nu);
    std::memcpy(wDeltaU.col(idx), duw, sizeof(double) * d.nu);
}
void LmpcController::set_state_bounds(const double *lo, const double *hi)
{
    shift_in(minX, lo, d.nx, d.ph); shift_in(maxX, hi, d.nx, d.ph);
}
void LmpcController::set_state_bounds_idx(int idx, const double *lo, const double *hi)
{
    put_idx(minX, idx, lo, d.nx); put_idx(maxX, idx, hi, d.nx);
}
void LmpcController::set_output_bounds(const double *lo, const double *hi)
{
    shift_in(minY, lo, d.ny, d.ph); shift_in(maxY, hi, d.ny, d.ph);
}
void LmpcController::set_output_bounds_idx(int idx, const double *lo, const double *hi)
{
    put_idx(minY, idx, lo, d.ny); put_idx(maxY, idx, hi, d.ny);
}
void LmpcController::set_input_bounds(const double *lo, const double *hi)
{
    // nu x ch in; the last control-horizon column fills the rest (ProblemBuilder.hpp:402-410)
    std::memcpy(minU.a.data(), lo, sizeof(double) * d.nu * d.ch);
    std::memcpy(maxU.a.data(), hi, sizeof(double) * d.nu * d.ch);
    for (int k = d.ch; k < d.ph; k++) {
        std::memcpy(minU.col(k), lo + (size_t)(d.ch - 1) * d.nu, sizeof(double) * d.nu);
        std::memcpy(maxU.col(k), hi + (size_t)(d.ch - 1) * d.nu, sizeof(double) * d.nu);
    }
}
void LmpcController::set_input_bounds_idx(int idx, const double *lo, const double *hi)
{
    std::memcpy(minU.col(idx), lo, sizeof(double) * d.nu);
    std::memcpy(maxU.col(idx), hi, sizeof(double) * d.nu);
}
void LmpcController::set_scalar_vec(const double *smin, const double *smax, const double *X, const double *U)
{
    for (int k = 0; k < d.ph; k++) { sMin[k + 1] = smin[k]; sMax[k + 1] = smax[k]; }
    sMin[0] = smin[0]; sMax[0] = smax[0];
    sX.assign(X, X + d.nx); sU.assign(U, U + d.nu);
}
void LmpcController::set_scalar_idx(int idx, double smin, double smax, const double *X, const double *U)
{
    sMin[idx + 1] = smin; sMax[idx + 1] = smax;
    if (idx == 0) { sMin[0] = smin; sMax[0] = smax; }
    sX.assign(X, X + d.nx); sU.assign(U, U + d.nu);   // the multiplier is shared by all steps
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

std::string LmpcController::condense(Condensed &o, const Condensed *like) const
{
    // like != nullptr: structure only (mpcx_lmpc_hetero_*: the O(n^3) arrays are computed on the device, lmpc_hetero.hip) -- the rows
    // are classified as in `like` (a controller of the same bank, condensed in full), the matrices are allocated and left zero
    const bool light = like != nullptr;
    const int nx = d.nx, nu = d.nu, ny = d.ny, ndu = d.ndu, ph = d.ph, ch = d.ch;
    if (!have_model) return "state-space model not set";
    o = Condensed();
    o.n_ref = (ph + 1) * na + ph * nu;
    o.neq_ref = (ph + 1) * na;
    const int nineq = (ph + 1) * na + (ph + 1) * ny + ph * nu + (ph + 1);
    o.m_ref = o.neq_ref + nineq;
    o.active_words = (o.m_ref + 31) / 32;
    const int off_y = (ph + 1) * na, off_s = off_y + (ph + 1) * ny + ph * nu;

    // delta-u is free for steps 0..ch inclusive (ProblemBuilder.hpp:782-793): v_1..v_nf free
    const int nf = std::min(ph, ch + 1);
    const int nz = nf * nu;
    o.nf = nf; o.nz = nz;
    o.blk.assign(ph + 1, 0);
    for (int i = 1; i <= ph; i++) o.blk[i] = std::min(i, nf) - 1;
    const int ldz = std::max(2, round_up(nz, 2));
    o.ldz = ldz;

    o.has_dist = false;
    for (double v : Bd.a) if (v != 0.0) o.has_dist = true;
    for (double v : Dd.a) if (v != 0.0) o.has_dist = true;

    // prediction: Sx_i = sum_{j<=i} A^{i-j} B E_j   (nx x nz), i = 1..ph  -- the "A^h, B-stack"
    std::vector<Mat> Sx(light ? 0 : ph + 1, Mat(nx, nz));
    if (!light) {
        // recursion Sx_i = A Sx_{i-1} + B E_i
        for (int i = 1; i <= ph; i++) {
            if (i > 1) Sx[i] = matmul(A, Sx[i - 1]);
            for (int j = 0; j < nu; j++)
                for (int a = 0; a < nx; a++) Sx[i](a, o.blk[i] * nu + j) += B(a, j);
        }
        o.flops_setup += 2.0 * ph * nx * nx * nz;
    }

    // Hessian
    Mat H(nz, nz);
    for (int i = 1; i <= ph && !light; i++) {
        Mat CS = matmul(C, Sx[i]);                       // ny x nz
        for (int q = 0; q < nz; q++)
            for (int p = 0; p <= q; p++) {
                double s = 0;
                for (int a = 0; a < ny; a++) s += CS(a, p) * wOutput(a, i) * CS(a, q);
                H(p, q) += s;
            }
        for (int j = 0; j < nu; j++) H(o.blk[i] * nu + j, o.blk[i] * nu + j) += wU(j, i);
        o.flops_setup += 2.0 * ny * nx * nz + 1.0 * ny * nz * nz;
    }
    for (int i = 0; i < ph; i++) {
        // delta_i = v_{i+1} - v_i (v_0 = u0 is data)
        int bn = o.blk[i + 1];
        for (int j = 0; j < nu; j++) {
            double w = wDeltaU(j, i);
            if (i == 0) { H(bn * nu + j, bn * nu + j) += w; continue; }
            int bp = o.blk[i];
            if (bp == bn) continue;
            H(bn * nu + j, bn * nu + j) += w;
            H(bp * nu + j, bp * nu + j) += w;
            int lo = std::min(bp, bn) * nu + j, hi = std::max(bp, bn) * nu + j;
            H(lo, hi) -= w;
        }
    }
    for (int q = 0; q < nz; q++)
        for (int p = 0; p < q; p++) H(q, p) = H(p, q);

    // rows
    std::vector<double> lw(ldz, -kInf), uw(ldz, kInf);
    std::vector<BoxRef> boxrefs;
    std::vector<GeneralRow> rows;
    std::vector<std::vector<double>> grow;     // coefficient rows of G
    auto finite_any = [](double lo, double hi) { return std::isfinite(lo) || std::isfinite(hi); };
    for (int i = 1; i <= ph; i++) {
        int k = std::min(i, ph - 1);           // x_u(i) takes input-bound column min(i, ph-1) (:735-749)
        for (int j = 0; j < nx; j++) {
            double lo = minX(j, i), hi = maxX(j, i);
            if (!finite_any(lo, hi)) continue;
            std::vector<double> g(nz);
            for (int q = 0; q < nz && !light; q++) g[q] = Sx[i](j, q);
            rows.push_back({G_STATE, i, j, o.neq_ref + i * na + j, lo, hi});
            grow.push_back(std::move(g));
        }
        for (int j = 0; j < nu; j++) {
            double lo = minU(j, k), hi = maxU(j, k);
            if (!finite_any(lo, hi)) continue;
            int var = o.blk[i] * nu + j;
            boxrefs.push_back({var, o.neq_ref + i * na + nx + j, lo, hi});
            lw[var] = std::max(lw[var], lo);
            uw[var] = std::min(uw[var], hi);
        }
        for (int j = 0; j < ny; j++) {
            double lo = minY(j, i), hi = maxY(j, i);
            if (!finite_any(lo, hi)) continue;
            std::vector<double> g(nz, 0.0);
            for (int q = 0; q < nz && !light; q++) {
                double s = 0;
                for (int a = 0; a < nx; a++) s += C(j, a) * Sx[i](a, q);
                g[q] = s;
            }
            rows.push_back({G_OUTPUT, i, j, o.neq_ref + off_y + i * ny + j, lo, hi});
            grow.push_back(std::move(g));
        }
        if (finite_any(sMin[i], sMax[i])) {
            std::vector<double> g(nz, 0.0);
            for (int q = 0; q < nz && !light; q++) {
                double s = 0;
                for (int a = 0; a < nx; a++) s += sX[a] * Sx[i](a, q);
                g[q] = s;
            }
            for (int j = 0; j < nu; j++) g[o.blk[i] * nu + j] += sU[j];
            rows.push_back({G_SCALAR, i, 0, o.neq_ref + off_s + i, sMin[i], sMax[i]});
            grow.push_back(std::move(g));
        }
    }
    // rows that do not see the decision variables are pure feasibility conditions
    {
        std::vector<GeneralRow> keep; std::vector<std::vector<double>> gk;
        for (size_t r = 0; r < rows.size(); r++) {
            double nrm = 0;
            for (double v : grow[r]) nrm = std::max(nrm, std::fabs(v));
            if (light) {
                nrm = 1.0;
                for (const auto &fr : like->fixed_rows)
                    if (fr.kind == rows[r].kind && fr.step == rows[r].step && fr.comp == rows[r].comp) nrm = 0.0;
            }
            if (nrm < 1e-14) o.fixed_rows.push_back(rows[r]);
            else { keep.push_back(rows[r]); gk.push_back(std::move(grow[r])); }
        }
        rows.swap(keep); grow.swap(gk);
    }
    const int mg = (int)rows.size();
    o.mg = mg;
    const int ldg = std::max(2, round_up(mg, 2));
    o.ldg = ldg;
    o.ldy = ldz + ldg;

    if (light) {
        // everything the device fills: allocated, zero (padding stays zero)
        const int ldy = o.ldy;
        // (sizes only: the bank reserves them in its device-only region, nothing is staged on the host)
        o.big_n[0] = o.big_n[1] = (size_t)ldz * ldz; o.big_n[2] = (size_t)ldg * ldz; o.big_n[3] = (size_t)ldz * ldg; o.big_n[4] = (size_t)ldy * ldy;
        o.lw = lw; o.uw = uw;
        o.rho_b.assign(ldz, 0.0);
        o.lg0.assign(ldg, -kInf); o.ug0.assign(ldg, kInf); o.rho_g.assign(ldg, 1.0);
        o.g_kind.assign(ldg, 0); o.g_step.assign(ldg, 0); o.g_comp.assign(ldg, 0); o.g_refrow.assign(ldg, -1);
        for (int r = 0; r < mg; r++) {
            o.lg0[r] = rows[r].lo; o.ug0[r] = rows[r].hi;
            o.g_kind[r] = rows[r].kind; o.g_step[r] = rows[r].step; o.g_comp[r] = rows[r].comp; o.g_refrow[r] = rows[r].refrow;
        }
    }
    // regularise a singular Hessian (zero weights): no unique optimum then anyway
    double maxd = 0;
    if (!light) {
    for (int q = 0; q < nz; q++) maxd = std::max(maxd, std::fabs(H(q, q)));
    Mat L = H;
    double piv = cholesky_lower(L);
    if (!(piv > 1e-13 * std::max(1.0, maxd))) {
        o.h_regularised = true;
        double delta = 1e-8 * std::max(1.0, maxd);
        L = H;
        for (int q = 0; q < nz; q++) L(q, q) += delta;
        piv = cholesky_lower(L);
        if (!(piv > 0)) return "condensed Hessian is not positive semidefinite";
    }
    Mat Hinv = spd_inverse_from_chol(L);
    o.flops_setup += 2.0 * nz * nz * nz / 3.0 + 2.0 * nz * nz * nz;
    // how well the computed inverse inverts: || H Hinv - I ||_max.  The solve kernel may take the optimal cost from the
    // stationarity identity (no pass over H) only while this is negligible; long horizons (N = 50: cond(H) ~ 8e8) are not
    {
        double worst = 0;
        for (int i = 0; i < nz; i++)
            for (int j = 0; j < nz; j++) {
                double acc = i == j ? -1.0 : 0.0;
                for (int k = 0; k < nz; k++) acc += H(i, k) * Hinv(k, j);
                worst = std::max(worst, std::fabs(acc));
            }
        o.inverse_residual = worst;
        o.flops_setup += 2.0 * nz * nz * nz;
    }

    // dual Hessian Y = N Hinv N', N = [I; G]
    Mat G(std::max(mg, 1), nz);
    for (int r = 0; r < mg; r++)
        for (int q = 0; q < nz; q++) G(r, q) = grow[r][q];
    Mat GH = matmul(G, Hinv);                 // mg x nz
    Mat GHG = matmul(GH, transpose(G));       // mg x mg
    o.flops_setup += 2.0 * mg * nz * nz + 2.0 * mg * mg * nz;
    const int ldy = o.ldy;
    o.Y.assign((size_t)ldy * ldy, 0.0);
    for (int p = 0; p < nz; p++)
        for (int q = 0; q < nz; q++) o.Y[(size_t)p * ldy + q] = Hinv(p, q);
    for (int r = 0; r < mg; r++)
        for (int q = 0; q < nz; q++) {
            o.Y[(size_t)(ldz + r) * ldy + q] = GH(r, q);
            o.Y[(size_t)q * ldy + ldz + r] = GH(r, q);
        }
    for (int r = 0; r < mg; r++)
        for (int s = 0; s < mg; s++) o.Y[(size_t)(ldz + r) * ldy + ldz + s] = GHG(r, s);

    // ADMM step sizes.  adaptive_rho: rho_i = kappa / Y_ii (Jacobi preconditioning of the dual
    // problem -- the model-level equivalent of what OSQP's per-solve adaptation converges to);
    // otherwise the caller's uniform rho.  Equality rows get 1e3 x, as OSQP does.
    const double kappa = 2.0, rho_min = 1e-6, rho_max = 1e6;
    auto clampr = [&](double r) { return std::min(std::max(r, rho_min), rho_max); };
    o.lw = lw; o.uw = uw;
    o.rho_b.assign(ldz, 0.0);
    for (int q = 0; q < nz; q++) {
        if (!finite_any(lw[q], uw[q])) continue;
        double r = prm.adaptive_rho ? kappa / std::max(Hinv(q, q), 1e-300) : prm.rho;
        if (lw[q] == uw[q]) r *= 1e3;
        o.rho_b[q] = clampr(r);
    }
    o.lg0.assign(ldg, -kInf); o.ug0.assign(ldg, kInf); o.rho_g.assign(ldg, 1.0);
    o.g_kind.assign(ldg, 0); o.g_step.assign(ldg, 0); o.g_comp.assign(ldg, 0); o.g_refrow.assign(ldg, -1);
    for (int r = 0; r < mg; r++) {
        o.lg0[r] = rows[r].lo; o.ug0[r] = rows[r].hi;
        double rr = prm.adaptive_rho ? kappa / std::max(GHG(r, r), 1e-300) : prm.rho;
        if (rows[r].lo == rows[r].hi) rr *= 1e3;
        o.rho_g[r] = clampr(rr);
        o.g_kind[r] = rows[r].kind; o.g_step[r] = rows[r].step; o.g_comp[r] = rows[r].comp;
        o.g_refrow[r] = rows[r].refrow;
    }

    // K = H + sigma I + diag(rho_b) + G' diag(rho_g) G ;  Kinv
    const double sigma = 1e-6;
    Mat K = H;
    for (int q = 0; q < nz; q++) K(q, q) += sigma + o.rho_b[q];
    for (int r = 0; r < mg; r++)
        for (int q = 0; q < nz; q++) {
            double gq = G(r, q) * o.rho_g[r];
            if (gq == 0.0) continue;
            for (int p = 0; p < nz; p++) K(p, q) += G(r, p) * gq;
        }
    Mat LK = K;
    if (!(cholesky_lower(LK) > 0)) return "ADMM matrix is not positive definite";
    Mat Kinv = spd_inverse_from_chol(LK);
    o.flops_setup += 2.0 * mg * nz * nz + 2.0 * nz * nz * nz / 3.0 + 2.0 * nz * nz * nz;

    // padded device layouts
    o.H.assign((size_t)ldz * ldz, 0.0);
    o.Kinv.assign((size_t)ldz * ldz, 0.0);
    for (int q = 0; q < nz; q++)
        for (int p = 0; p < nz; p++) {
            o.H[(size_t)q * ldz + p] = H(p, q);
            o.Kinv[(size_t)q * ldz + p] = Kinv(p, q);
        }
    o.Gr.assign((size_t)ldg * ldz, 0.0);
    o.Gc.assign((size_t)ldz * ldg, 0.0);
    for (int r = 0; r < mg; r++)
        for (int q = 0; q < nz; q++) {
            o.Gr[(size_t)r * ldz + q] = G(r, q);
            o.Gc[(size_t)q * ldg + r] = G(r, q);
        }

    }   // !light
    // var -> reference rows (CSR)
    o.boxrow_ptr.assign(ldz + 1, 0);
    for (auto &b : boxrefs) o.boxrow_ptr[b.var + 1]++;
    for (int q = 0; q < ldz; q++) o.boxrow_ptr[q + 1] += o.boxrow_ptr[q];
    o.boxrow_ref.assign(boxrefs.size() + 1, -1);
    o.boxrow_lo.assign(boxrefs.size() + 1, 0.0);
    o.boxrow_hi.assign(boxrefs.size() + 1, 0.0);
    {
        std::vector<int> fillp(o.boxrow_ptr.begin(), o.boxrow_ptr.end() - 1);
        for (auto &b : boxrefs) {
            int p = fillp[b.var]++;
            o.boxrow_ref[p] = b.refrow; o.boxrow_lo[p] = b.lo; o.boxrow_hi[p] = b.hi;
        }
    }
    (void)ndu;
    if (!light) build_fast_maps(o);
    return std::string();
}

// Mirrors assemble_one() of lmpc_kernels.hip in host doubles: free-response roll-out, weighted
// output error, adjoint pass, constraint offsets, feasibility-row values, cost constant.
void LmpcController::assemble_host(const Condensed &o, const double *x0, const double *u0, const Mat &yR,
                                   const Mat &uR, const Mat &dR, const Mat &dM, AsmOut &out) const
{
    const int nx = d.nx, nu = d.nu, ny = d.ny, ndu = d.ndu, ph = d.ph;
    std::vector<double> xb((size_t)(ph + 1) * nx), ey((size_t)(ph + 1) * ny);
    for (int a = 0; a < nx; a++) xb[a] = x0[a];
    for (int i = 1; i <= ph; i++)
        for (int a = 0; a < nx; a++) {
            double s = 0;
            for (int c = 0; c < nx; c++) s += A(a, c) * xb[(size_t)(i - 1) * nx + c];
            for (int dd = 0; dd < ndu; dd++) s += Bd(a, dd) * dM(dd, i - 1);
            xb[(size_t)i * nx + a] = s;
        }
    double c0 = 0;
    for (int i = 0; i <= ph; i++) {
        const int k = i > 0 ? i - 1 : 0;
        for (int a = 0; a < ny; a++) {
            double cx = 0;
            for (int c = 0; c < nx; c++) cx += C(a, c) * xb[(size_t)i * nx + c];
            double r = yR(a, k);
            for (int dd = 0; dd < ndu; dd++) r -= Dd(a, dd) * dM(dd, k);
            const double w = wOutput(a, i);
            ey[(size_t)i * ny + a] = w * (cx - r);
            c0 += w * (0.5 * cx * cx - r * cx);
        }
    }
    for (int j = 0; j < nu; j++) {
        const double u = u0[j];
        c0 += wU(j, 0) * (0.5 * u * u - uR(j, 0) * u);
        c0 += wDeltaU(j, 0) * (0.5 * u * u + dR(j, 0) * u);
    }
    out.c0 = c0;
    auto rowval = [&](int kind, int st, int cp) {
        double v = 0;
        if (kind == G_STATE) v = xb[(size_t)st * nx + cp];
        else if (kind == G_OUTPUT) {
            for (int c = 0; c < nx; c++) v += C(cp, c) * xb[(size_t)st * nx + c];
            for (int dd = 0; dd < ndu; dd++) v += Dd(cp, dd) * dM(dd, st > 0 ? st - 1 : 0);
        } else {
            for (int c = 0; c < nx; c++) v += sX[c] * xb[(size_t)st * nx + c];
        }
        return v;
    };
    out.goff.assign(o.mg, 0.0);
    for (int r = 0; r < o.mg; r++) out.goff[r] = rowval(o.g_kind[r], o.g_step[r], o.g_comp[r]);
    // feasibility rows: x0, lastU, y0, scalar row 0, then the rows that do not see the inputs
    out.sval.clear();
    for (int a = 0; a < nx; a++) out.sval.push_back(x0[a]);
    for (int j = 0; j < nu; j++) out.sval.push_back(u0[j]);
    for (int a = 0; a < ny; a++) {
        double v = 0;
        for (int c = 0; c < nx; c++) v += C(a, c) * x0[c];
        for (int dd = 0; dd < ndu; dd++) v += Dd(a, dd) * dM(dd, 0);
        out.sval.push_back(v);
    }
    {
        double v = 0;
        for (int c = 0; c < nx; c++) v += sX[c] * x0[c];
        for (int j = 0; j < nu; j++) v += sU[j] * u0[j];
        out.sval.push_back(v);
    }
    for (auto &fr : o.fixed_rows) out.sval.push_back(rowval(fr.kind, fr.step, fr.comp));
    // adjoint pass
    out.f.assign(o.nz, 0.0);
    std::vector<double> p(nx, 0.0), pn(nx);
    for (int i = ph; i >= 1; i--) {
        for (int b = 0; b < nx; b++) {
            double s = 0;
            for (int a = 0; a < ny; a++) s += C(a, b) * ey[(size_t)i * ny + a];
            for (int a = 0; a < nx; a++) s += A(a, b) * p[a];
            pn[b] = s;
        }
        p = pn;
        for (int j = 0; j < nu; j++) {
            double g = 0;
            for (int a = 0; a < nx; a++) g += B(a, j) * p[a];
            g -= wU(j, i) * uR(j, i - 1);
            out.f[o.blk[i] * nu + j] += g;
        }
    }
    for (int j = 0; j < nu; j++) {
        out.f[o.blk[1] * nu + j] -= wDeltaU(j, 0) * (u0[j] + dR(j, 0));
        for (int i = 1; i < ph; i++) {
            const int bn = o.blk[i + 1], bp = o.blk[i];
            if (bn != bp) {
                const double t = -wDeltaU(j, i) * dR(j, i - 1);
                out.f[bn * nu + j] += t;
                out.f[bp * nu + j] -= t;
            }
        }
    }
}

// Every quantity above is affine (the cost constant: quadratic) in (x0, lastU, yref): probe the
// host evaluation on unit vectors to tabulate the maps the MFMA assemble kernel multiplies with.
void LmpcController::build_fast_maps(Condensed &o) const
{
    const int nx = d.nx, nu = d.nu, ny = d.ny, ph = d.ph;
    auto r4 = [](int v) { return (v + 3) / 4 * 4; };
    auto r16 = [](int v) { return (v + 15) / 16 * 16; };
    o.nxp = r4(nx); o.nup = r4(nu); o.nyp = r4(ny);
    o.kin = o.nxp + o.nup + o.nyp + 4;
    o.ione = o.nxp + o.nup + o.nyp;
    o.nz16 = r16(o.nz); o.mg16 = r16(std::max(o.mg, 1));
    o.ns = nx + nu + ny + 1 + (int)o.fixed_rows.size();
    o.ns16 = r16(o.ns); o.kq16 = r16(o.kin);
    o.rowsA = o.nz16 + o.mg16 + o.ns16 + o.kq16;
    o.ldy16 = r16(o.ldy);
    const int offg = o.nz16, offs = offg + o.mg16, offq = offs + o.ns16;

    o.slo.assign(o.ns16, -kInf); o.shi.assign(o.ns16, kInf);
    {
        int q = 0;
        for (int a = 0; a < nx; a++, q++) { o.slo[q] = minX(a, 0); o.shi[q] = maxX(a, 0); }
        for (int j = 0; j < nu; j++, q++) { o.slo[q] = minU(j, 0); o.shi[q] = maxU(j, 0); }
        for (int a = 0; a < ny; a++, q++) { o.slo[q] = minY(a, 0); o.shi[q] = maxY(a, 0); }
        o.slo[q] = sMin[0]; o.shi[q] = sMax[0]; q++;
        for (auto &fr : o.fixed_rows) { o.slo[q] = fr.lo; o.shi[q] = fr.hi; q++; }
    }

    const int nin = nx + nu + ny;              // physical inputs
    auto col_of = [&](int k) { return k < nx ? k : (k < nx + nu ? o.nxp + (k - nx) : o.nxp + o.nup + (k - nx - nu)); };
    Mat zero_y(ny, ph);
    for (int variant = 0; variant < 2; variant++) {
        std::vector<double> &M = o.MA[variant];
        M.assign((size_t)o.rowsA * o.kin, 0.0);
        auto eval = [&](const std::vector<double> &v, AsmOut &out) {
            Mat yR = variant == 0 ? yRef : zero_y;
            if (variant == 1)
                for (int k = 0; k < ph; k++)
                    for (int a = 0; a < ny; a++) yR(a, k) = v[nx + nu + a];
            assemble_host(o, v.data(), v.data() + nx, yR, uRef, duRef, dMeas, out);
        };
        const int nprobe = variant == 0 ? nx + nu : nin;
        std::vector<double> v(nin, 0.0);
        AsmOut base; eval(v, base);
        auto put = [&](int row, int col, double val) { M[(size_t)col * o.rowsA + row] = val; };
        for (int r = 0; r < o.nz; r++) put(r, o.ione, base.f[r]);
        for (int r = 0; r < o.mg; r++) put(offg + r, o.ione, base.goff[r]);
        for (int r = 0; r < o.ns; r++) put(offs + r, o.ione, base.sval[r]);
        std::vector<AsmOut> e1(nprobe);
        std::vector<double> cneg(nprobe);
        for (int k = 0; k < nprobe; k++) {
            v.assign(nin, 0.0); v[k] = 1.0; eval(v, e1[k]);
            AsmOut neg; v[k] = -1.0; eval(v, neg); cneg[k] = neg.c0;
            const int c = col_of(k);
            for (int r = 0; r < o.nz; r++) put(r, c, e1[k].f[r] - base.f[r]);
            for (int r = 0; r < o.mg; r++) put(offg + r, c, e1[k].goff[r] - base.goff[r]);
            for (int r = 0; r < o.ns; r++) put(offs + r, c, e1[k].sval[r] - base.sval[r]);
        }
        // quadratic form of the cost constant: 0.5 [v;1]' Qc [v;1]
        const double c = base.c0;
        put(offq + o.ione, o.ione, 2.0 * c);
        for (int k = 0; k < nprobe; k++) {
            const double gp = e1[k].c0 - c, gm = cneg[k] - c;      // 0.5 A_kk +- b_k
            const double akk = gp + gm, bk = 0.5 * (gp - gm);
            const int ck = col_of(k);
            put(offq + ck, ck, akk);
            put(offq + ck, o.ione, bk); put(offq + o.ione, ck, bk);
            for (int m = 0; m < k; m++) {
                v.assign(nin, 0.0); v[k] = 1.0; v[m] = 1.0;
                AsmOut both; eval(v, both);
                const double akm = both.c0 - e1[k].c0 - e1[m].c0 + c;
                const int cm = col_of(m);
                put(offq + ck, cm, akm); put(offq + cm, ck, akm);
            }
        }
    }
    // -Y[:, 0:nz] in tile-padded layout
    o.Ym.assign((size_t)o.ldy16 * o.nz16, 0.0);
    for (int q = 0; q < o.nz; q++)
        for (int r = 0; r < o.ldy; r++) o.Ym[(size_t)q * o.ldy16 + r] = -o.Y[(size_t)q * o.ldy + r];
    compose_fused_maps(o);
}

// [t0; gt0] = (-Y[:, :nz]) f and f = MA_f vin: compose the two products once per controller, so that a wavefront gets the whole
// record of its instance from one mat-vec with the 32-odd inputs (lmpc_solve_fused) instead of reading it back from HBM
void LmpcController::compose_fused_maps(Condensed &o) const
{
    const int offg = o.nz16, offs = offg + o.mg16, offq = offs + o.ns16;
    o.nsp = (o.ns + 1) / 2 * 2;
    o.rowsF = o.ldy + o.ldg + o.ldz + o.nsp + o.kin;
    const int r_goff = o.ldy, r_f = r_goff + o.ldg, r_s = r_f + o.ldz, r_q = r_s + o.nsp;
    for (int variant = 0; variant < 2; variant++) {
        const std::vector<double> &M = o.MA[variant];
        std::vector<double> &F = o.MF[variant];
        F.assign((size_t)o.rowsF * o.kin, 0.0);
        for (int c = 0; c < o.kin; c++) {
            const double *mc = &M[(size_t)c * o.rowsA];
            double *fc = &F[(size_t)c * o.rowsF];
            for (int q = 0; q < o.nz; q++) {
                const double fq = mc[q];
                if (fq == 0.0) continue;
                const double *yq = &o.Ym[(size_t)q * o.ldy16];
                for (int r = 0; r < o.ldy; r++) fc[r] += yq[r] * fq;
            }
            for (int r = 0; r < o.mg; r++) fc[r_goff + r] = mc[offg + r];
            for (int r = 0; r < o.nz; r++) fc[r_f + r] = mc[r];
            for (int r = 0; r < o.ns; r++) fc[r_s + r] = mc[offs + r];
            for (int r = 0; r < o.kin; r++) fc[r_q + r] = mc[offq + r];
        }
    }
}

void LmpcController::refresh_fast_maps(Condensed &o) const
{
    const int nx = d.nx, nu = d.nu, ny = d.ny, ph = d.ph;
    const int offg = o.nz16, offs = offg + o.mg16, offq = offs + o.ns16;
    const int nin = nx + nu + ny;
    auto col_of = [&](int k) { return k < nx ? k : (k < nx + nu ? o.nxp + (k - nx) : o.nxp + o.nup + (k - nx - nu)); };
    Mat zero_y(ny, ph);
    for (int variant = 0; variant < 2; variant++) {
        std::vector<double> &M = o.MA[variant];
        auto eval = [&](const std::vector<double> &v, AsmOut &out) {
            Mat yR = variant == 0 ? yRef : zero_y;
            if (variant == 1)
                for (int k = 0; k < ph; k++)
                    for (int a = 0; a < ny; a++) yR(a, k) = v[nx + nu + a];
            assemble_host(o, v.data(), v.data() + nx, yR, uRef, duRef, dMeas, out);
        };
        auto put = [&](int row, int col, double val) { M[(size_t)col * o.rowsA + row] = val; };
        auto get = [&](int row, int col) { return M[(size_t)col * o.rowsA + row]; };
        const int nprobe = variant == 0 ? nx + nu : nin;
        std::vector<double> v(nin, 0.0);
        AsmOut base; eval(v, base);
        for (int r = 0; r < o.nz; r++) put(r, o.ione, base.f[r]);
        for (int r = 0; r < o.mg; r++) put(offg + r, o.ione, base.goff[r]);
        for (int r = 0; r < o.ns; r++) put(offs + r, o.ione, base.sval[r]);
        const double c = base.c0;
        put(offq + o.ione, o.ione, 2.0 * c);
        for (int k = 0; k < nprobe; k++) {
            // c0(e_k) = c + b_k + A_kk / 2 with A_kk kept from the full build
            v.assign(nin, 0.0); v[k] = 1.0;
            AsmOut e1; eval(v, e1);
            const int ck = col_of(k);
            const double bk = e1.c0 - c - 0.5 * get(offq + ck, ck);
            put(offq + ck, o.ione, bk); put(offq + o.ione, ck, bk);
        }
    }
    compose_fused_maps(o);
}

void lmpc_pack_mfma_tiles(const double *src, int rows, int K, double *out)
{
    const int T = rows / 16, G = (K + 15) / 16;
    for (int t = 0; t < T; t++)
        for (int g = 0; g < G; g++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 4; e++) {
                    const int k = 4 * (4 * g + e) + (lane >> 4);
                    out[(((size_t)t * G + g) * 64 + lane) * 4 + e] = k < K ? src[(size_t)k * rows + 16 * t + (lane & 15)] : 0.0;
                }
}

}  // namespace mpcx
