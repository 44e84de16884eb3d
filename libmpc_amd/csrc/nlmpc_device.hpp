// POD views handed to the NLMPC kernels (nlmpc_kernels.hip).
#pragma once

#include <cstdint>

namespace mpcx {

constexpr int kNlMaxWorking = 128;     // rows the QP sub-solver may hold active at once
constexpr int kNlLdsWorking = 24;      // up to this many, their Schur complement is factored in LDS

// offsets (in doubles) into one instance's slice of the SQP workspace
struct NlmpcWsLayout {
    int z, d, g, c, jeq, gin, jin;      // iterate, step, transcription outputs
    int r, phi, einv;                   // condensing: x-step for p = 0, d x / d p, inverses of dc_i/dx_{i+1}
    int gr, art, br;                    // reduced gradient, reduced inequality Jacobian (transposed), its offset
    int hinv, mu, glold, s, p;          // inverse BFGS matrix, multipliers, BFGS memory, QP solution
    int qn, qv, qs, qs2;                   // QP: normals and Hinv*normals of the working set, their Schur complement
    int scal;                           // scalars: [0] cost, [2..7] per-phase cycle counts
    int lamw;                           // right-hand side of the backward sweep for the dynamics multipliers
    int total;
};

struct NlmpcDev {
    int model_id, nx, nu, ph, ch, nz, neq, nineq;
    int ny;                     // outputs (OptSequence::output)
    int nue;                    // user equalities: rows nineq .. nineq+nue-1 of the user constraint arrays
    int nzu, nr;                // ch*nu, ch*nu + 1
    int kw;                     // working-set capacity: min(kNlMaxWorking, rows, variables)
    int lds_per_wave;           // doubles
    double Ts;
    const double *params;       // model parameters in HBM
    // box bounds on the decision vector (NLOptimizer::lb / ub): all of them, and the finite ones as sub-problem rows
    const double *zlb, *zub;    // [nz]
    int nbnd;
    const int *bnd_idx;         // [nbnd] index into z
    const double *bnd_sign;     // [nbnd] +1: z <= val, -1: z >= val
    const double *bnd_val;      // [nbnd]
    NlmpcWsLayout ws;
};

struct NlmpcBatchDev {
    int batch;
    const double *z;            // [B x nz]   decision vectors [X(1..ph) | U blocks | slack]
    const double *x0;           // [B x nx]
    double *cost, *grad;        // [B], [B x nz]
    double *ceq, *jeq;          // [B x ph*nx], [B x ph x nx x (2nx+nu)] blocks [dc/dx_i | dc/dx_{i+1} | dc/du_i]
    double *cineq, *jineq;      // [B x (nineq+nue)], [B x (nineq+nue) x nz] row-major: user inequalities, then user equalities
};

struct NlmpcSolveDev {
    int batch;
    const double *x0, *u0;      // [B x nx], [B x nu]
    const double *z_warm;       // [B x nz] previous solutions (shifted one step on entry) or null = cold start
    double *ws;                 // [B x ws.total]
    int max_iter, hard;
    int keep_curvature;         // 1: start from the inverse BFGS matrix already in the workspace (receding-horizon extension)
    double tol_step, tol_con, ieq_tol, eq_tol;
    double *cmd, *cost, *z_out; // [B x nu], [B], [B x nz]
    int32_t *status, *solver_status, *is_feasible, *iterations;
    double *seq_state, *seq_input;      // [B x (ph+1) x nx], [B x (ph+1) x nu]
    double *seq_output;                 // [B x (ph+1) x ny]
};

int nlmpc_model_dims(int model_id, int *nx, int *nu, int *ny, int ph, int *nineq, int *nue);
void nlmpc_plan(NlmpcDev &m);           // fills nzu, nr, lds_per_wave, ws from the dimensions
int nlmpc_launch(const NlmpcDev &m, const NlmpcBatchDev &b, void *stream);
int nlmpc_launch_solve(const NlmpcDev &m, const NlmpcSolveDev &b, void *stream);

}  // namespace mpcx
