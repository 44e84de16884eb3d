// POD views handed to the NLMPC kernels (mpcx/nlmpc_engine.hpp).  Shared by the library (zoo models, run-time compiled
// hooks) and by a user's translation unit that instantiates the engine for its own hooks (mpcx/nlmpc_hooks.hpp): the
// library owns every buffer these structs point to and fills them; the kernels only read the layout.
#pragma once


namespace mpcx {

constexpr int kNlMaxWorking = 128;     // rows the QP sub-solver may hold active at once
constexpr int kNlLdsWorking = 24;      // the LDS slice holds the factor of a working set of at least this many rows
constexpr int kNlTrials = 8;           // step lengths the line search evaluates at a time
constexpr int kSpDense = 0x7fff0000;   // marker in the LDS word of a row that is not in the sparse form
constexpr int kNlSparse = 4;           // sub-problem rows with at most this many entries are also kept as (index, value) lists

// offsets (in doubles) into one instance's slice of the SQP workspace
struct NlmpcWsLayout {
    int z, d, g, c, jeq, gin, jin;      // iterate, step, transcription outputs
    int r, phi, einv;                   // condensing: x-step for p = 0, d x / d p, inverses of dc_i/dx_{i+1}
    int gr, art, br;                    // reduced gradient, reduced inequality Jacobian (transposed), its offset
    int hinv, mu, glold, s, p;          // inverse BFGS matrix, multipliers, BFGS memory, QP solution
    int qn, qv, qs, qs2;                   // QP: normals and Hinv*normals of the working set, their Schur complement
    int scal;                           // scalars: [0] cost, [2..7] per-phase cycle counts
    int lamw;                           // right-hand side of the backward sweep for the dynamics multipliers
    int hook;                           // vector-valued user hooks only: two column buffers [64 x rows] for the central
                                        // differences, the line search's constraint values [kNlTrials x rows] and output
                                        // trajectories [kNlTrials x (ph+1) x ny]
    int sp;                             // sparse form of the sub-problem's rows: values [rows x kNlSparse], indices (int), counts (int)
    int total;
};

struct NlmpcDev {
    int model_id, nx, nu, ph, ch, nz, neq, nineq;
    int ny;                     // outputs (OptSequence::output)
    int nue;                    // user equalities: rows nineq .. nineq+nue-1 of the user constraint arrays
    int nzu, nr;                // ch*nu, ch*nu + 1
    int kw;                     // working-set capacity: min(kNlMaxWorking, rows, variables)
    int nl;                     // rows of the working set whose factor fits the LDS slice (beyond: factored in the workspace)
    int lds_per_wave;           // doubles
    int lds_blocks;             // offset (doubles) in the wavefront's LDS slice of the LDS-resident dynamics blocks, operands of the sweeps:
                                // jeq [ph nx (2nx+nu)] | einv [ph nx nx] | c [ph nx] | lamw [ph nx] | p [nr]; -1: they live in the workspace
    int continuous;             // hook models: 1 = the state function is dx/dt (setDiscretizationSamplingTime was called)
    int has_output;             // hook models: 1 = an output function was set (otherwise Y reads as zeros, Model.hpp:72-96)
    int vector_hooks;           // 1 = user hooks with the reference's whole-vector signatures (sizes the hook workspace)
    double Ts;
    const double *params;       // model parameters in HBM (zoo) or the hook closures (mpcx/nlmpc_hooks.hpp)
    // Mapping scalings (Mapping.hpp:71-86): U = input_scale * z_u, X = z_x / state_scale; arrays of ones by default
    const double *su, *ss, *iss; // [nu], [nx], [nx] = 1 / ss
    int scaled;                 // 0: every factor is 1, the kernels skip them
    // box bounds on the decision vector (NLOptimizer::lb / ub): all of them, and the finite ones as sub-problem rows
    const double *zlb, *zub;    // [nz]
    int nbnd;
    int nbnd_state;             // how many of them bound a state: the first rows of the table (ascending index into z)
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
    double *hook_ws;            // vector-valued user hooks: [B x hook scratch] (handle-owned), null otherwise
    int hook_ld;
    const double *params_b;     // optional [B x nparams]: every instance its own model parameters (built-in systems)
    int nparams;
};

struct NlmpcSolveDev {
    int batch;
    const double *x0, *u0;      // [B x nx], [B x nu]
    const double *z_warm;       // [B x nz] previous solutions (shifted one step on entry) or null = cold start
    double *ws;                 // [B x ws.total]
    int max_iter, hard;
    int keep_curvature;         // 1: start from the inverse BFGS matrix already in the workspace (receding-horizon extension)
    double tol_step, tol_con, ieq_tol, eq_tol;
    // nlopt's stopping tolerances (NLOptimizer.hpp:135-138; <= 0: disabled, the reference's default)
    double ftol_rel, ftol_abs, xtol_rel, xtol_abs;
    double *cmd, *cost, *z_out; // [B x nu], [B], [B x nz]
    int *status, *solver_status, *is_feasible, *iterations;     // int32
    double *seq_state, *seq_input;      // [B x (ph+1) x nx], [B x (ph+1) x nu]
    double *seq_output;                 // [B x (ph+1) x ny]
    double *mu_out;                     // [B x (nineq + nue + nbnd)] multipliers of the last sub-problem (0 = inactive)
    const double *params_b;             // optional [B x nparams]: every instance its own model parameters (built-in systems)
    int nparams;
};

// doubles of hook scratch per instance (NlmpcWsLayout::hook, NlmpcBatchDev::hook_ws): two column buffers [rows x 64], the
// line search's constraint values [kNlTrials x rows] and its output trajectories [kNlTrials x (ph+1) x ny]
inline int nlmpc_hook_scratch(const NlmpcDev &m)
{
    const int rows = m.nineq + m.nue;
    return 2 * 64 * rows + kNlTrials * rows + kNlTrials * (m.ph + 1) * m.ny + 2;
}

// wavefronts per workgroup of the two NLMPC kernels: a power of two, so that the blocks of a CU (160 KB of LDS) leave no slice
// unused; 0 = the slice exceeds the 64 KB a workgroup may ask for.  One definition for the compiled and the run-time compiled path.
inline int nlmpc_waves_per_block(const NlmpcDev &m)
{
    const unsigned long bytes = (unsigned long)m.lds_per_wave * sizeof(double);
    return bytes <= 16 * 1024 ? 4 : bytes <= 32 * 1024 ? 2 : bytes <= 64 * 1024 ? 1 : 0;
}

// ---- host-side plumbing ---------------------------------------------------------------------------------------
// How the library launches the two kernels of a controller.  Zoo models: thunks inside libmpcx.so.  User hooks compiled
// in the user's translation unit (mpcx/nlmpc_hooks.hpp): thunks instantiated there and registered through
// mpcx_nlmpc_create_custom (include/mpcx.h).  Both return 0, -2 (LDS budget) or -3 (launch error).
// Where a solve leaves the inverse BFGS matrix for a receding-horizon successor (NlmpcSolveDev::keep_curvature): the fields that decide it,
// compared one by one (form: 0 nlmpc_sqp, 1 nlmpc_sqp_wg, < 0 nowhere)
struct NlmpcCurvLayout {
    int form, waves, f_lds, w_hinv, hard, nbnd_state, nr;
    bool operator==(const NlmpcCurvLayout &o) const
    {
        return form >= 0 && form == o.form && waves == o.waves && f_lds == o.f_lds && w_hinv == o.w_hinv && hard == o.hard && nbnd_state == o.nbnd_state && nr == o.nr;
    }
};
typedef int (*nlmpc_launch_eval_fn)(void *ctx, const NlmpcDev *m, const NlmpcBatchDev *b, void *stream);
typedef int (*nlmpc_launch_solve_fn)(void *ctx, const NlmpcDev *m, const NlmpcSolveDev *b, void *stream);

}  // namespace mpcx
