/* mpcx -- C ABI of the MI355X-native batched MPC solve engine.
 *
 * This is the drop-in boundary for ONE path of libmpc++: what sits behind
 * IOptimizer<sizer>::run (reference include/mpc/IOptimizer.hpp:24-58), for a
 * *batch* of independent MPC instances that share one controller set-up:
 *   mpcx_lmpc_*   the linear back-end, LOptimizer::run + ProblemBuilder
 *                 (include/mpc/LMPC/LOptimizer.hpp:189-368, LMPC/ProblemBuilder.hpp);
 *   mpcx_nlmpc_*  the non-linear back-end, NLOptimizer::run with Mapping / Model /
 *                 Objective / Constraints (include/mpc/NLMPC/NLOptimizer.hpp:412-638);
 *   mpcx_discretize_batch  the c2d set-up helper (include/mpc/Utils.hpp:23-89).
 *
 * Conventions
 *   - plain C, opaque handle, no C++/torch types in any signature;
 *   - every matrix argument of a setter is a HOST pointer to column-major
 *     doubles (Eigen's default layout, reference include/mpc/Types.hpp:42);
 *   - every pointer inside mpcx_lmpc_batch is a DEVICE pointer (HBM), batch
 *     arrays are instance-major: x0[b*nx + j];
 *   - every call returns MPCX_OK (0) or a negative MPCX_E_* code; the per-instance
 *     outcome of a solve is reported in status[] / solver_status[] / is_feasible[]
 *     exactly like mpc::Result (Types.hpp:168-182).
 *   - horizon-slice semantics are the reference's: a slice is the half-open
 *     step range [start, end); {-1,-1} = whole horizon (Types.hpp:57-82,
 *     IMPC.hpp:244-283).  Replicate-along-horizon setters follow LMPC.hpp.
 */
#ifndef MPCX_H
#define MPCX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCX_OK              0
#define MPCX_E_INVALID      -1   /* bad argument / dimension / slice            */
#define MPCX_E_UNSUPPORTED  -2   /* dimension outside what the kernels cover    */
#define MPCX_E_DEVICE       -3   /* HIP runtime error                           */
#define MPCX_E_NUMERIC      -4   /* set-up factorisation failed                 */
#define MPCX_E_STATE        -5   /* call order (e.g. solve before a model)      */

/* mpc::ResultStatus (Types.hpp:87-94) */
#define MPCX_STATUS_SUCCESS        0
#define MPCX_STATUS_MAX_ITERATION  1
#define MPCX_STATUS_INFEASIBLE     2
#define MPCX_STATUS_ERROR          3
#define MPCX_STATUS_UNKNOWN        4

/* solver_status: OSQP v0.6.3 status_val codes, which is what LOptimizer stores
 * in Result::solver_status (LOptimizer.hpp:342) */
#define MPCX_SOLVER_SOLVED               1
#define MPCX_SOLVER_SOLVED_INACCURATE    2
#define MPCX_SOLVER_MAX_ITER_REACHED    -2
#define MPCX_SOLVER_PRIMAL_INFEASIBLE   -3
#define MPCX_SOLVER_NON_CVX             -7
#define MPCX_SOLVER_UNSOLVED           -10

/* Dimensions: the run-time twin of MPCSize (reference include/mpc/Dim.hpp). */
typedef struct mpcx_dims {
    int nx, nu, ndu, ny, ph, ch;
} mpcx_dims;

/* POD mirror of mpc::LParameters (Types.hpp:99-114,146-161). */
typedef struct mpcx_lparams {
    int    maximum_iteration;   /* default 100 */
    double time_limit;          /* accepted, ignored (no wall clock inside a kernel) */
    int    enable_warm_start;   /* default 0 */
    double alpha;               /* 1.6  */
    double rho;                 /* 1e-6; used as the uniform ADMM step only when adaptive_rho == 0 */
    double eps_rel, eps_abs;    /* 1e-4 */
    double eps_prim_inf, eps_dual_inf;  /* 1e-3 */
    int    verbose;
    int    adaptive_rho;        /* 1: step sizes from the model's dual Hessian diagonal */
    int    polish;              /* 1: finish on the exact active-set solution, as OSQP's polish   */
} mpcx_lparams;

typedef struct mpcx_lmpc *mpcx_lmpc_t;

/* How a per-solve reference / exogenous-input array is laid out in HBM. */
#define MPCX_REF_SHARED        0  /* use the matrix given to the host setter (LMPC::setReferences) */
#define MPCX_REF_PER_INSTANCE  1  /* [B x n]      one vector per instance, constant along horizon  */
#define MPCX_REF_PER_STEP      2  /* [B x ph x n] full matrix per instance (column k = step k)     */

/* One batched LOptimizer::run.  All pointers are device pointers; optional
 * outputs may be NULL.  Replaces LOptimizer::run(x0,u0) (LOptimizer.hpp:189). */
typedef struct mpcx_lmpc_batch {
    int batch;
    const double *x0;              /* [B x nx]  measured state                                  */
    const double *u0;              /* [B x nu]  last applied input (lastU)                       */
    const double *yref;  int yref_mode;    /* output reference   (ny)  */
    const double *uref;  int uref_mode;    /* input reference    (nu)  */
    const double *duref; int duref_mode;   /* delta-u reference  (nu)  */
    const double *dmeas; int dmeas_mode;   /* exogenous inputs   (ndu) */
    /* Result<nu> as structure-of-arrays */
    double *cmd;                   /* [B x nu]  required: optimal first input, Result::cmd       */
    double *cost;                  /* [B]       0.5 z'Pz + q'z of the reference QP               */
    int32_t *status;               /* [B]       MPCX_STATUS_*                                    */
    int32_t *solver_status;        /* [B]       MPCX_SOLVER_*                                    */
    int32_t *is_feasible;          /* [B]                                                        */
    int32_t *iterations;           /* [B]       ADMM iterations spent                            */
    /* active set at the returned point, reference row numbering of A/l/u
     * (ProblemBuilder.hpp:814-822), one bit per row, [B x active_words] words */
    uint32_t *active_lower;
    uint32_t *active_upper;
    /* OptSequence (Types.hpp:184-198): row i = horizon step i, instance-major
     * [B x (ph+1) x n], row-major inside an instance */
    double *seq_state;
    double *seq_output;
    double *seq_input;
    /* extension, for work accounting: active-set (polish) rounds spent and size of the final
     * working set, [B] each */
    int32_t *polish_rounds;
    int32_t *active_count;
    /* warm start (inputs, may be NULL): the active_lower / active_upper words of the previous solve
     * of each instance.  The reference warm-starts OSQP with the previous primal/dual pair
     * (LOptimizer.hpp:268-281, LParameters::enable_warm_start); for the active-set iteration the
     * information that carries over is which rows were active: they are the first working set, and
     * an unchanged active set verifies in one round.  Results do not depend on it.               */
    const uint32_t *warm_active_lower;
    const uint32_t *warm_active_upper;
    int warm_shift;                /* 1: receding horizon -- the previous tick's row of step i+1 seeds step i */
} mpcx_lmpc_batch;

/* ---- lifetime (replaces LMPC::onSetup / new LOptimizer, LMPC.hpp:728-735) ---- */
int mpcx_lmpc_create(const mpcx_dims *dims, int device, mpcx_lmpc_t *out);
int mpcx_lmpc_destroy(mpcx_lmpc_t h);
void mpcx_lparams_default(mpcx_lparams *p);
const char *mpcx_last_error(void);

/* ---- controller set-up: names and semantics of LMPC.hpp ---------------------- */
/* LMPC::setStateSpaceModel (LMPC.hpp:493) */
int mpcx_lmpc_set_state_space_model(mpcx_lmpc_t h, const double *A, const double *B, const double *C);
/* LMPC::setDisturbances (LMPC.hpp:518) */
int mpcx_lmpc_set_disturbances(mpcx_lmpc_t h, const double *Bd, const double *Dd);
/* LMPC::setObjectiveWeights matrix form (LMPC.hpp:306) / vector+slice form (LMPC.hpp:436) */
int mpcx_lmpc_set_objective_weights(mpcx_lmpc_t h, const double *OW, const double *UW, const double *DUW);
int mpcx_lmpc_set_objective_weights_slice(mpcx_lmpc_t h, const double *ow, const double *uw, const double *duw,
                                          int start, int end);
/* LMPC::setStateBounds / setInputBounds / setOutputBounds (LMPC.hpp:111-292) */
int mpcx_lmpc_set_state_bounds(mpcx_lmpc_t h, const double *xmin, const double *xmax);
int mpcx_lmpc_set_state_bounds_slice(mpcx_lmpc_t h, const double *xmin, const double *xmax, int start, int end);
int mpcx_lmpc_set_input_bounds(mpcx_lmpc_t h, const double *umin, const double *umax);
int mpcx_lmpc_set_input_bounds_slice(mpcx_lmpc_t h, const double *umin, const double *umax, int start, int end);
int mpcx_lmpc_set_output_bounds(mpcx_lmpc_t h, const double *ymin, const double *ymax);
int mpcx_lmpc_set_output_bounds_slice(mpcx_lmpc_t h, const double *ymin, const double *ymax, int start, int end);
/* LMPC::setScalarConstraint slice form (LMPC.hpp:355) and index form (LMPC.hpp:409) */
int mpcx_lmpc_set_scalar_constraint_slice(mpcx_lmpc_t h, double smin, double smax, const double *X, const double *U,
                                          int start, int end);
int mpcx_lmpc_set_scalar_constraint_index(mpcx_lmpc_t h, int index, double smin, double smax,
                                          const double *X, const double *U);
/* LMPC::setReferences matrix form (LMPC.hpp:596) / vector+slice form (LMPC.hpp:616) */
int mpcx_lmpc_set_references(mpcx_lmpc_t h, const double *yref, const double *uref, const double *duref);
int mpcx_lmpc_set_references_slice(mpcx_lmpc_t h, const double *yref, const double *uref, const double *duref,
                                   int start, int end);
/* LMPC::setExogenousInputs matrix form (LMPC.hpp:534) / vector+slice form (LMPC.hpp:550) */
int mpcx_lmpc_set_exogenous_inputs(mpcx_lmpc_t h, const double *dmeas);
int mpcx_lmpc_set_exogenous_inputs_slice(mpcx_lmpc_t h, const double *dmeas, int start, int end);
/* LMPC::setOptimizerParameters -> LOptimizer::setParameters (LMPC.hpp:79, LOptimizer.hpp:100) */
int mpcx_lmpc_set_optimizer_parameters(mpcx_lmpc_t h, const mpcx_lparams *p);

/* Extension (no reference counterpart).  With the reference an infeasible QP is never reported
 * INFEASIBLE: libmpc++ hands OSQP true infinities, OSQP v0.6.3's certificate test then evaluates
 * inf*0 = NaN and never fires, the solve runs out of iterations and LOptimizer returns the last
 * ADMM iterate flagged MAX_ITERATION / is_feasible = true (LOptimizer.hpp:344).  Default (0):
 * same flags, cmd = the optimum of the QP without the rows that (x0, lastU) alone violate, or the
 * last ADMM iterate when the infeasibility involves the inputs.  1: status INFEASIBLE, cmd = NaN. */
int mpcx_lmpc_set_strict_infeasibility(mpcx_lmpc_t h, int on);

/* ---- the hot path --------------------------------------------------------------- */
/* Condense the controller into device-resident matrices.  Called implicitly by
 * the first solve after any setter; exposed so that set-up cost can be paid
 * (and timed) up front. */
int mpcx_lmpc_setup(mpcx_lmpc_t h);
/* Batched LOptimizer::run on `stream` (a hipStream_t, NULL = default stream).
 * Asynchronous with respect to the host; outputs are valid once the stream
 * has been synchronised.
 * Concurrency: a handle owns ONE per-instance workspace and ONE set of dispatch queues.  Launches of the
 * same handle on the same stream are ordered and safe; two launches of the same handle in flight on
 * DIFFERENT streams race on them (instances could be dropped or solved twice).  Overlap batches by giving
 * each stream its own handle (set-up cost is per handle and paid once), as bench.py's pipelined leg does.
 * maximum_iteration: the main iteration here is OSQP's polish step repeated until the KKT conditions verify
 * (DESIGN.md 4.3); it needs no ADMM iterations, so LParameters::maximum_iteration only bounds the ADMM
 * fallback that takes over when the polish does not verify (or when polish = 0).  `iterations[]` reports
 * the ADMM iterations actually spent (0 on the polish path), `polish_rounds[]` the active-set rounds.  A QP the
 * reference would leave MAX_ITER_REACHED and un-polished comes back here as the exact optimum with SUCCESS. */
int mpcx_lmpc_solve_batch(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, void *stream);
/* Same launch, timed with HIP events recorded on `stream` around `repeats`
 * back-to-back launches; returns the mean kernel time in milliseconds. */
int mpcx_lmpc_time_solve_batch(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, void *stream, int repeats, float *ms_mean);

/* One step as a HIP graph.  A control loop solves the same-sized batch from the same buffers every tick: the launches of
 * one mpcx_lmpc_solve_batch (dispatch-queue reset, assemble, solve, fallback) are captured once for a batch descriptor and
 * replayed with a single hipGraphLaunch.  The descriptor's pointers are baked into the graph; what they point to may change
 * between launches (new x0, u0, references), the controller's set-up may not (create a new graph after a setter).
 * `stream` of create: any non-default stream (used for one untimed warm-up solve and for the capture).  A handle still
 * has one workspace: one launch of it in flight at a time, graph or not.                                          */
typedef struct mpcx_lmpc_graph *mpcx_lmpc_graph_t;
int mpcx_lmpc_graph_create(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, void *stream, mpcx_lmpc_graph_t *out);
int mpcx_lmpc_graph_launch(mpcx_lmpc_graph_t g, void *stream);
int mpcx_lmpc_graph_destroy(mpcx_lmpc_graph_t g);

/* Convenience for callers whose data lives in host memory (the reference's optimize(x0, lastU) is
 * such a caller): stages the inputs to HBM, runs mpcx_lmpc_solve_batch on the default stream, copies
 * the results back and synchronises.  References use the matrices given to the host setters.  Any
 * output pointer may be NULL except cmd.  seq_* are [B x (ph+1) x n] row-major, as in the batch struct. */
int mpcx_lmpc_solve_host(mpcx_lmpc_t h, int batch, const double *x0, const double *u0,
                         double *cmd, double *cost, int32_t *status, int32_t *solver_status, int32_t *is_feasible,
                         double *seq_state, double *seq_output, double *seq_input);

/* ---- introspection -------------------------------------------------------------- */
/* sizes of the reference QP (ProblemBuilder.hpp:70-76) and of the condensed one */
typedef struct mpcx_lmpc_info {
    int n_ref, m_ref, neq_ref;     /* reference QP variables / rows / equality rows          */
    int nz, mg;                    /* condensed: decision variables, general inequality rows */
    int active_words;              /* uint32 words per instance in active_lower/upper        */
    int kernel_variant;            /* which template instantiation serves these dimensions   */
    double flops_setup;            /* algorithmic flops of one set-up                         */
    double flops_per_admm_iter;    /* algorithmic flops of one ADMM iteration of one instance */
    double flops_fixed_per_solve;  /* assembly + unconstrained solve + cost, per instance     */
    double bytes_per_solve;        /* algorithmic HBM bytes per instance (inputs + outputs)   */
} mpcx_lmpc_info;
int mpcx_lmpc_get_info(mpcx_lmpc_t h, mpcx_lmpc_info *info);

/* ---- NLMPC transcription (rows a10-a20 of SURVEY.md 8) ---------------------------- */
/* What the reference computes inside every NLopt callback of NLOptimizer<> (NLOptimizer.hpp:760-997):
 * Mapping::unwrapVector (Mapping.hpp:174-211), Objective::evaluate + forward-difference gradient
 * (Objective.hpp:91-265), the dynamics equalities with their central-difference Jacobian blocks
 * (Constraints.hpp:490-628, 844-905) and the user inequalities with theirs (Constraints.hpp:211-316),
 * for a batch of decision vectors in one launch.  The reference's user hooks are host std::function
 * objects (IDimensionable.hpp:94-149); device code cannot call those.  The hooks are device code instead:
 * the reference's example systems are built in and picked by id (below); any other system comes in through
 * mpcx_nlmpc_create_custom (hooks compiled by the caller with hipcc, what mpc::NLMPC<>'s setters use) or
 * mpcx_nlmpc_create_from_source (hooks compiled at run time from the lambda bodies, what the Python front-end uses). */
typedef struct mpcx_nlmpc *mpcx_nlmpc_t;
enum { MPCX_MODEL_VANDERPOL = 1,   /* examples/vanderpol_ex.cpp: nx=2 nu=1, continuous, ineq u_i <= 0.5        */
       MPCX_MODEL_UGV = 2,         /* examples/ugv_ex.cpp: nx=4 nu=2, discrete, two circular obstacles           */
       MPCX_MODEL_OSCILLATORS6 = 3,/* examples/networked_oscillators_ex.cpp: 6 coupled oscillators, nx=12 nu=6   */
       MPCX_MODEL_OSCILLATORS8 = 4,/* the same network with 8 oscillators (BASELINE config 5), nx=16 nu=8        */
       MPCX_MODEL_VANDERPOL_TERMINAL = 5,/* Van der Pol + the user equality x(ph) = 0 (setEqConFunction path)    */
       MPCX_MODEL_VANDERPOL_RATE = 6 };/* Van der Pol + a rate limit |u_i - u_{i-1}| <= params[0] (default 0.1): inequality rows
                                          with two entries, several rows on one input (no reference example has them)  */
typedef struct mpcx_nlmpc_dims {
    int nx, nu, ph, ch;
    int nz;      /* decision variables  ph*nx + ch*nu + 1 (Objective.hpp:45)                         */
    int neq;     /* dynamics equalities ph*nx                                                         */
    int nineq;   /* user inequalities                                                                 */
    int jeq_w;   /* width of one equality Jacobian block row: 2*nx + nu                               */
    int neq_user;/* user equalities (NLMPC::setEqConFunction)                                         */
    int ny;      /* outputs (NLMPC::setOutputFunction; zeros in the sequence when the model has none)  */
    int nbnd;    /* finite state / input bounds: rows nineq + neq_user .. of the sub-problem (multipliers) */
    int n_params;/* model parameters of a built-in system = columns of mpcx_nlmpc_batch.params (0: hooks)  */
} mpcx_nlmpc_dims;
/* NLMPC::setDiscretizationSamplingTime / setStateSpaceFunction / setObjectiveFunction /
 * setIneqConFunction (NLMPC.hpp:108-214) for a built-in model; `params` (n doubles, may be NULL
 * for the model's defaults) are the constants its functors capture in the reference example.
 * UGV: [v_pref_x, v_pref_y, ox0, oy0, r0, ox1, oy1, r1, Ts].  Oscillators: [mu, k].  Van der Pol:
 * none (Ts is the collocation step).                                                             */
int mpcx_nlmpc_create(int model_id, int ph, int ch, double Ts, const double *params, int n_params,
                      int device, mpcx_nlmpc_t *out);
int mpcx_nlmpc_destroy(mpcx_nlmpc_t h);
int mpcx_nlmpc_get_dims(mpcx_nlmpc_t h, mpcx_nlmpc_dims *d);

/* ---- user hooks (NLMPC::setStateSpaceFunction / setOutputFunction / setObjectiveFunction / setIneqConFunction /
 * setEqConFunction, NLMPC.hpp:139-281; handle types IDimensionable.hpp:94-149) ----------------------------------------
 * The reference's hooks are host closures, which a kernel cannot call.  Two ways to give this library device code with the
 * same signatures instead of picking a built-in model:
 *
 * (1) compiled by the caller.  The engine is a header (include/mpcx/nlmpc_engine.hpp); a translation unit compiled with
 *     hipcc instantiates it for its own hook types (include/mpcx/nlmpc_hooks.hpp -- what mpc::NLMPC<>'s setters do in
 *     include/mpcx/NLMPC.hpp) and registers the two launch thunks here.  The library keeps owning the workspace, bounds,
 *     parameters and every entry point below; `hooks` (the closure objects, hooks_bytes of them) is copied to HBM and
 *     handed to the kernels.                                                                                           */
typedef struct mpcx_nlmpc_custom {
    int nx, nu, ny, ph, ch, nineq, neq_user;
    int has_output;        /* an output function was set (otherwise outputs read as zeros, Model.hpp:72-96)              */
    int vector_hooks;      /* 1: hooks with the reference's whole-vector signatures (mpcx::HookModel); 0: a zoo-style
                              model struct with component-wise constraints and declared structure                        */
    const void *hooks; int hooks_bytes;
    /* mpcx::engine::launch_evaluate<Model> / launch_solve<Model>; the struct arguments are mpcx::NlmpcDev,
     * mpcx::NlmpcBatchDev, mpcx::NlmpcSolveDev (include/mpcx/nlmpc_device.hpp), ctx is passed back unchanged          */
    int (*launch_evaluate)(void *ctx, const void *dev, const void *batch, void *stream);
    int (*launch_solve)(void *ctx, const void *dev, const void *solve, void *stream);
    void *launch_ctx;
} mpcx_nlmpc_custom;
/* Ts > 0: the state function is a vector field dx/dt and the transcription is trapezoidal collocation with step Ts
 * (NLMPC::setDiscretizationSamplingTime, NLMPC.hpp:80-90); Ts <= 0: the state function returns x(k+1).                  */
int mpcx_nlmpc_create_custom(const mpcx_nlmpc_custom *c, double Ts, int device, mpcx_nlmpc_t *out);

/* (2) compiled at run time (hipRTC).  The hooks are given as C++ source: each string is the BODY of the corresponding
 *     reference lambda, with these parameter names --
 *       state_fn      (dx, x, u, step)           fills dx (or the next state, Ts <= 0)          required
 *       objective_fn  (x, y, u, e)               returns the cost                               required
 *       ineq_fn       (in_con, x, y, u, e)       fills in_con, in_con <= 0                      NULL when nineq = 0
 *       eq_fn         (eq_con, x, u)             fills eq_con, eq_con = 0                       NULL when neq_user = 0
 *       output_fn     (y, x, u, step)            fills y                                        NULL: outputs are zeros
 *     with the reference's types (mpc::cvec<n>, mpc::mat<ph+1, n>, include/mpcx/matrix.hpp) and the constants num_states,
 *     num_inputs, num_output, pred_hor, ctrl_hor, ineq_c, eq_c in scope; `preamble` (may be NULL) is pasted before them at
 *     namespace scope (constants, helper __device__ functions).  Example, the reference's examples/vanderpol_ex.cpp:
 *       state_fn     "dx(0) = ((1.0 - (x(1) * x(1))) * x(0)) - x(1) + u(0); dx(1) = x(0);"
 *       objective_fn "return x.array().square().sum() + u.array().square().sum();"
 *       ineq_fn      "for (int i = 0; i < ineq_c; i++) in_con(i) = u(i, 0) - 0.5;"
 *     The compiler's log is available through mpcx_last_error() when compilation fails.  Needs libhiprtc.so.7.         */
typedef struct mpcx_nlmpc_source {
    int nx, nu, ny, ph, ch, nineq, neq_user;
    const char *preamble, *state_fn, *objective_fn, *ineq_fn, *eq_fn, *output_fn;
} mpcx_nlmpc_source;
int mpcx_nlmpc_create_from_source(const mpcx_nlmpc_source *src, double Ts, int device, mpcx_nlmpc_t *out);

/* NLMPC::setInputScale / setStateScale (NLMPC.hpp:108,123 -> Mapping::setInputScaling / setStateScaling, Mapping.hpp:71-86):
 * the hooks see U = input_scale * z_u and X = [x0; z_x] / state_scale (Mapping.hpp:174-211), and the Jacobians carry the
 * factors the reference gives them, including what it leaves out (no chain rule on the cost's state gradient,
 * Objective.hpp:107-144; inequality state columns multiplied, not divided, Constraints.hpp:269-284).  nu / nx doubles.   */
int mpcx_nlmpc_set_input_scale(mpcx_nlmpc_t h, const double *scaling);
int mpcx_nlmpc_set_state_scale(mpcx_nlmpc_t h, const double *scaling);
/* Device pointers, fp64.  z [B x nz] (layout [x_1..x_ph | u blocks (ch) | slack]), x0 [B x nx].
 * Any output may be NULL.  cost [B]; grad [B x nz]; ceq [B x neq]; cineq [B x (nineq + neq_user)]: the user
 * inequalities, then the user equalities (Constraints::evaluateEq, Constraints.hpp:365-442);
 * jineq [B x (nineq + neq_user) x nz] row-major (dense: a user constraint may depend on anything);
 * jeq [B x ph x nx x jeq_w] row-major blocks [dc_i/dx_i | dc_i/dx_{i+1} | dc_i/du_i] -- the
 * non-zeros of the reference's dense [neq x nz] Jacobian: block i sits in rows i*nx.., columns
 * (i-1)*nx.. (absent for i = 0, x_0 is data), i*nx.., ph*nx + min(i, ch-1)*nu...                */
int mpcx_nlmpc_evaluate_batch(mpcx_nlmpc_t h, int batch, const double *z, const double *x0,
                              double *cost, double *grad, double *ceq, double *jeq,
                              double *cineq, double *jineq, void *stream);

/* mpc::NLParameters (Types.hpp:99-144), field for field.  The iteration stops when the largest step component falls below
 * 1e-6 * max(1, |z|_inf) with the dynamics defects below 1e-8 (solver_status 4), or when the line search finds no decrease
 * (finite-difference noise floor).  Positive tolerances add nlopt's own rules (set_ftol_rel / set_ftol_abs / set_xtol_rel /
 * set_xtol_abs, NLOptimizer.hpp:135-138, unit x weights :140), tested like SLSQP tests them, after a step that ends at a
 * feasible point: |f - f_prev| < absolute_ftol or < relative_ftol * (|f| + |f_prev|) / 2 -> solver_status 3 (FTOL_REACHED);
 * sum |dz| <= relative_xtol * sum |z| or every |dz_i| < absolute_xtol -> 4 (XTOL_REACHED).  Negative (the reference's
 * default): disabled.                                                                                              */
typedef struct mpcx_nlparams {
    int maximum_iteration;    /* 100 */
    double time_limit;        /* 0, accepted and ignored */
    int enable_warm_start;    /* 0; warm starts are driven by mpcx_nlmpc_batch.z_warm */
    double relative_ftol, relative_xtol, absolute_ftol, absolute_xtol;   /* -1 */
    int hard_constraints;     /* 1: slack fixed at zero (NLOptimizer.hpp:182-186) */
} mpcx_nlparams;
void mpcx_nlparams_default(mpcx_nlparams *p);
/* NLMPC::setOptimizerParameters -> NLOptimizer::setParameters (NLOptimizer.hpp:129-195) */
int mpcx_nlmpc_set_optimizer_parameters(mpcx_nlmpc_t h, const mpcx_nlparams *p);

/* NLMPC::setStateBounds / setInputBounds, vector + HorizonSlice form (NLMPC.hpp:346-398 -> NLOptimizer.hpp:346-404):
 * box bounds on the states x_1..x_ph (slice over the prediction horizon) and on the input blocks (slice over the
 * control horizon); {-1,-1} = the whole horizon.  They become rows of the sub-problem.  A starting point outside
 * them is moved to (ub - lb) / 2 as NLOptimizer::fixOptimalSolution does (NLOptimizer.hpp:705-716).  Output bounds
 * do not exist for NLMPC (NLMPC.hpp:318-325 throws).                                                        */
int mpcx_nlmpc_set_state_bounds_slice(mpcx_nlmpc_t h, const double *lo, const double *hi, int start, int end);
int mpcx_nlmpc_set_input_bounds_slice(mpcx_nlmpc_t h, const double *lo, const double *hi, int start, int end);

/* One batched NLOptimizer::run (NLOptimizer.hpp:412-638).  Device pointers.  Outputs other than
 * cmd may be NULL.  status uses MPCX_STATUS_* (ResultStatus), solver_status nlopt's result codes
 * (3 FTOL_REACHED, 4 XTOL_REACHED, 5 MAXEVAL_REACHED, -1 FAILURE = inconsistent linearised constraints,
 * -3 OUT_OF_MEMORY = more than 128 rows active at once, -4 ROUNDOFF_LIMITED = line search stalled far from a solution,
 * -5 FORCED_STOP = the kernel's own consistency guard tripped: a wavefront reached a phase boundary with lanes missing -- never seen, tests assert it) as mapped at NLOptimizer.hpp:729-750; on failure
 * cmd = u0 and cost = inf as at :613-624.  is_feasible = every user inequality <= 1e-10 and every user
 * equality within 1e-10 (Constraints.hpp:157-202, tolerances NLMPC.hpp:166, 262).                 */
typedef struct mpcx_nlmpc_batch {
    int batch;
    const double *x0;          /* [B x nx] */
    const double *u0;          /* [B x nu] */
    const double *z_warm;      /* [B x nz] previous solutions, shifted one step on entry as at
                                  NLOptimizer.hpp:460-510; NULL = cold start (:431-451)            */
    double *cmd;               /* [B x nu] */
    double *cost;              /* [B] */
    int32_t *status, *solver_status, *is_feasible, *iterations;   /* [B] each */
    double *z;                 /* [B x nz] the optimal decision vectors (next call's z_warm; may be the
                                  same buffer as this call's z_warm: an instance reads its start
                                  before its result is written)                                     */
    double *seq_state;         /* [B x (ph+1) x nx] row-major, row 0 = x0                           */
    double *seq_input;         /* [B x (ph+1) x nu]                                                 */
    double *seq_output;        /* [B x (ph+1) x ny] Model::getOutput (Model.hpp:72-96)                */
    int warm_curvature;        /* extension, with z_warm only: 1 = also keep the curvature estimate the previous solve of
                                  the same batch left in the handle's workspace (NLopt restarts its BFGS matrix on every
                                  optimize(); the optimum is the same, the iterations are fewer)                   */
    double *multipliers;       /* extension, may be NULL: [B x (nineq + neq_user + nbnd)] multipliers of the last quadratic
                                  sub-problem -- user inequalities, user equalities, then the finite bounds in the order
                                  (index into z ascending; upper before lower); non-zero = in the active set        */
    const double *params;      /* extension, may be NULL: [B x n_params] -- every instance its own parameters of the built-in
                                  system (the constants the reference example's closures capture: each UGV its own obstacles
                                  and preferred velocity, each oscillator network its own mu and coupling), in the order of
                                  mpcx_nlmpc_create's `params`.  NULL: the controller's parameters for all.  Not for hook
                                  models (their constants live in the closures).                                     */
} mpcx_nlmpc_batch;
int mpcx_nlmpc_solve_batch(mpcx_nlmpc_t h, const mpcx_nlmpc_batch *b, void *stream);
/* The same for callers whose data lives in host memory (the reference's optimize(x0, lastU) is such a caller): stages,
 * solves on the default stream, copies back, synchronises.  Outputs other than cmd may be NULL; z_warm may be NULL. */
int mpcx_nlmpc_solve_host(mpcx_nlmpc_t h, int batch, const double *x0, const double *u0, const double *z_warm, double *cmd,
                          double *cost, int32_t *status, int32_t *solver_status, int32_t *is_feasible, int32_t *iterations,
                          double *z, double *seq_state, double *seq_input);
/* `repeats` back-to-back launches bracketed by HIP events on `stream`; mean milliseconds. */
int mpcx_nlmpc_time_solve_batch(mpcx_nlmpc_t h, const mpcx_nlmpc_batch *b, void *stream, int repeats, float *ms_mean);

/* ---- set-up utility -------------------------------------------------------- */
/* mpc::discretization<nx, nu>(A, B, Ts, Ad, Bd) (Utils.hpp:23-47) for a batch of continuous-time models on the
 * device: [Ad Bd; 0 I] = exp([[A B]; [0 0]] Ts).  Device pointers, column-major matrices per instance
 * (A [B x nx x nx], B [B x nx x nu]); Ts one value (ts_per_instance = 0) or [B].  The variant with a
 * disturbance matrix (Utils.hpp:63-89) is the same call with Be appended to B's columns.  nx + nu <= 48.   */
int mpcx_discretize_batch(int device, int nx, int nu, int batch, const double *A, const double *B, const double *Ts,
                          int ts_per_instance, double *Ad, double *Bd, void *stream);

/* ---- multi-GPU: the one collective on the path --------------------------------- */
/* The reference solves one controller per object and has no coupling between objects (LMPC.hpp:751), so a batch shards
 * as contiguous slices, one process per GPU, and nothing is exchanged during the solve.  Afterwards every rank
 * contributes its block of optimal controls and receives everybody's: one RCCL ncclAllGather of doubles over xGMI.
 * mpcx_comm_* wraps the communicator so that a C++ host needs neither rccl.h nor PyTorch: rank 0 calls
 * mpcx_comm_get_unique_id, ships the MPCX_COMM_ID_BYTES bytes to the other ranks by any means (file, socket, MPI,
 * torch.distributed store), then every rank calls mpcx_comm_create (collective).  RCCL is bound at run time
 * (librccl.so.1); without it these calls return MPCX_E_DEVICE.                                                      */
#define MPCX_COMM_ID_BYTES 128
typedef struct mpcx_comm *mpcx_comm_t;
int mpcx_comm_get_unique_id(void *id_out /* MPCX_COMM_ID_BYTES bytes */);
int mpcx_comm_create(int device, int rank, int world, const void *id, mpcx_comm_t *out);
int mpcx_comm_destroy(mpcx_comm_t c);
int mpcx_comm_rank(mpcx_comm_t c);
int mpcx_comm_world(mpcx_comm_t c);
/* u_local [rows_per_rank x nu] -> u_all [world x rows_per_rank x nu], device pointers, every rank the same
 * rows_per_rank (ragged shards: pad the local block to the largest shard, libmpc_amd/distributed.py shows how).
 * Enqueued on `stream` -- pass the stream of the preceding mpcx_*_solve_batch so that the collective starts when the
 * solve retires, with no host synchronisation in between.  u_all may not alias u_local.                              */
int mpcx_allgather_u(mpcx_comm_t c, const double *u_local, int rows_per_rank, int nu, double *u_all, void *stream);

/* ---- heterogeneous batches (every instance its own model) ------------------------------------------
 * In the reference every controller object owns its model, weights and bounds (LMPC.hpp:751; ProblemBuilder.hpp:184-211,
 * 642-825).  A bank is K configured controllers (host-only handles are enough) with the same dimensions and the same pattern
 * of finite bounds, solved together: instance b of a batch uses controller model_index[b] (device array; NULL: controller b,
 * batch = K).  "Shared" references mean each controller's own setReferences / setExogenousInputs values.  The bank copies what
 * it needs: the controllers may be destroyed or changed afterwards (changes do not reach the bank).
 * Preconditions the library does not check on the device: every model_index[b] lies in [0, K) (an index outside reads another
 * controller's -- or no -- factors); a HIP graph captured from a solve holds the handle's workspace pointers, which a later solve
 * with a LARGER batch re-allocates: re-capture after growing the batch.
 * Set-up errors name the controller: MPCX_E_NUMERIC (its condensed Hessian or ADMM matrix does not factor), MPCX_E_INVALID (its
 * constraint structure differs from controller 0's, e.g. a constraint row that does not depend on the inputs in one of the two). */
typedef struct mpcx_lmpc_hetero *mpcx_lmpc_hetero_t;
int mpcx_lmpc_hetero_create(const mpcx_lmpc_t *controllers, int count, int device, mpcx_lmpc_hetero_t *out);
/* condense_on_host = 1: every controller's prediction matrices, Hessian, factors and dual Hessian on the host cores (what a single
 * controller's set-up does); 0: on the device, one workgroup per controller, the products on the f64 matrix pipe (falls back to the
 * host when the condensed problem exceeds 96 variables / the kernel's LDS plan)                                                 */
int mpcx_lmpc_hetero_create_ex(const mpcx_lmpc_t *controllers, int count, int device, int condense_on_host, mpcx_lmpc_hetero_t *out);
int mpcx_lmpc_hetero_destroy(mpcx_lmpc_hetero_t f);
int mpcx_lmpc_hetero_get_info(mpcx_lmpc_hetero_t f, int *count, int *active_words, int *m_ref, double *bytes_per_model);
int mpcx_lmpc_hetero_solve_batch(mpcx_lmpc_hetero_t f, const mpcx_lmpc_batch *b, const int32_t *model_index, void *stream);
int mpcx_lmpc_hetero_time_solve_batch(mpcx_lmpc_hetero_t f, const mpcx_lmpc_batch *b, const int32_t *model_index, void *stream,
                                      int repeats, float *ms_mean);

/* Sharding: this handle solves contiguous shards of a batch of `total` instances -- the kernel form is chosen for
 * the whole batch's size, so that a shard's results are bit for bit the rows of the unsharded solve (0 = every call is a whole batch). */
int mpcx_lmpc_set_total_batch(mpcx_lmpc_t h, int total);

/* ---- profiling and testing aids ------------------------------------------------------------------------------------
 * Not part of the reference-facing surface, but part of the exported ABI: bench.py's roofline block, tools/ and tests/ call
 * them, so they are declared (and kept) here.  "debug" in a name = may change between versions.                        */
/* mean time (ms) of [0] the assemble kernel, [1] the solve kernel, [2] the ADMM fallback kernel of one batch, each timed alone
 * with HIP events on `stream` over `repeats` launches (one-kernel forms: [0] = 0, [1] = the whole step's kernel; controllers whose
 * cost comes from its definition -- "flags"[0] below -- : [1] = lmpc_solve + lmpc_cost_mfma, launched back to back)                */
int mpcx_lmpc_debug_time_kernels(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, void *stream, int repeats, float *ms3);
/* condensed arrays of the host set-up by name ("H", "Kinv", "Gr", "Gc", "Y", "lw", "uw", "rho_b", "lg0", "ug0", "rho_g", "dims",
 * "dims_maps", "MA0", "MA1", "g_refrow", "g_step", "g_kind", "g_comp", "flags" = [cost from its definition, one-workgroup form
 * available, fused mat-vec form available]); out = NULL returns the length                                                      */
int mpcx_lmpc_debug_get(mpcx_lmpc_t h, const char *name, double *out, int cap);
/* how many full set-ups (condensing + device rebuild) and how many reference-only refreshes have run on this handle */
int mpcx_lmpc_debug_setup_counts(mpcx_lmpc_t h, int *full, int *refs);
/* solve path: 0 = assemble and solve as two kernels; 1 = the record computed inside the solve kernel by one mat-vec (persistent
 * form from 1024 instances on); 2 = assemble + solve in one workgroup of sixteen wavefronts; -1 = automatic (the default: the
 * workgroup form up to 4096 instances, two kernels beyond; the fused forms only on request)                                   */
int mpcx_lmpc_debug_use_fused(mpcx_lmpc_t h, int mode);
/* 1 = always the generic (roll-out) assemble kernel, whatever the reference layout */
int mpcx_lmpc_debug_force_generic(mpcx_lmpc_t h, int on);
/* rounds the polish-only kernel may spend before an instance goes to the ADMM kernel (default 30), and the ADMM iterations
 * between two polish attempts there (default 10); rounds0 = 1 sends nearly every instance through the ADMM path            */
int mpcx_lmpc_debug_set_rounds(mpcx_lmpc_t h, int rounds0, int check_every);
/* device buffer [B x 8] of int64 receiving per-instance cycle stamps of the solve kernel (NULL: off) */
int mpcx_lmpc_debug_set_cycle_buffer(mpcx_lmpc_t h, void *dev_ptr);
/* the SQP kernel's own convergence test: step length relative to max(1, |z|) and largest constraint defect */
int mpcx_nlmpc_debug_set_tolerances(mpcx_nlmpc_t h, double tol_step, double tol_con);
/* which kernel the last solve of a built-in system went through: 0 = nlmpc_sqp (one wavefront per instance, the reduced problem in a
 * per-instance HBM workspace), 1 | 2 | 4 = nlmpc_sqp_wg (one workgroup of that many wavefronts per instance, the reduced problem in LDS);
 * -1 = none yet.  This one: the last launch of the process.  MPCX_NLMPC_FORM=wg|wave, MPCX_NLMPC_WAVES=1|2|4|8 and MPCX_NLMPC_BLOCKS=1|0 in
 * the environment force a form for the handles created afterwards (measurements, tests/test_nlmpc_forms.py). */
int mpcx_nlmpc_debug_last_form(void);
/* the same for the last solve of this handle (the form is a property of the controller: chosen when its bounds are set, from the plan of the
 * workgroup form, the same for every batch size -- a shard of a batch takes the kernel the whole batch would; the overrides are read when
 * the handle is created) */
int mpcx_nlmpc_last_form(mpcx_nlmpc_t h);
/* one instance's slice of the SQP workspace copied to the host, with the offsets of its arrays (NlmpcWsLayout) */
int mpcx_nlmpc_debug_get_ws(mpcx_nlmpc_t h, int instance, double *out, int cap, int *layout, int nlayout);
/* user hooks given as source: the translation unit the run-time compiler is fed / a compile-only check (> 0: it builds, the size of the
 * code object; a negative MPCX_E_* code otherwise, the compiler's log through mpcx_last_error()) */
int mpcx_nlmpc_debug_generated_source(const mpcx_nlmpc_source *src, char *out, int cap);
int mpcx_nlmpc_debug_compile_source(const mpcx_nlmpc_source *src);
/* one O(n^3) array ("H", "Kinv", "Gr", "Gc", "Y", "rho_b", "rho_g"; "flags" = [cost_direct, condensed on the device]) of controller k
 * of a bank, copied to the host; out = NULL returns the length */
int mpcx_lmpc_hetero_debug_get(mpcx_lmpc_hetero_t f, int k, const char *name, double *out, int cap);

const char *mpcx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPCX_H */
