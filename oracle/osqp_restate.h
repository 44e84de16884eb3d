/* TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
 *
 * CPU restatement of the OSQP v0.6.3 algorithm as libmpc++ drives it from
 * LOptimizer::run (reference include/mpc/LMPC/LOptimizer.hpp:244-284).
 * OSQP itself is a third-party dependency pinned by configure.sh:39-42 and is
 * NOT in /root/reference; the algorithm is restated from the OSQP paper (cited
 * at docs/source/cite/cite.rst:60-70) with v0.6.3's default constants.
 *
 * PARITY STATUS: pinned on the polished solution by the reference's own known
 * answer (test/LMPC/test_common.cpp:230-236) -- see tests/test_oracle.py.
 * ADMM iterates/iteration counts are NOT reproducible against a real OSQP run:
 * OSQP derives its adaptive-rho interval from wall-clock time; here it is fixed
 * (default 25 iterations).
 */
#ifndef ORACLE_OSQP_RESTATE_H
#define ORACLE_OSQP_RESTATE_H

#include "sparse_ldl.h"

/* OSQP status values (v0.6.3 constants.h) */
#define OQ_DUAL_INFEASIBLE_INACCURATE   4
#define OQ_PRIMAL_INFEASIBLE_INACCURATE 3
#define OQ_SOLVED_INACCURATE            2
#define OQ_SOLVED                       1
#define OQ_MAX_ITER_REACHED            -2
#define OQ_PRIMAL_INFEASIBLE           -3
#define OQ_DUAL_INFEASIBLE             -4
#define OQ_SIGINT                      -5
#define OQ_TIME_LIMIT_REACHED          -6
#define OQ_NON_CVX                     -7
#define OQ_UNSOLVED                   -10

typedef struct {
    /* libmpc++ LParameters (Types.hpp:99-161) */
    double alpha, rho, eps_rel, eps_abs, eps_prim_inf, eps_dual_inf;
    int max_iter, adaptive_rho, polish, warm_start;
    /* OSQP v0.6.3 defaults the reference inherits */
    double sigma, delta, adaptive_rho_tolerance;
    int scaling, adaptive_rho_interval, check_termination, polish_refine_iter;
    /* 1: reproduce IEEE inf*0 = NaN in the primal-infeasibility test when the caller passes
     *    true infinities -- libmpc++ does (mpc::inf, Types.hpp:227), so with the reference an
     *    infeasible QP is never reported PRIMAL_INFEASIBLE: it runs to max_iter and returns the
     *    last ADMM iterate as MAX_ITER_REACHED.  The reference's own "Scalar constraints" test
     *    (test/LMPC/test_constraints.cpp:95-167, infeasible step-0 row) can only pass that way.
     *    The LMPC oracle defaults to 1.  0: infinite bounds contribute nothing and
     *    infeasibility is detected (OSQP's documented behaviour with 1e30 "infinities"). */
    int nan_faithful;
} oq_settings;

void oq_default_settings(oq_settings *s);

typedef struct {
    int status;        /* OQ_* */
    int iters;
    int polished;      /* 1 ok, -1 unsuccessful, 0 not attempted */
    int rho_updates;
    double rho;        /* final rho */
    double obj;        /* 0.5 x'Px + q'x (unscaled) */
    double pri_res, dua_res;
} oq_info;

typedef struct oq_cache oq_cache;   /* cached orderings (pattern-level only) */
oq_cache *oq_cache_new(void);
void oq_cache_free(oq_cache *c);

/* P: upper-triangular CSC (n x n); A: CSC (m x n).  x[n], y[m] out (and warm
 * start in when s->warm_start and warm_x/warm_y non-NULL).
 * act_lo/act_up (m bytes each, may be NULL): the active sets polish used. */
int oq_solve(const csc_t *P, const double *q, const csc_t *A, const double *l, const double *u,
             const oq_settings *s, oq_cache *cache, const double *warm_x, const double *warm_y,
             double *x, double *y, oq_info *info, unsigned char *act_lo, unsigned char *act_up);

#endif
