/* TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
 *
 * Sparse symmetric quasi-definite LDL^T used by the OSQP restatement in
 * osqp_restate.c.  The reference's solver stack (OSQP v0.6.3 -> QDLDL + AMD,
 * configure.sh:39-42) is not in /root/reference; this is a from-scratch
 * restatement of the published up-looking LDL^T algorithm (T. Davis,
 * "Algorithm 849: a concise sparse Cholesky factorization package", 2005 --
 * the algorithm QDLDL implements) plus a plain greedy minimum-degree ordering
 * standing in for AMD.  Same mathematics, different round-off and fill.
 */
#ifndef ORACLE_SPARSE_LDL_H
#define ORACLE_SPARSE_LDL_H

typedef struct {
    int n;        /* columns */
    int m;        /* rows    */
    int *p;       /* column pointers, n+1 */
    int *i;       /* row indices           */
    double *x;    /* values                */
    int nzmax;
} csc_t;

csc_t *csc_alloc(int m, int n, int nzmax);
void csc_free(csc_t *A);

typedef struct {
    int n;
    int *perm;     /* perm[k] = original index placed at position k */
    int *iperm;
    int *parent, *Lnz, *Lp, *Li;
    double *Lx, *D, *Dinv;
    /* permuted upper-triangular copy of K and the map from K's entries */
    int *Cp, *Ci; double *Cx; int *KtoC;
    /* work */
    int *flag, *pattern; double *y, *bp;
} ldl_t;

/* K: upper-triangular CSC of a symmetric matrix.  perm may be NULL (computed by
 * greedy minimum degree) or a cached ordering of length n. */
ldl_t *ldl_analyze(const csc_t *K, const int *perm);
/* numeric factorization using K's current values (same pattern as analyzed). */
int ldl_factor(ldl_t *F, const csc_t *K);
/* solve K x = b in place */
void ldl_solve(const ldl_t *F, double *b);
void ldl_free(ldl_t *F);
/* greedy minimum-degree ordering of a symmetric pattern (upper triangle given) */
void min_degree_order(const csc_t *K, int *perm);

#endif
