/* TEST INFRASTRUCTURE ONLY (oracle).  See sparse_ldl.h. */
#include "sparse_ldl.h"
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

csc_t *csc_alloc(int m, int n, int nzmax)
{
    csc_t *A = (csc_t *)calloc(1, sizeof(csc_t));
    A->m = m; A->n = n; A->nzmax = nzmax > 0 ? nzmax : 1;
    A->p = (int *)calloc((size_t)n + 1, sizeof(int));
    A->i = (int *)calloc((size_t)A->nzmax, sizeof(int));
    A->x = (double *)calloc((size_t)A->nzmax, sizeof(double));
    return A;
}

void csc_free(csc_t *A)
{
    if (!A) return;
    free(A->p); free(A->i); free(A->x); free(A);
}

/* ---- greedy minimum degree on a bitset adjacency --------------------------
 * Eliminating a node turns its neighbourhood into a clique; we keep explicit
 * adjacency rows as bitsets (n <= a few thousand for the MPC KKT systems). */
void min_degree_order(const csc_t *K, int *perm)
{
    int n = K->n;
    int W = (n + 63) / 64;
    uint64_t *adj = (uint64_t *)calloc((size_t)n * W, sizeof(uint64_t));
    int *deg = (int *)calloc((size_t)n, sizeof(int));
    char *gone = (char *)calloc((size_t)n, 1);
    for (int j = 0; j < n; j++)
        for (int p = K->p[j]; p < K->p[j + 1]; p++) {
            int i = K->i[p];
            if (i == j) continue;
            adj[(size_t)i * W + j / 64] |= 1ull << (j % 64);
            adj[(size_t)j * W + i / 64] |= 1ull << (i % 64);
        }
    for (int i = 0; i < n; i++) {
        int d = 0;
        for (int w = 0; w < W; w++) d += __builtin_popcountll(adj[(size_t)i * W + w]);
        deg[i] = d;
    }
    int *nb = (int *)malloc((size_t)n * sizeof(int));
    for (int k = 0; k < n; k++) {
        int best = -1, bd = 1 << 30;
        for (int i = 0; i < n; i++)
            if (!gone[i] && deg[i] < bd) { bd = deg[i]; best = i; }
        perm[k] = best;
        gone[best] = 1;
        uint64_t *rb = adj + (size_t)best * W;
        int cnt = 0;
        for (int w = 0; w < W; w++) {
            uint64_t bits = rb[w];
            while (bits) {
                int b = __builtin_ctzll(bits);
                bits &= bits - 1;
                nb[cnt++] = w * 64 + b;
            }
        }
        for (int a = 0; a < cnt; a++) {
            int i = nb[a];
            uint64_t *ri = adj + (size_t)i * W;
            for (int w = 0; w < W; w++) ri[w] |= rb[w];
            ri[best / 64] &= ~(1ull << (best % 64));
            ri[i / 64] &= ~(1ull << (i % 64));
            int d = 0;
            for (int w = 0; w < W; w++) d += __builtin_popcountll(ri[w]);
            deg[i] = d;
        }
        memset(rb, 0, (size_t)W * sizeof(uint64_t));
    }
    free(nb); free(adj); free(deg); free(gone);
}

/* symmetric permutation of an upper-triangular matrix: C = P K P^T (upper),
 * records for every entry of K where it lands in C. */
static void sym_perm(const csc_t *K, const int *iperm, int *Cp, int *Ci, double *Cx, int *KtoC)
{
    int n = K->n;
    int *w = (int *)calloc((size_t)n, sizeof(int));
    for (int j = 0; j < n; j++) {
        int j2 = iperm[j];
        for (int p = K->p[j]; p < K->p[j + 1]; p++) {
            int i = K->i[p];
            if (i > j) continue;
            int i2 = iperm[i];
            w[i2 > j2 ? i2 : j2]++;
        }
    }
    Cp[0] = 0;
    for (int j = 0; j < n; j++) { Cp[j + 1] = Cp[j] + w[j]; w[j] = Cp[j]; }
    for (int j = 0; j < n; j++) {
        int j2 = iperm[j];
        for (int p = K->p[j]; p < K->p[j + 1]; p++) {
            int i = K->i[p];
            if (i > j) { KtoC[p] = -1; continue; }
            int i2 = iperm[i];
            int col = i2 > j2 ? i2 : j2, row = i2 < j2 ? i2 : j2;
            int q = w[col]++;
            Ci[q] = row;
            if (Cx) Cx[q] = K->x[p];
            KtoC[p] = q;
        }
    }
    free(w);
}

ldl_t *ldl_analyze(const csc_t *K, const int *perm)
{
    int n = K->n, nz = K->p[n];
    ldl_t *F = (ldl_t *)calloc(1, sizeof(ldl_t));
    F->n = n;
    F->perm = (int *)malloc((size_t)n * sizeof(int));
    F->iperm = (int *)malloc((size_t)n * sizeof(int));
    if (perm) memcpy(F->perm, perm, (size_t)n * sizeof(int));
    else min_degree_order(K, F->perm);
    for (int k = 0; k < n; k++) F->iperm[F->perm[k]] = k;
    F->Cp = (int *)malloc(((size_t)n + 1) * sizeof(int));
    F->Ci = (int *)malloc((size_t)(nz > 0 ? nz : 1) * sizeof(int));
    F->Cx = (double *)malloc((size_t)(nz > 0 ? nz : 1) * sizeof(double));
    F->KtoC = (int *)malloc((size_t)(nz > 0 ? nz : 1) * sizeof(int));
    sym_perm(K, F->iperm, F->Cp, F->Ci, NULL, F->KtoC);
    F->parent = (int *)malloc((size_t)n * sizeof(int));
    F->Lnz = (int *)malloc((size_t)n * sizeof(int));
    F->Lp = (int *)malloc(((size_t)n + 1) * sizeof(int));
    F->flag = (int *)malloc((size_t)n * sizeof(int));
    F->pattern = (int *)malloc((size_t)n * sizeof(int));
    F->y = (double *)calloc((size_t)n, sizeof(double));
    F->bp = (double *)calloc((size_t)n, sizeof(double));
    F->D = (double *)malloc((size_t)n * sizeof(double));
    F->Dinv = (double *)malloc((size_t)n * sizeof(double));
    /* elimination tree and column counts */
    for (int k = 0; k < n; k++) {
        F->parent[k] = -1; F->flag[k] = k; F->Lnz[k] = 0;
        for (int p = F->Cp[k]; p < F->Cp[k + 1]; p++) {
            int i = F->Ci[p];
            if (i < k)
                for (; F->flag[i] != k; i = F->parent[i]) {
                    if (F->parent[i] == -1) F->parent[i] = k;
                    F->Lnz[i]++;
                    F->flag[i] = k;
                }
        }
    }
    F->Lp[0] = 0;
    for (int k = 0; k < n; k++) F->Lp[k + 1] = F->Lp[k] + F->Lnz[k];
    int lnz = F->Lp[n];
    F->Li = (int *)malloc((size_t)(lnz > 0 ? lnz : 1) * sizeof(int));
    F->Lx = (double *)malloc((size_t)(lnz > 0 ? lnz : 1) * sizeof(double));
    return F;
}

int ldl_factor(ldl_t *F, const csc_t *K)
{
    int n = F->n, nz = K->p[n];
    for (int p = 0; p < nz; p++)
        if (F->KtoC[p] >= 0) F->Cx[F->KtoC[p]] = K->x[p];
    int *Lp = F->Lp, *Li = F->Li, *Lnz = F->Lnz, *parent = F->parent, *flag = F->flag, *pattern = F->pattern;
    double *Lx = F->Lx, *D = F->D, *y = F->y;
    for (int k = 0; k < n; k++) {
        y[k] = 0.0;
        int top = n;
        flag[k] = k;
        Lnz[k] = 0;
        for (int p = F->Cp[k]; p < F->Cp[k + 1]; p++) {
            int i = F->Ci[p];
            y[i] += F->Cx[p];
            int len = 0;
            for (; flag[i] != k; i = parent[i]) { pattern[len++] = i; flag[i] = k; }
            while (len > 0) pattern[--top] = pattern[--len];
        }
        D[k] = y[k];
        y[k] = 0.0;
        for (; top < n; top++) {
            int i = pattern[top];
            double yi = y[i];
            y[i] = 0.0;
            int p2 = Lp[i] + Lnz[i];
            for (int p = Lp[i]; p < p2; p++) y[Li[p]] -= Lx[p] * yi;
            double lki = yi * F->Dinv[i];
            D[k] -= lki * yi;
            Li[p2] = k;
            Lx[p2] = lki;
            Lnz[i]++;
        }
        if (D[k] == 0.0) return k + 1;
        F->Dinv[k] = 1.0 / D[k];
    }
    return 0;
}

void ldl_solve(const ldl_t *F, double *b)
{
    int n = F->n;
    double *x = F->bp;
    for (int k = 0; k < n; k++) x[k] = b[F->perm[k]];
    for (int j = 0; j < n; j++) {
        double xj = x[j];
        for (int p = F->Lp[j]; p < F->Lp[j] + F->Lnz[j]; p++) x[F->Li[p]] -= F->Lx[p] * xj;
    }
    for (int j = 0; j < n; j++) x[j] *= F->Dinv[j];
    for (int j = n - 1; j >= 0; j--) {
        double s = x[j];
        for (int p = F->Lp[j]; p < F->Lp[j] + F->Lnz[j]; p++) s -= F->Lx[p] * x[F->Li[p]];
        x[j] = s;
    }
    for (int k = 0; k < n; k++) b[F->perm[k]] = x[k];
}

void ldl_free(ldl_t *F)
{
    if (!F) return;
    free(F->perm); free(F->iperm); free(F->parent); free(F->Lnz); free(F->Lp); free(F->Li);
    free(F->Lx); free(F->D); free(F->Dinv); free(F->Cp); free(F->Ci); free(F->Cx); free(F->KtoC);
    free(F->flag); free(F->pattern); free(F->y); free(F->bp);
    free(F);
}
